"""nep_wep_sylv_solve at 999 x 1003: error against the dense diagonalisation and time per solve, for the DFT variant selected by
NEP_WEP_DFT_SYM (0 = plain dense stages; 42 / 43 / 22 / 23 = symmetric-half stages, columns per workgroup * 10 + k per thread)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
L_ = na._lib.lib
for nz, nx in ((299, 303), (999, 1003)):
    rng = np.random.default_rng(nz)
    hx, hz, sigma, kbar = 0.37, 0.21, -3 - 3.5j, 2.1 + 0.3j
    v = np.zeros(nz, dtype=complex); v[0] = -2; v[1] += 1; v[nz - 1] += 1; v /= hz ** 2
    w = np.zeros(nz, dtype=complex); w[1] += 1; w[nz - 1] += -1; w *= sigma / hz
    D = np.fft.fft(v + w) + (sigma ** 2 + kbar)
    S = -(4.0 / hx ** 2) * np.sin(np.pi * np.arange(1, nx + 1) / (2 * (nx + 1))) ** 2
    jx = np.arange(1, nx + 1)
    W = np.sqrt(2.0 / (nx + 1)) * np.sin(np.pi * np.outer(jx, jx) / (nx + 1))
    Cm = rng.standard_normal((nz, nx)) + 1j * rng.standard_normal((nz, nx))
    ref = np.fft.fft((np.fft.ifft(Cm @ W, axis=0)) / (D[:, None] + S[None, :]), axis=0) @ W
    h = C.c_void_p()
    Dc = np.ascontiguousarray(D)
    assert L_.nep_wep_sylv_create(nz, nx, na._lib.hptr(Dc), 1.0 / hx ** 2, C.byref(h)) == 0
    Xd = na.to_dev(Cm)
    assert L_.nep_wep_sylv_solve(h, C.c_void_p(Xd.data_ptr()), None) == 0
    X = na.to_host(Xd)
    err = np.linalg.norm(X - ref) / np.linalg.norm(ref)
    for _ in range(5):
        L_.nep_wep_sylv_solve(h, C.c_void_p(Xd.data_ptr()), None)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        L_.nep_wep_sylv_solve(h, C.c_void_p(Xd.data_ptr()), None)
    e1.record(); torch.cuda.synchronize()
    print("NEP_WEP_DFT_SYM=%s  %d x %d: rel err %.2e, %.1f us per solve" % (os.environ.get("NEP_WEP_DFT_SYM", "(default)"), nz, nx, err,
                                                                     e0.elapsed_time(e1) / 200 * 1e3), flush=True)
    L_.nep_wep_sylv_destroy(h)
