"""device numeric factorisation vs the host factor on the gun matrix: values, solve, timing"""
import os, sys, time, ctypes as C
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
import nep_amd as na
from nep_amd._lib import lib, check, hptr, c_vp
from nep_amd.nep import stream_ptr
import nep_amd_hostlu as hl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9956
nep = na.nep_gallery("gun_spmf_scaled", n)
A = sp.csc_matrix(nep.compute_Mder(0.0)).astype(np.complex128); A.sort_indices()
F = hl.factor(A.data, A.indices, A.indptr, A.shape)
lu = na.DeviceLU(factors=F)
print("block schedule", lu.block_schedule, "levels", lu.levels, "blocks", lu.blocks)
h = c_vp()
t0 = time.perf_counter()
check(lib.nep_lu_refac_create(lu.h, A.shape[0], hptr(F["Lp"]), hptr(F["Li"]), hptr(F["Up"]), hptr(F["Ui"]), hptr(F["perm_r"]), hptr(F["perm_c"]),
                              hptr(np.ascontiguousarray(A.indptr, dtype=np.int32)), hptr(np.ascontiguousarray(A.indices, dtype=np.int32)), C.byref(h)))
info = (C.c_int64 * 6)(); check(lib.nep_lu_refac_info(h, info))
print("refac_create %.1f ms" % ((time.perf_counter() - t0) * 1e3), "info", list(info))
for lam in (0.0, 0.2 + 0.1j):
    A2 = sp.csc_matrix(nep.compute_Mder(lam)).astype(np.complex128); A2.sort_indices()
    assert np.array_equal(A2.indices, A.indices)
    F2 = hl.factor(A2.data, A2.indices, A2.indptr, A2.shape)
    same_piv = np.array_equal(F2["perm_r"], F["perm_r"]) and np.array_equal(F2["Lp"], F["Lp"])
    LU = np.empty(len(F["Lx"]) + len(F["Ux"]), dtype=np.complex128); health = np.zeros(3)
    out = c_vp()
    Ax = np.ascontiguousarray(A2.data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.nep_lu_factor_dev(h, hptr(Ax), 200, 1e8, hptr(health), hptr(LU), C.byref(out), stream_ptr()))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    # compare as matrices (the entry order inside a column may differ between two host factorisations)
    nn = A.shape[0]; nL = len(F["Lx"])
    Ld = sp.csc_matrix((LU[:nL], F["Li"], F["Lp"]), shape=(nn, nn)); Ud = sp.csc_matrix((LU[nL:], F["Ui"], F["Up"]), shape=(nn, nn))
    Lh = sp.csc_matrix((F2["Lx"], F2["Li"], F2["Lp"]), shape=(nn, nn)); Uh = sp.csc_matrix((F2["Ux"], F2["Ui"], F2["Up"]), shape=(nn, nn))
    dL = abs(Ld - Lh).max() / abs(Lh).max(); dU = abs(Ud - Uh).max() / abs(Uh).max()
    print("lam", lam, "same pivots as first:", same_piv, "health", health, "rel diff L %.2e U %.2e" % (dL, dU))
    print("   factor_dev returned after %.2f ms, device done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    # solve through the new handle
    b = np.random.default_rng(0).standard_normal(A.shape[0]) + 0j
    bd = torch.from_numpy(b).to("cuda"); x = torch.empty_like(bd)
    check(lib.nep_lu_solve(out, 1, c_vp(bd.data_ptr()), A.shape[0], c_vp(x.data_ptr()), A.shape[0], 1.0, stream_ptr()))
    xs = x.cpu().numpy()
    print("   solve residual %.2e" % (np.linalg.norm(A2 @ xs - b) / np.linalg.norm(b)))
    lib.nep_lu_destroy(out)
# timing loop without the value read-back
Ax = np.ascontiguousarray(A.data); ts = []
for rep in range(6):
    out = c_vp(); torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.nep_lu_factor_dev(h, hptr(Ax), 200, 1e8, None, None, C.byref(out), stream_ptr()))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3)); lib.nep_lu_destroy(out)
print("factor_dev (host returns, device done) ms:", [("%.2f" % a, "%.2f" % b) for a, b in ts])
lib.nep_lu_refac_destroy(h)
