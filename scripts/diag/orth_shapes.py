"""K6 (one classical Gram-Schmidt pass, asynchronous entry) at the shapes where few columns are streamed: GMRES on the waveguide
(rows = 1 001 997, k = 2 .. 24, full columns) and the first steps of an iar run on gun (rows = n (k + 1), block-triangular).
Checks against torch, then HIP-event time per call and fraction of the HBM peak.   python scripts/diag/orth_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
from nep_amd import dense


def one(rows, k, stair_n=None, reps=30, check=True):
    g = torch.Generator(device="cuda"); g.manual_seed(rows + k)
    V = torch.zeros((k, rows), dtype=torch.complex128, device="cuda")
    if stair_n:
        for j in range(k):
            a = (j + 1) * stair_n
            V[j, :a] = torch.randn(a, dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn(a, dtype=torch.float64, device="cuda", generator=g)
        active = torch.from_numpy((np.arange(1, k + 1) * stair_n).astype(np.int64)).to("cuda")
        byts = 2 * 16 * int(active.sum()) + 3 * 16 * rows
    else:
        V.copy_(torch.randn((k, rows), dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn((k, rows), dtype=torch.float64, device="cuda", generator=g))
        active = None
        byts = 2 * 16 * rows * k + 3 * 16 * rows
    V /= np.sqrt(rows)
    w0 = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    out = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
    err = None
    if check:
        w = w0.clone()
        dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=active, method=dense.CGS)
        torch.cuda.synchronize()
        h = V.conj() @ w0
        wr = w0 - V.T @ h
        beta = torch.linalg.norm(wr)
        err = max(float(torch.linalg.norm(out[:k] - h) / torch.linalg.norm(h)), float(torch.linalg.norm(w - wr / beta)),
                  float(abs(out[k].real - beta) / beta))
    w = w0.clone()
    for _ in range(3):
        dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=active, method=dense.CGS)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=active, method=dense.CGS)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("rows %8d k %3d %-5s  %7.1f us  %6.0f GB/s  frac %.3f  err %s" % (rows, k, "stair" if stair_n else "full", us, byts / us / 1e3,
                                                                        byts / us / 1e3 / 8000, "%.1e" % err if err is not None else "-"), flush=True)


print("NEP_ORTH_ROWS_K =", os.environ.get("NEP_ORTH_ROWS_K", "(default)"))
if os.environ.get("ORTH_SHAPES"):                     # "rows:k[:n]" items, e.g. 1001997:12,1005556:100:9956 (for a profiler run)
    for it in os.environ["ORTH_SHAPES"].split(","):
        f = [int(x) for x in it.split(":")]
        one(f[0], f[1], stair_n=f[2] if len(f) > 2 else None, check=False, reps=50)
    sys.exit(0)
one(100, 3); one(1000, 7); one(70000, 41)
for k in (2, 4, 8, 12, 16, 20, 24, 40, 41, 60):
    one(1001997, k, check=k in (8, 40, 41))
n = 9956
for k in (4, 8, 16, 24, 32, 40, 41, 48, 64, 100):
    one(n * (k + 1), k, stair_n=n, check=k in (8, 40, 100), reps=20 if k < 64 else 8)
