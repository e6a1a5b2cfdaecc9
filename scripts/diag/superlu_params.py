import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "1")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
from oracle import gallery as og
if len(sys.argv) > 1 and sys.argv[1] == "wep":
    from oracle import wep as ow
    o = ow.WEP_FD(int(sys.argv[2]), int(sys.argv[3]), "JARLEBRING")
    A = sp.csc_matrix(o.compute_Mder(-3 - 3.5j), dtype=complex)
else:
    nep = og.gun_spmf_scaled()
    A = sp.csc_matrix(nep.compute_Mder(0.0), dtype=complex)
base = dict(permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.001, options=dict(SymmetricMode=True))
def t(**kw):
    best = 1e9
    for _ in range(int(os.environ.get("REPS", "3"))):
        t0 = time.perf_counter(); lu = spla.splu(A, **dict(base, **kw)); best = min(best, time.perf_counter() - t0)
    return best * 1e3, lu.L.nnz + lu.U.nnz
print("default", t())
combos = [(1, 1), (4, 2), (8, 4)] if os.environ.get("QUICK") else [(p_, r_) for p_ in (1, 8, 20) for r_ in (1, 2, 4, 8, 16)]
for ps, rl in combos:
    if True:
        print("panel_size", ps, "relax", rl, "%.1f ms nnz %d" % t(panel_size=ps, relax=rl))
