import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from threadpoolctl import ThreadpoolController
from oracle import wep as ow
nx, nz = int(sys.argv[1]), int(sys.argv[2])
A = sp.csc_matrix(ow.WEP_FD(nx, nz, "JARLEBRING").compute_Mder(-3 - 3.5j), dtype=complex)
ctl = ThreadpoolController()
kw = dict(permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.001, options=dict(SymmetricMode=True), panel_size=8, relax=4)
for spec in sys.argv[3:]:
    nt, ps, rl = [int(x) for x in (spec.split(",") + ["8", "4"])[:3]]
    kw.update(panel_size=ps, relax=rl)
    with ctl.limit(limits=nt, user_api="blas"):
        t0 = time.perf_counter(); lu = spla.splu(A, **kw); dt = time.perf_counter() - t0
    print("blas threads", nt, "panel", ps, "relax", rl, "factor %.2f s" % dt, flush=True)
    del lu
