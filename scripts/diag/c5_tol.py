"""C5 with different inner tolerances (first solve / refinement sweeps): python scripts/diag/c5_tol.py"""
import os, sys, time
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nep_amd as na, torch
import baseline_configs as bc
bc.c5_device(na)
ref = None
for rt, st in [(1e-9, None), (1e-9, 1e-6), (1e-9, 1e-4), (1e-9, 1e-3), (1e-7, 1e-5), (1e-6, 1e-6), (1e-6, 1e-4), (1e-5, 1e-5)]:
    tm = {}
    lam, Q, res, info = bc.c5_device(na, reltol=rt, sweep_reltol=st, timers=tm)
    Qh = na.to_host(Q) if not isinstance(Q, np.ndarray) else Q
    hres = bc.c5_host_residuals(1003, 999, lam, Qh)
    if ref is None:
        ref = lam
    ok, worst = bc.match(lam, ref, 1e-8) if len(lam) == len(ref) else (False, None)
    print("reltol %g sweep %s: solve_s %.3f (solve phase %.3f) pairs %d maxres %.2e host %.2e eig-vs-first %s %s" % (
        rt, st, info["solve_s"], tm.get("solve", 0), len(lam), max(res), max(hres), ok, worst), flush=True)
