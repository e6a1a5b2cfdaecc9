"""C5 (waveguide tiar at n = 1e6): GMRES iteration counts and where the solve phase goes.  python scripts/diag/c5_split.py"""
import json, os, sys, time
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd import linsolvers, wep_linsolvers
its = []
orig = linsolvers.GMRESLinSolver.solve_dev
def wrapped(self, b, out=None, scale=1.0, tol=None):
    i0 = self.iterations if isinstance(self.iterations, int) else 0
    t0 = time.perf_counter(); r = orig(self, b, out=out, scale=scale, tol=tol); torch.cuda.synchronize()
    its.append((self.iterations, time.perf_counter() - t0))
    return r
linsolvers.GMRESLinSolver.solve_dev = wrapped
for rep in range(2):
    del its[:]
    tm = {}
    t0 = time.perf_counter()
    lam, Q, res, info = bc.c5_device(na, timers=tm)
    print(json.dumps({"eigenpairs": len(lam), "solve_s": info["solve_s"], "setup_s": info.get("preconditioner_setup_s"), "gmres_calls": len(its),
                      "gmres_s": sum(t for _, t in its), "gmres_iterations": int(sum(i for i, _ in its)), "phases": {k: round(v, 4) for k, v in tm.items()}, "maxres": max(res)}))
