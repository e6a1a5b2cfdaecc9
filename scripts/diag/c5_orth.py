import json, os, sys, time
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd import linsolvers
its = []
orig = linsolvers.GMRESLinSolver.solve_dev
def wrapped(self, b, out=None, scale=1.0, tol=None):
    r = orig(self, b, out=out, scale=scale, tol=tol); its.append(self.iterations); return r
linsolvers.GMRESLinSolver.solve_dev = wrapped
for orth in (sys.argv[1:] or ["dgks", "cgs"]):
    os.environ["NEP_GMRES_ORTH_FORCE"] = orth
    for rep in range(2):
        del its[:]
        lam, Q, res, info = bc.c5_device(na)
    print(json.dumps({"orth": orth, "eigenpairs": len(lam), "solve_s": info["solve_s"], "gmres_calls": len(its), "gmres_iterations": int(sum(its)), "maxres": max(res)}))
