import os, sys
os.environ["NEP_LU_DEV"] = sys.argv[1] if len(sys.argv) > 1 else "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
from nep_amd.linsolvers import FactorizeLinSolver
orig = FactorizeLinSolver.review_recorded
cnt = [0]
def spy(self, w, plan):
    ok = orig(self, w, plan)
    cnt[0] += 1
    if not ok:
        print("review #%d plan %d w %s -> %s (next plan %s)" % (cnt[0], plan, [float(x) for x in w[:plan + 1]], ok, self._recorded_plan), flush=True)
    return ok
FactorizeLinSolver.review_recorded = spy
nep = na.nep_gallery("gun_spmf_scaled", 9956); nep.dev
def step():
    cnt[0] = 0
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    return na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator, return_device=True)
for i in range(40):
    step()
