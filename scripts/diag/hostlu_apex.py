"""host-LU iar step with and without the deferred apex: where does the time go"""
import os, sys, time
os.environ["NEP_LU_DEV"] = "0"
os.environ["NEP_IAR_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled", 9956); nep.dev
def step():
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    return na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator, return_device=True)
for _ in range(5): step()
