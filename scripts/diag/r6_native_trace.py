import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nep_amd as na
import torch
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf_scaled")
for r in range(8):
    if r >= 4:
        os.environ["NEP_IAR_RUN_TRACE"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    out = na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator)
    torch.cuda.synchronize(); print("call %d: %.2f ms" % (r, (time.perf_counter() - t0) * 1e3), flush=True)
    if r == 0:
        _DeviceRefactor.wait()
