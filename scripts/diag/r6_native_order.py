import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nep_amd as na
import torch
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf_scaled")
def run(native, reps=6):
    os.environ["NEP_IAR_NATIVE_RUN"] = "1" if native else "0"
    ts = []
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        creator = na.FactorizeLinSolverCreator(max_factorizations=0)
        out = na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        _DeviceRefactor.wait()
    return np.round(np.array(ts) * 1e3, 1)
order = sys.argv[1] if len(sys.argv) > 1 else "npn"
for c in order:
    print("native" if c == "n" else "python", run(c == "n"), flush=True)
