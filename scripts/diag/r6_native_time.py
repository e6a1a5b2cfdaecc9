import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nep_amd as na
import torch
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf_scaled")
import contextlib
side = torch.cuda.Stream() if os.environ.get("DIAG_SIDE_STREAM") else None
ts = []
for r in range(int(os.environ.get("REPS", "24"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        out = na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if r == 0:
        _DeviceRefactor.wait()
print("pairs %d; ms per call: median %.2f min %.2f (calls 4..)" % (len(out[0]), np.median(ts[4:]), np.min(ts[4:])), " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("NEP_")))
