import os, sys, time
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled", 9956); nep.dev
def step():
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    return na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator, return_device=True)
for lag in sys.argv[1:]:
    os.environ["NEP_IAR_LAG"] = lag
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): lam, _, _ = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("LAG %s: %.1f ms/step, %d pairs" % (lag, dt * 1e3, len(lam)), flush=True)
