import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
for rep in range(6):
    if rep >= 3: os.environ["NEP_IAR_TRACE"] = "1"
    torch.cuda.synchronize(); t = time.perf_counter()
    lam, Q = na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)[:2]
    torch.cuda.synchronize(); print("run %d: %.1f ms, %d pairs" % (rep, (time.perf_counter() - t) * 1e3, len(lam)), flush=True)
