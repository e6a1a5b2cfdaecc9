"""device memory in use over many iar calls (device LU path): should plateau"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled", 9956); nep.dev
def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
N = int(sys.argv[1]) if len(sys.argv) > 1 else 61
for i in range(N):
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    lam, Q, V = na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator, return_device=True)
    del Q, V
    if i % (10 if N <= 100 else 50) == 0:
        import psutil
        torch.cuda.synchronize(); print("call %d: %.0f MiB in use, host RSS %.0f MiB, %d pairs" % (i, used(), psutil.Process().memory_info().rss / 2**20, len(lam)), flush=True)
import threading
print("threads alive:", threading.active_count())
