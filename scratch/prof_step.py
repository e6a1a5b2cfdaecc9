import sys, time; sys.path.insert(0,'.')
import numpy as np, nep_amd as na, torch, cProfile, pstats
nep=na.nep_gallery("gun_spmf_scaled"); nep.dev
def step():
    lam,Q,V=na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, return_device=True)
    torch.cuda.synchronize()
step(); step()
pr=cProfile.Profile(); pr.enable(); t0=time.perf_counter(); step(); dt=time.perf_counter()-t0; pr.disable()
print("step %.1f ms"%(dt*1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
