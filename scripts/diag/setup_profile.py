"""host-side breakdown of create_linsolver + the whole iar step on the gun problem (config C2)"""
import os, sys, time, cProfile, pstats, io
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled", 9956); nep.dev
def step():
    creator = na.FactorizeLinSolverCreator(max_factorizations=0)
    return na.iar(nep, sigma=0.0, gamma=1.0, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator, return_device=True)
for _ in range(3): step()
torch.cuda.synchronize()
# setup only
for rep in range(3):
    t0 = time.perf_counter()
    A = nep.compute_Mder(0.0); t1 = time.perf_counter()
    lu = na.DeviceLU(A); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("compute_Mder %.2f ms | DeviceLU %.2f ms (splu %.2f, create %.2f) | device build wait %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, lu.t_factor*1e3, lu.t_create*1e3, (t3-t2)*1e3))
pr = cProfile.Profile(); pr.enable()
t0=time.perf_counter(); step(); torch.cuda.synchronize(); dt=time.perf_counter()-t0
pr.disable()
print("step %.1f ms" % (dt*1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
