import sys, os, time, cProfile, pstats; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
kw=dict(maxit=100,neigs=np.inf,v=np.ones(nep.n),tol=1e-10)
for _ in range(2): na.iar(nep,**kw)
torch.cuda.synchronize()
t=time.perf_counter(); na.iar(nep,**kw); torch.cuda.synchronize(); print("plain %.1f ms"%((time.perf_counter()-t)*1e3))
pr=cProfile.Profile(); pr.enable(); na.iar(nep,**kw); torch.cuda.synchronize(); pr.disable()
st=pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
