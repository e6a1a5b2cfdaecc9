import sys, os, time, cProfile, pstats; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf"); nep.dev
kw=dict(sigma=250.0**2, radius=1e4, N=64, k=32, neigs=10**6, tol=1e-6, sanity_check=True)
for _ in range(2): lam,V=na.contour_beyn(nep,**kw)
torch.cuda.synchronize()
t=time.perf_counter(); lam,V=na.contour_beyn(nep,**kw); torch.cuda.synchronize(); print("plain %.1f ms pairs %d"%((time.perf_counter()-t)*1e3,len(lam)))
pr=cProfile.Profile(); pr.enable(); na.contour_beyn(nep,**kw); torch.cuda.synchronize(); pr.disable()
st=pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
