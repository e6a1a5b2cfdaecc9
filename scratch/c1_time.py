import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na, cProfile, pstats
nep=na.nep_gallery("dep0")
for i in range(3):
    t=time.perf_counter(); lam,v=na.resinv(nep,lam=0,v=np.ones(5)); print("resinv %.1f ms"%((time.perf_counter()-t)*1e3), lam)
pr=cProfile.Profile(); pr.enable(); na.resinv(nep,lam=0,v=np.ones(5)); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
