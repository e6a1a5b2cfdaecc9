import sys, os, time, importlib; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nep_amd
from concurrent.futures import ThreadPoolExecutor
he=importlib.import_module("nep_amd._hosteig"); hl=importlib.import_module("_nep_hostlu")
rng=np.random.default_rng(0)
Hs=[np.triu(rng.standard_normal((k,k))+1j*rng.standard_normal((k,k)),-1) for k in range(1,101)]
ctl=hl.blas_controller()
for lim in (None,1):
    for fn,name in ((he.eig,"ctypes zgeev"),(np.linalg.eig,"numpy eig")):
        for nw in (1,4):
            def run():
                t=time.perf_counter()
                with ThreadPoolExecutor(nw) as ex: list(ex.map(fn,Hs))
                return time.perf_counter()-t
            if lim is None: dt=run()
            else:
                with ctl.limit(limits=lim): dt=run()
            print("blas limit %s %-13s workers %d: %.1f ms"%(lim,name,nw,dt*1e3))
