import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
_lib=sys.modules["nep_amd._lib"]; nepmod=sys.modules["nep_amd.nep"]; iarmod=sys.modules["nep_amd.iar"]
parts={}
class TimedLib:
    def __init__(self, lib): self._lib=lib; self._cache={}
    def __getattr__(self, n):
        f=getattr(self._lib,n)
        if n in self._cache: return self._cache[n]
        def g(*a):
            t=time.perf_counter(); r=f(*a); parts[n]=parts.get(n,0)+time.perf_counter()-t; return r
        self._cache[n]=g; return g
tl=TimedLib(_lib.lib)
for m in ("nep_amd.nep","nep_amd.dense","nep_amd.iar","nep_amd.linsolvers","nep_amd.errmeasure"):
    mod=sys.modules[m]
    if hasattr(mod,"lib"): mod.lib=tl
# also time stream_ptr and fut.result
orig_sp=nepmod.stream_ptr
def sp():
    t=time.perf_counter(); r=orig_sp(); parts["stream_ptr()"]=parts.get("stream_ptr()",0)+time.perf_counter()-t; return r
for m in ("nep_amd.nep","nep_amd.dense","nep_amd.iar","nep_amd.linsolvers"):
    mod=sys.modules[m]
    if hasattr(mod,"stream_ptr"): mod.stream_ptr=sp
from concurrent.futures import Future
orig_res=Future.result
def res(self,timeout=None):
    t=time.perf_counter(); r=orig_res(self,timeout); parts["Future.result"]=parts.get("Future.result",0)+time.perf_counter()-t; return r
Future.result=res
nep=na.nep_gallery("gun_spmf_scaled"); nep.dev
def step(): return na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, return_device=True)
step(); step()
for rep in range(2):
    parts.clear(); torch.cuda.synchronize(); t=time.perf_counter(); step(); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("step %.1f ms; "%(dt*1e3)+", ".join("%s %.1f"%(k,v*1e3) for k,v in sorted(parts.items(), key=lambda kv:-kv[1])[:12]))
