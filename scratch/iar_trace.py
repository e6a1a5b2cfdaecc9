import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
iarmod=sys.modules["nep_amd.iar"]; dense=sys.modules["nep_amd.dense"]; ls=sys.modules["nep_amd.linsolvers"]
acc={}
def wrap(mod, name, key):
    f=getattr(mod,name)
    def g(*a,**k):
        t=time.perf_counter(); r=f(*a,**k); acc[key]=acc.get(key,0)+time.perf_counter()-t; return r
    setattr(mod,name,g)
wrap(iarmod,"create_linsolver","create_linsolver")
wrap(dense,"orthogonalize_and_normalize","orth(sync)")
wrap(dense,"gemm_ts","gemm_ts call")
wrap(iarmod,"estimate_errors","resid(sync)")
wrap(ls.FactorizeLinSolver,"solve_dev","solve_dev(sync in refine check)")
nep=na.nep_gallery("gun_spmf_scaled"); nep.dev
orig=nep.lincomb_rowscale
def lr(*a,**k):
    t=time.perf_counter(); r=orig(*a,**k); acc["mlincomb call"]=acc.get("mlincomb call",0)+time.perf_counter()-t; return r
nep.lincomb_rowscale=lr
def step():
    return na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, return_device=True)
step(); step()
for rep in range(3):
    acc.clear(); torch.cuda.synchronize(); t=time.perf_counter(); step(); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("step %.1f ms: "%(dt*1e3) + ", ".join("%s %.1f"%(k,v*1e3) for k,v in acc.items()) + ", other %.1f"%((dt-sum(acc.values()))*1e3))

# ---- finer split of gemm_ts inside the iar loop
import ctypes
_lib=sys.modules["nep_amd._lib"]
orig_gemm=_lib.lib.nep_gemm_ts
parts={}
def timed_as(a, order="F"):
    t=time.perf_counter(); r=orig_as(a,order); parts["as_c128"]=parts.get("as_c128",0)+time.perf_counter()-t; return r
orig_as=_lib.as_c128; _lib.as_c128=timed_as
class W:
    def __call__(self,*a):
        t=time.perf_counter(); r=orig_gemm(*a); parts["lib.nep_gemm_ts"]=parts.get("lib.nep_gemm_ts",0)+time.perf_counter()-t; return r
dense.lib=type("L",(),{"__getattr__":lambda self,n: (W() if n=="nep_gemm_ts" else getattr(_lib.lib,n))})()
orig_empty=torch.empty
def timed_empty(*a,**k):
    t=time.perf_counter(); r=orig_empty(*a,**k); parts["torch.empty"]=parts.get("torch.empty",0)+time.perf_counter()-t; return r
dense.torch.empty=timed_empty
acc.clear(); parts.clear(); step(); print({k:round(v*1e3,1) for k,v in parts.items()}, {k:round(v*1e3,1) for k,v in acc.items()})
