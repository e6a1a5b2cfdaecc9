import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
nx,nz=int(sys.argv[1]),int(sys.argv[2])
nep=na.nep_gallery("WEP",nx=nx,nz=nz,benchmark_problem="JARLEBRING"); n=nep.n
v0=np.ones(n)/np.sqrt(n)
tm={}
eh=[]
t=time.perf_counter()
lam,Q=na.tiar(nep,sigma=-3-3.5j,gamma=1.0,maxit=60,neigs=np.inf,v=v0,tol=1e-8,timers=tm,errhist=eh)[:2]
print("pairs",len(lam),"time %.1f"%(time.perf_counter()-t), {k:round(v,2) for k,v in tm.items()})
for it in (19,29,39,49,59):
    if it < len(eh): print(it+1, np.sort(eh[it])[:9])
