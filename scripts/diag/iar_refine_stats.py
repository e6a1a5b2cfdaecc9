import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
from nep_amd import linsolvers
made=[]
orig=linsolvers.create_linsolver
def cl(c,n,l):
    s=orig(c,n,l); made.append(s); return s
linsolvers.create_linsolver=cl
iarm=sys.modules[na.iar.__module__]
iarm.create_linsolver=cl
R=int(sys.argv[1]) if len(sys.argv)>1 else 10
for rep in range(4):
    tm={}
    t=time.perf_counter()
    lam,Q=na.iar(nep,maxit=100,neigs=np.inf,v=np.ones(nep.n),tol=1e-10,timers=tm,linsolvercreator=na.FactorizeLinSolverCreator(umfpack_refinements=R))[:2]
    dt=time.perf_counter()-t
    s=made[-1]
    print("pairs",len(lam),"time %.1f ms"%(dt*1e3),{k:round(v*1e3,1) for k,v in tm.items()},"solves",s.solves,"checks",s.refine_checks,"steps",s.refine_steps_taken,"plan",s._plan,"omega",s.last_omega)
