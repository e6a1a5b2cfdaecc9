import sys; sys.path.insert(0,'.')
import numpy as np, nep_amd as na, torch
from oracle import gallery as og, solvers as osol
nep=na.nep_gallery("dep0"); onep=og.dep0()
v=np.ones(5)/np.sqrt(5); c=np.ones(5)
print("rf oracle", osol.compute_rf(onep, v.astype(complex), y=c.astype(complex), lam=0j, target=0j))
print("rf gpu   ", na.compute_rf(nep, v, y=c, lam=0j, target=0j))
z1=nep.compute_Mlincomb(0.3, v); z2=nep.compute_Mlincomb(0.3, v, [1.0], 1)
print(z1, onep.compute_Mlincomb(0.3,v)); print(z2, onep.compute_Mlincomb(0.3,v,[1.0],1))
from nep_amd.newton import _dots2
Z2=na.to_dev(np.column_stack([z1,z2])); yd=na.to_dev(c)[0]
print(_dots2(yd,Z2,5), np.vdot(c,z1), np.vdot(c,z2))
hg=[];ho=[]
try: na.resinv(nep,lam=0,v=np.ones(5),hist=hg,maxit=3)
except Exception as e: pass
try: osol.resinv(onep,lam=0,v=np.ones(5),hist=ho,maxit=3)
except Exception as e: pass
print(hg); print(ho)
