import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from importlib import import_module
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nonlineareigenproblems.jl_amd"))
import _nep_hostlu as hl
import scipy.sparse as sp
nx,nz=int(sys.argv[1]),int(sys.argv[2])
# 5-point-like stand-in with the same pattern family as the WEP interior (timing of the ordering/fill only)
n=nx*nz
T=lambda m: sp.diags([np.ones(m-1),-2*np.ones(m),np.ones(m-1)],[-1,0,1])
A=(sp.kron(sp.eye(nz),T(nx))+sp.kron(T(nz),sp.eye(nx))).tocsc().astype(np.complex128)
A=A+sp.diags((-3-3.5j)**2*0.01*np.ones(n)+0.3j)
A=A.tocsc()
t=time.perf_counter(); f=hl.factor(A.data,A.indices,A.indptr,A.shape); dt=time.perf_counter()-t
print("n",n,"factor s",round(dt,2),"total",round(f["t_total"],2),f["strategy"],"nnzL",len(f["Lx"]),"nnzU",len(f["Ux"]))
