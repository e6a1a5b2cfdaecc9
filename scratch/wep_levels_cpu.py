import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nonlineareigenproblems.jl_amd"))
import numpy as np, scipy.sparse as sp
import _nep_hostlu as hl
nx,nz=int(sys.argv[1]),int(sys.argv[2])
n=nx*nz
T=lambda m: sp.diags([np.ones(m-1),-2*np.ones(m),np.ones(m-1)],[-1,0,1])
A=(sp.kron(sp.eye(nz),T(nx))+sp.kron(T(nz),sp.eye(nx))).tocsc().astype(np.complex128)
A=(A+sp.diags((-3-3.5j)**2*0.01*np.ones(n)+0.3j)).tocsc()
f=hl.factor(A.data,A.indices,A.indptr,A.shape)
print("factor",round(f["t_factor"],2),"nnzL",len(f["Lx"]))
Lp,Li=f["Lp"],f["Li"]
lev=np.zeros(n,dtype=np.int64)
for i in range(n):
    c=Li[Lp[i]:Lp[i+1]-1] if Li[Lp[i+1]-1]==i else Li[Lp[i]:Lp[i+1]]
    c=c[c<i]
    if len(c): lev[i]=lev[c].max()+1
nl=lev.max()+1
w=np.bincount(lev)
rl=np.diff(Lp)
print("levels",nl)
# rows (in index order): level of row vs index: how many of the last T rows cover levels beyond X
for T_ in (4096,8192,16384,32768,65536):
    head=lev[:n-T_]
    print("tail",T_,"head levels",head.max()+1,"nnz in tail rows",rl[n-T_:].sum(),"frac",rl[n-T_:].sum()/rl.sum())
# width histogram from the end
cum=np.cumsum(w[::-1])
for k in (100,500,1000,2000,4000,6000):
    if k<nl: print("last",k,"levels hold",cum[k-1],"rows")
np.save("/tmp/lev.npy",lev)
