import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
from nep_amd import nep_amd_hostlu as _nep_hostlu
import scipy.sparse as sp
if sys.argv[1]=="gun":
    nep=na.nep_gallery("gun_spmf_scaled"); n=nep.n
    A=sp.csc_matrix(nep.compute_Mder(0.0),dtype=np.complex128)
else:
    nx,nz=(int(sys.argv[1]),int(sys.argv[2]))
    nep=na.nep_gallery("WEP",nx=nx,nz=nz,benchmark_problem="JARLEBRING"); n=nep.n
    A=sp.csc_matrix(nep.compute_Mder(-3-3.5j),dtype=np.complex128)
F=_nep_hostlu.factor(A.data,A.indices,A.indptr,A.shape)
print("factor %.2f s"%F["t_factor"])
b=torch.randn(n,dtype=torch.float64,device='cuda').to(torch.complex128)
bh=na.to_host(b.reshape(1,-1))[:,0]
for cfg in sys.argv[3:]:
    for kv in cfg.split(","):
        k,v=kv.split("="); 
        if v=="": os.environ.pop(k,None)
        else: os.environ[k]=v
    t=time.perf_counter(); lu=na.DeviceLU(factors=F, expected_solves=180); torch.cuda.synchronize(); dt=time.perf_counter()-t
    x=lu.solve(b); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): x=lu.solve(b)
    torch.cuda.synchronize(); ts=(time.perf_counter()-t)/50
    xh=na.to_host(x.reshape(1,-1))[:,0]
    print("%-50s create %.2f s tail %d mid %d/%d head levels %d launches %d solve %.3f ms resid %.1e"%(cfg,dt,lu.tail,lu.mid_rows,lu.mid_block,lu.levL,lu.launches_last_solve(),ts*1e3,np.linalg.norm(A@xh-bh)/np.linalg.norm(bh)),flush=True)
    del lu
