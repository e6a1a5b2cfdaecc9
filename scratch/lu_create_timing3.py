import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na, scipy.sparse as sp
from nep_amd import _nep_hostlu
nep = na.nep_gallery("gun_spmf"); nep.dev
A = sp.csc_matrix(nep.compute_Mder(250.0**2+1e4), dtype=np.complex128)
F=_nep_hostlu.factor(A.data,A.indices,A.indptr,A.shape)
B=torch.randn((32,nep.n),dtype=torch.float64,device="cuda").to(torch.complex128)
for i in range(4):
    t=time.perf_counter(); lu=na.DeviceLU(factors=F, expected_solves=1); t1=time.perf_counter(); X=lu.solve(B); torch.cuda.synchronize(); t2=time.perf_counter()
    print("rep",i,"create %.2f ms (lib %.2f) solve32 %.2f ms tail %d mid %d levels %d launches %d"%((t1-t)*1e3, lu.t_create*1e3,(t2-t1)*1e3,lu.tail,lu.mid_rows,lu.levL,lu.launches_last_solve()), flush=True)
    del lu
