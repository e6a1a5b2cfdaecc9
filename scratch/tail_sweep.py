import sys, time, os; sys.path.insert(0,'.')
import numpy as np, nep_amd as na, torch
nep=na.nep_gallery("gun_spmf_scaled"); nep.dev
A=nep.compute_Mder(0.0)
for T in (0, 600, 1067, 1500, 2000, 2500, 3000):
    os.environ["NEP_LU_TAIL"]=str(T)
    t1=time.perf_counter(); lu=na.DeviceLU(A); torch.cuda.synchronize(); t2=time.perf_counter()
    b=torch.randn(9956,dtype=torch.float64,device='cuda').to(torch.complex128)
    x=lu.solve(b); torch.cuda.synchronize()
    t3=time.perf_counter()
    for _ in range(30): x=lu.solve(b)
    torch.cuda.synchronize()
    ts=(time.perf_counter()-t3)/30*1e3
    r=na.to_host(x.reshape(1,-1))[:,0]; bb=na.to_host(b.reshape(1,-1))[:,0]
    print("T=%d tail=%d setup-minus-factor %.1f ms solve %.3f ms launches %d levels %d/%d segs %d/%d resid %.1e"%(T,lu.tail,(t2-t1-lu.t_factor)*1e3,ts,lu.launches_last_solve(),lu.levL,lu.levU,lu.wide_segments,lu.narrow_segments,np.linalg.norm(A@r-bb)/np.linalg.norm(bb)))
