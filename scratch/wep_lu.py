import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
nx,nz=(int(sys.argv[1]),int(sys.argv[2])) if len(sys.argv)>2 else (303,299)
nep=na.nep_gallery("WEP",nx=nx,nz=nz,benchmark_problem="JARLEBRING"); n=nep.n
t=time.perf_counter(); A=nep.compute_Mder(-3-3.5j); print("Mder %.2f s nnz %d"%(time.perf_counter()-t, A.nnz))
for spec in (None,):
    t=time.perf_counter(); lu=na.DeviceLU(A, permc_spec=spec, expected_solves=60); torch.cuda.synchronize(); dt=time.perf_counter()-t
    b=torch.randn(n,dtype=torch.float64,device='cuda').to(torch.complex128)
    x=lu.solve(b); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): x=lu.solve(b)
    torch.cuda.synchronize(); ts=(time.perf_counter()-t)/5
    xh=na.to_host(x.reshape(1,-1))[:,0]; bh=na.to_host(b.reshape(1,-1))[:,0]
    print("spec %s strategy %s: setup %.2f s (factor %.2f create %.2f) nnzL %d nnzU %d tail %d mid %d/%d levels %d/%d (full %d/%d) segs w%d n%d launches %d solve %.2f ms resid %.1e"%(spec,lu.strategy,dt,lu.t_factor,lu.t_create,lu.nnzL,lu.nnzU,lu.tail,lu.mid_rows,lu.mid_block,lu.levL,lu.levU,lu.levL_full,lu.levU_full,lu.wide_segments,lu.narrow_segments,lu.launches_last_solve(),ts*1e3,np.linalg.norm(A@xh-bh)/np.linalg.norm(bh)))
    del lu
