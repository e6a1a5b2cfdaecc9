import sys, time; sys.path.insert(0,'.')
import numpy as np, nep_amd as na, torch
n=9956; m=100
V=torch.randn((m+1, n*(m+1)), dtype=torch.float64, device='cuda').to(torch.complex128)
rng=np.random.default_rng(0)
for trial in range(2):
    ts=[]
    for k in range(1,m+1):
        Z=rng.standard_normal((k,k))+1j*rng.standard_normal((k,k))
        t=time.perf_counter()
        QT=na.gemm_ts(V, Z, rowmajor=True, k=k, rows=n, ldz=n*(m+1))
        ts.append(time.perf_counter()-t)
        torch.cuda.synchronize()
    ts=np.array(ts)*1e3
    print("trial",trial,"sum %.1f ms  k=10: %.3f k=50: %.3f k=100: %.3f max %.3f"%(ts.sum(),ts[9],ts[49],ts[99],ts.max()))
    # breakdown at k=100: torch.empty vs ctypes
    Z=rng.standard_normal((100,100))+1j*rng.standard_normal((100,100))
    out=torch.empty((n,100),dtype=torch.complex128,device='cuda')
    t=time.perf_counter()
    for _ in range(20): na.gemm_ts(V, Z, rowmajor=True, k=100, rows=n, ldz=n*(m+1), out=out)
    t1=(time.perf_counter()-t)/20; torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(20): torch.empty((n,100),dtype=torch.complex128,device='cuda')
    t2=(time.perf_counter()-t)/20
    print("  k=100 with out= (no alloc): %.3f ms/call host ; torch.empty alone %.3f ms"%(t1*1e3,t2*1e3))
