import numpy as np, time, os
from threadpoolctl import threadpool_limits, threadpool_info
print([ (p['internal_api'],p['num_threads']) for p in threadpool_info()], os.cpu_count())
rng=np.random.default_rng(0)
for k in (25,50,100):
    H=np.triu(rng.standard_normal((k,k))+1j*rng.standard_normal((k,k)),-1)
    for nt in (1,2,4,16,None):
        with threadpool_limits(limits=nt):
            np.linalg.eig(H)
            t=time.perf_counter()
            for _ in range(10): np.linalg.eig(H)
            dt=(time.perf_counter()-t)/10
        print(k, nt, "%.2f ms"%(dt*1e3))
import scipy.linalg as sla
H=np.triu(rng.standard_normal((100,100))+1j*rng.standard_normal((100,100)),-1)
with threadpool_limits(limits=1):
    t=time.perf_counter()
    for _ in range(10): sla.eig(H)
    print("scipy eig 1 thread %.2f ms"%((time.perf_counter()-t)/10*1e3))
    from scipy.linalg import lapack
    t=time.perf_counter()
    for _ in range(10):
        Hh=H.copy()
        w,z,info=lapack.zhseqr(Hh, compute_q=0) if False else (None,None,0)
    # hseqr + trevc path
    t=time.perf_counter()
    for _ in range(10):
        T_,Q_,info = lapack.zhseqr(H.copy(), z=np.eye(100,dtype=complex), job='S', compz='V') if hasattr(lapack,'zhseqr') else (None,None,-1)
    print("zhseqr available", hasattr(lapack,'zhseqr'), "%.2f ms"%((time.perf_counter()-t)/10*1e3))
