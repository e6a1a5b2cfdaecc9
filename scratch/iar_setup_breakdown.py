import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, scipy.sparse as sp, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
T=time.perf_counter
for i in range(4):
    torch.cuda.synchronize()
    t0=T(); A=nep.compute_Mder(0.0); t1=T(); lu=na.DeviceLU(A, expected_solves=200); t2=T(); torch.cuda.synchronize(); t3=T()
    print("Mder %.1f | DeviceLU %.1f (factor %.1f, t_total? create %.1f convert %.1f) sync %.1f"%((t1-t0)*1e3,(t2-t1)*1e3,lu.t_factor*1e3,lu.t_create*1e3,lu.t_convert*1e3,(t3-t2)*1e3))
    t0=T(); V=torch.zeros((101, 9956*101), dtype=torch.complex128, device="cuda"); torch.cuda.synchronize(); t1=T(); x=torch.from_numpy(np.ones(9956)+0j).to("cuda"); t2=T(); y=torch.from_numpy(np.arange(101)).to("cuda"); t3=T()
    tab=nep.derivative_table(0.0,100,rowscale=np.ones(100)); t4=T()
    print("   zeros %.2f to1 %.2f to2 %.2f table %.2f"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3))
    del V; ta=T(); del lu; print('   del lu %.2f ms'%((T()-ta)*1e3))
