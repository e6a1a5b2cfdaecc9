import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
A = nep.compute_Mder(0.0)
for i in range(4):
    t=time.perf_counter(); lu=na.DeviceLU(A, expected_solves=200); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("rep",i,"total %.1f ms factor %.1f create %.1f"%(dt*1e3, lu.t_factor*1e3, lu.t_create*1e3), flush=True)
    del lu
