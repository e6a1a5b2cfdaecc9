import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
from nep_amd._lib import lib, check, c_vp
from nep_amd.nep import stream_ptr
T=time.perf_counter
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
A=nep.compute_Mder(0.0)
shape=(101, 9956*101)
mode=sys.argv[1]
for i in range(5):
    if mode in("lu","lu_keep"): lu=na.DeviceLU(A, expected_solves=200)
    if mode=="hostalloc": junk=[np.ones(3_000_000) for _ in range(8)]
    torch.cuda.synchronize(); t0=T(); V=torch.zeros(shape,dtype=torch.complex128,device="cuda"); torch.cuda.synchronize(); t1=T()
    print(mode,"torch.zeros c128 %.2f ms  reserved %.2f GB"%((t1-t0)*1e3, torch.cuda.memory_reserved()/1e9))
    del V
    if mode=="lu": del lu
