import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na, ctypes as C
from nep_amd._lib import lib, check, c_vp
T=time.perf_counter
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
A=nep.compute_Mder(0.0)
shape=(101, 9956*101)
mode=sys.argv[1]
for i in range(4):
    if mode=="lu": lu=na.DeviceLU(A, expected_solves=200)
    if mode=="alloc":
        p=C.c_void_p(); check(lib.nep_dev_alloc(C.byref(p), 300<<20)); check(lib.nep_dev_free(p))
    if mode=="hipmalloc":
        t=torch.empty(300<<20,dtype=torch.uint8,device="cuda"); del t; torch.cuda.empty_cache()
    torch.cuda.synchronize(); t0=T(); V=torch.empty(shape,dtype=torch.complex128,device="cuda"); torch.cuda.synchronize(); t1=T(); V.zero_(); torch.cuda.synchronize(); t2=T(); V.zero_(); torch.cuda.synchronize(); t3=T()
    print(mode,"empty %.2f ms zero_ %.2f again %.2f ptr %x"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3, V.data_ptr()))
    del V
    if mode=="lu": del lu
