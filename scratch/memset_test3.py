import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na, ctypes as C
from nep_amd._lib import lib, check, c_vp
from nep_amd.nep import stream_ptr
T=time.perf_counter
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
A=nep.compute_Mder(0.0)
shape=(101, 9956*101)
mode=sys.argv[1]
V=torch.zeros(shape,dtype=torch.complex128,device="cuda"); torch.cuda.synchronize()
for i in range(4):
    if mode=="lu": lu=na.DeviceLU(A, expected_solves=200)
    if mode=="alloc":
        p=C.c_void_p(); check(lib.nep_dev_alloc(C.byref(p), 300<<20)); check(lib.nep_dev_free(p))
    if mode=="bigh2d":
        x=np.ones(2_000_000)+0j; xd=na.to_dev(x)
    if mode=="torchh2d":
        x=torch.from_numpy(np.ones(2_000_000)).to("cuda")
    torch.cuda.synchronize(); t0=T(); V.zero_(); torch.cuda.synchronize(); t1=T(); V.zero_(); torch.cuda.synchronize(); t2=T()
    W=torch.empty(1<<26,dtype=torch.float64,device="cuda"); torch.cuda.synchronize(); t3=T(); W.zero_(); torch.cuda.synchronize(); t4=T()
    print(mode,"V.zero_ first %.2f ms second %.2f ms | other 512MB block zero %.2f"%((t1-t0)*1e3,(t2-t1)*1e3,(t4-t3)*1e3))
    if mode=="lu": del lu
