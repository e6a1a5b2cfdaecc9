import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
from nep_amd._lib import lib, check, c_vp
from nep_amd.nep import stream_ptr
T=time.perf_counter
shape=(101, 9956*101)
for i in range(4):
    torch.cuda.synchronize(); t0=T(); V=torch.zeros(shape,dtype=torch.complex128,device="cuda"); torch.cuda.synchronize(); t1=T()
    W=torch.empty(shape,dtype=torch.complex128,device="cuda"); torch.cuda.synchronize(); t2=T()
    check(lib.nep_dev_memset(c_vp(W.data_ptr()), 0, W.numel()*16, stream_ptr())); torch.cuda.synchronize(); t3=T()
    W2=torch.empty(shape[0]*shape[1]*2,dtype=torch.float64,device="cuda"); torch.cuda.synchronize(); t4=T(); W2.zero_(); torch.cuda.synchronize(); t5=T()
    print("torch.zeros c128 %.2f ms | empty %.2f | nep_dev_memset %.2f | f64 zero_ %.2f"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t5-t4)*1e3))
    del V,W,W2
