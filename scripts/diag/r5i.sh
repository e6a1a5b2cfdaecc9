out=gpurun_out/r5i; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gauss_jordan" > $out/pytest_gj.log 2>&1; tail -3 $out/pytest_gj.log
python - <<'PY'
import time, ctypes as C, numpy as np, torch, nep_amd as na
from nep_amd._lib import lib, check, c_vp
n=1517
M=torch.randn((n,n),dtype=torch.float64,device="cuda").to(torch.complex128)*0.1
out=torch.empty_like(M); work=torch.empty(2*n+2,dtype=torch.complex128,device="cuda"); info=C.c_int32(0)
for _ in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    check(lib.nep_zinv_h_dev(n,c_vp(M.data_ptr()),n,1.0,c_vp(out.data_ptr()),n,c_vp(work.data_ptr()),C.byref(info),None))
    print("nep_zinv_h_dev n=1517: %.1f ms info %d"%((time.perf_counter()-t)*1e3, info.value))
Mh=M.cpu().numpy().T+np.eye(n); t=time.perf_counter(); R=np.linalg.inv(Mh); print("numpy inv %.1f ms"%((time.perf_counter()-t)*1e3))
print("rel diff", np.linalg.norm(out.cpu().numpy().T-R.conj().T)/np.linalg.norm(R))
PY
