"""N calls of config C3 (nleigs R1 on gun) and nothing else -- for kernel statistics / traces"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import torch
import nep_amd as na
import baseline_configs as bc
nep = bc.c3_device_nep(na)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(N):
    t0 = time.perf_counter()
    lam = bc.c3_device(na, nep)[0]
    torch.cuda.synchronize()
    print("call %d: %.2f ms, %d pairs" % (i, (time.perf_counter() - t0) * 1e3, len(lam)), flush=True)
    if i == 0:
        from nep_amd.linsolvers import _DeviceRefactor
        _DeviceRefactor.wait()
