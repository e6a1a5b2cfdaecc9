"""host-side profile (cProfile, by own time and cumulative) of the fourth of four C3 calls (nleigs R1 on gun): python scripts/diag/c3_cprofile.py"""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
nep = bc.c3_device_nep(na)
for i in range(3):
    bc.c3_device(na, nep); torch.cuda.synchronize()
    if i == 0:
        _DeviceRefactor.wait()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
lam = bc.c3_device(na, nep)[0]
torch.cuda.synchronize()
pr.disable()
print("call under cProfile %.1f ms, %d pairs" % ((time.perf_counter() - t0) * 1e3, len(lam)))
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
