"""N calls of config C2 (gun SPMF iar m=100) and nothing else -- the command behind the per-dispatch kernel traces
(rocprofv3 --kernel-trace -- python scripts/iar_runs.py 6)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
import contextlib
hp = torch.cuda.Stream(priority=-1) if os.environ.get("IAR_RUNS_HIGH_PRIO") else None      # experiment: the caller's stream at high priority
for i in range(N):
    t0 = time.perf_counter()
    with (torch.cuda.stream(hp) if hp is not None else contextlib.nullcontext()):
        lam, Q = bc.c2_device(na, nep, 100)
    torch.cuda.synchronize()
    print("call %d: %.2f ms, %d pairs" % (i, (time.perf_counter() - t0) * 1e3, len(lam)), flush=True)
    if i == 1:
        _DeviceRefactor.wait()
