"""The headline call under a TWO-CPU budget: what a replica gets when 8 ranks share a 16-CPU quota.
A fresh process restricted to two CPUs of the GPU's NUMA node (sched_setaffinity before anything else starts a thread):
    python scripts/cpu2_call.py [calls] [ncpus]   -> one JSON line (wall per call, CPU seconds per call, eigenpairs)"""
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ncpu = int(sys.argv[2]) if len(sys.argv) > 2 else 2
try:
    cur = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, set(cur[:ncpu]))
except Exception:
    pass
os.environ["NEP_NO_PIN"] = "1"                      # keep the two CPUs chosen above
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
from nep_amd._affinity import cpu_budget


def cpu_s():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.cuda.set_device(0)
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
for _ in range(2):
    bc.c2_device(na, nep, 100)
_DeviceRefactor.wait()
for _ in range(3):
    lam, Q = bc.c2_device(na, nep, 100)
torch.cuda.synchronize()
ts = []; c0 = cpu_s()
for _ in range(calls):
    t0 = time.perf_counter(); lam, Q = bc.c2_device(na, nep, 100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
c1 = cpu_s()
print(json.dumps({"cpus": len(os.sched_getaffinity(0)), "cpu_budget": cpu_budget(), "calls": calls, "ms_per_call_mean": float(np.mean(ts)),
                  "ms_per_call_median": float(np.median(ts)), "ms_per_call_max": float(np.max(ts)), "cpu_s_per_call": (c1 - c0) / calls,
                  "eigenpairs": int(len(lam)), "eig": os.environ.get("NEP_IAR_EIG", "dev"), "dev_eig_fallbacks": na.iar.dev_eig_fallbacks}))
