import json, os, sys, time, resource
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import nep_amd as na, torch
import baseline_configs as bc
def cpu_s():
    r = resource.getrusage(resource.RUSAGE_SELF); return r.ru_utime + r.ru_stime
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
for _ in range(4): lam, Q = bc.c2_device(na, nep)
ts=[]; c0=cpu_s()
for _ in range(15):
    t0=time.perf_counter(); lam,Q=bc.c2_device(na,nep); ts.append((time.perf_counter()-t0)*1e3)
print(json.dumps({"env": {k: os.environ.get(k) for k in ("NEP_IAR_EIG","NEP_IAR_EIG_STREAMS","GPU_MAX_HW_QUEUES")}, "ms_median": float(np.median(ts)), "ms_min": float(min(ts)), "cpu_s_per_call": (cpu_s()-c0)/15, "pairs": len(lam)}))
os.environ["NEP_IAR_TRACE"]="1"; bc.c2_device(na, nep)
