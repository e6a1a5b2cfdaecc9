"""What a FIRST call costs: a fresh process, config C2 (gun SPMF iar m=100), wall time of call 1, 2, ... with
the one-off pieces named.  `python scripts/cold_call.py [ncalls]` prints one JSON line."""
import json
import os
import sys
import time

t00 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import numpy as np
import torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor

t_import = time.perf_counter() - t00
torch.cuda.set_device(0)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t_ctx = time.perf_counter() - t00 - t_import
t0 = time.perf_counter()
nep = na.nep_gallery("gun_spmf_scaled")
t_nep = time.perf_counter() - t0
t0 = time.perf_counter(); nep.dev; torch.cuda.synchronize(); t_upload = time.perf_counter() - t0
ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
calls = []; states = []; pairs = []
for i in range(ncalls):
    t0 = time.perf_counter()
    lam, Q = bc.c2_device(na, nep, 100)
    torch.cuda.synchronize()
    calls.append(round((time.perf_counter() - t0) * 1e3, 2)); pairs.append(int(len(lam)))
    states.append([p["state"] for p in _DeviceRefactor.plans.values()])
print(json.dumps({"import_s": round(t_import, 3), "hip_context_s": round(t_ctx, 3), "nep_build_s": round(t_nep, 3),
                  "upload_s": round(t_upload, 4), "calls_ms": calls, "eigenpairs": pairs, "plan_state_after_call": states,
                  "eigenpairs_per_s_cold": round(pairs[0] / (calls[0] * 1e-3), 1),
                  "eigenpairs_per_s_steady": round(pairs[-1] / (min(calls[-2:]) * 1e-3), 1)}))
