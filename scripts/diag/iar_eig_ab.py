"""A/B of the headline iar call with eig(H_k) on the device (csrc/hesseig.hip) against LAPACK on host worker threads
(NEP_IAR_EIG=host): wall per call, CPU seconds per call (getrusage, all threads), eigenpairs, eigenvalue agreement.
    python scripts/diag/iar_eig_ab.py [calls]"""
import json
import os
import resource
import sys
import time

os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nep_amd as na
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import baseline_configs as bc


def cpu_s():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def batch(nep, calls, mode):
    os.environ["NEP_IAR_EIG"] = mode
    for _ in range(3):
        lam, Q = bc.c2_device(na, nep)
    torch.cuda.synchronize()
    ts = []; c0 = cpu_s()
    for _ in range(calls):
        t0 = time.perf_counter()
        lam, Q = bc.c2_device(na, nep)
        ts.append((time.perf_counter() - t0) * 1e3)
    c1 = cpu_s()
    return lam, {"mode": mode, "ms_median": float(np.median(ts)), "ms_min": float(np.min(ts)), "ms_max": float(np.max(ts)),
                 "cpu_s_per_call": (c1 - c0) / calls, "eigenpairs": int(len(lam)), "fallbacks": na.iar.dev_eig_fallbacks}


if __name__ == "__main__":
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
    out = []
    ref = None
    for mode in ("host", "dev", "host", "dev"):
        lam, r = batch(nep, calls, mode)
        if ref is None:
            ref = lam
        ok, worst = bc.match(lam, ref, 1e-8)
        r["match_first_batch_1e-8"] = bool(ok); r["max_rel_diff"] = worst
        print(json.dumps(r), flush=True)
    if os.environ.get("NEP_IAR_TRACE_ONCE"):
        os.environ["NEP_IAR_TRACE"] = "1"
        bc.c2_device(na, nep)
