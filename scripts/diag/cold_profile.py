"""Where the FIRST iar call of a fresh process (config C2) spends its extra time: wall-clock wrappers around the set-up pieces
(no profiler: cProfile adds 15 ms to the call), calls 1..4"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd import linsolvers as ls, iar as iarmod, errmeasure as em, nep as nepmod
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t) * 1e3
    setattr(obj, name, g)
wrap(ls, "create_linsolver", "create_linsolver")
wrap(ls.DeviceLU, "__init__", "  DeviceLU.__init__")
wrap(ls._nep_hostlu, "factor", "    host factor (SuperLU)")
wrap(ls._DeviceRefactor, "maybe_start", "    plan thread start")
wrap(nepmod.SPMF_NEP, "compute_Mder", "  compute_Mder")
wrap(na.dense, "gemm_ts", "gemm_ts calls")
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev; torch.cuda.synchronize()
for call in (1, 2, 3, 4):
    acc.clear()
    t0 = time.perf_counter()
    lam, Q = bc.c2_device(na, nep, 100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"call": call, "ms": round((t2 - t0) * 1e3, 1), "host_return_ms": round((t1 - t0) * 1e3, 1),
                      **{k: round(v, 1) for k, v in acc.items()}}))
