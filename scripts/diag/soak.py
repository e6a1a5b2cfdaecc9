"""soak: C2 / C3 / C4 interleaved for a while; every result must match the first one of its kind"""
import os, sys, time
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nep2 = na.nep_gallery("gun_spmf_scaled"); nep2.dev
nep4 = na.nep_gallery("gun_spmf"); nep4.dev
nep3 = bc.c3_device_nep(na)
Vh = na.probe_block(nep4.n, 32)
na.HostLUPool.warm(8)
ref = {}
cnt = {"c2": 0, "c3": 0, "c4": 0}
bad = 0
t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < secs:
    kind = ("c2", "c2", "c3", "c2", "c4")[it % 5]; it += 1
    try:
        if kind == "c2":
            lam, Q = bc.c2_device(na, nep2, 100)
        elif kind == "c3":
            lam = bc.c3_device(na, nep3)[0]
        else:
            lam, V = bc.c4_device(na, nep4, Vh=Vh)
    except Exception as e:
        bad += 1; print("EXC", kind, repr(e)[:200], flush=True); continue
    lam = np.sort_complex(np.asarray(lam)); cnt[kind] += 1
    if kind not in ref:
        ref[kind] = lam; _DeviceRefactor.wait()
    elif len(lam) != len(ref[kind]) or np.abs(lam - ref[kind]).max() > 1e-8 * np.abs(ref[kind]).max():
        bad += 1; print("MISMATCH", kind, len(lam), len(ref[kind]), flush=True)
print("runs", cnt, "bad", bad, "plans", [(p["state"], p["uses"], p["fails"]) for p in _DeviceRefactor.plans.values()],
      "mem MiB", (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) // 2**20)
