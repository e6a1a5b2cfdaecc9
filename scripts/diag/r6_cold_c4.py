"""config C4 in a FRESH process: what the first call on a sparsity pattern costs (NEP_BEYN_COLD_PLAN=0: all nodes through the host pool)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import nep_amd as na
import baseline_configs as bc
nep = na.nep_gallery("gun_spmf"); nep.dev
Vh = na.probe_block(nep.n, 32)
torch.cuda.synchronize()
out = []
for i in range(4):
    t0 = time.perf_counter()
    lam, V = bc.c4_device(na, nep, None, Vh=Vh)
    torch.cuda.synchronize()
    out.append(((time.perf_counter() - t0) * 1e3, np.sort_complex(np.asarray(lam))))
print("calls ms:", [round(o[0], 1) for o in out], "pairs", [len(o[1]) for o in out],
      "max |lam(call 0) - lam(call 3)| %.2e" % (np.abs(out[0][1] - out[3][1]).max() if len(out[0][1]) == len(out[3][1]) else -1),
      "NEP_BEYN_COLD_PLAN=%s" % os.environ.get("NEP_BEYN_COLD_PLAN", "1"))
pi = {"phases_s": {}}
bc.c4_device(na, nep, None, Vh=Vh, info=pi); torch.cuda.synchronize()
print("phases ms:", {k: round(v * 1e3, 2) for k, v in pi["phases_s"].items()})
from nep_amd.linsolvers import _DeviceRefactor
for k_, p_ in _DeviceRefactor.plans.items():
    out_ = (na._lib.c_i64 * 6)(); na._lib.lib.nep_lu_refac_info(p_["handle"], out_); w_ = (na._lib.c_i64 * 5)(); na._lib.lib.nep_lu_refac_wide_info(p_["handle"], w_)
    print("plan:", p_["state"], "info", list(out_), "wide", list(w_), p_["strategy"])
# schedule of a factor made by the plan (batched device factorisation), and the time of one 32-rhs solve with it
import ctypes as C
al = nep.aligned_terms_dev(); indptr, indices, D_dev, G = al
fv = nep.get_fv(); lam0 = 250.0 ** 2 + 1e4 * np.exp(0.3j)
Cf = np.array([[f.derivs(lam0, 1)[0] for f in fv]], dtype=np.complex128)
plan = [p for p in _DeviceRefactor.plans.values() if p["state"] == "ready"][0]
lu = _DeviceRefactor.factor_batch_terms(plan, nep.n, D_dev, Cf, np.ones(1), expected_solves=1, growth=1e3)[0]
sch = (na._lib.c_i64 * 8)(); na._lib.lib.nep_lu_schedule(lu.h, sch); inf = (na._lib.c_i64 * 6)(); na._lib.lib.nep_lu_info(lu.h, inf)
Vd = na.to_dev(Vh)
X = lu.solve(Vd); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): X = lu.solve(Vd)
torch.cuda.synchronize()
print("schedule", list(sch), "info", list(inf), "solve 32 rhs: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), "launches", lu.launches_last_solve())
