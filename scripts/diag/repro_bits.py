"""are two iar runs of config C2 bit-identical? (the switch to the dense apex of K5 used to depend on timing)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
runs = []
for i in range(8):
    lam, Q = bc.c2_device(na, nep, 100, return_device=False)
    torch.cuda.synchronize()
    if i == 1: _DeviceRefactor.wait()
    runs.append((np.array(lam), np.array(Q)))
ref = runs[3]
out = []
for i, (l, q) in enumerate(runs):
    same_l = l.shape == ref[0].shape and bool(np.array_equal(l.view(np.float64), ref[0].view(np.float64)))
    same_q = q.shape == ref[1].shape and bool(np.array_equal(q.view(np.float64), ref[1].view(np.float64)))
    d = float(np.max(np.abs(l - ref[0]) / np.abs(ref[0]))) if l.shape == ref[0].shape else None
    out.append({"run": i, "pairs": int(len(l)), "lam_bit_identical_to_run3": same_l, "Q_bit_identical": same_q, "max_rel_lam_diff": d})
print(json.dumps(out, indent=0))
