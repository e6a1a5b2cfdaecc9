"""are repeated iar runs bit-identical? (the switch to the dense apex of K5 used to depend on timing)
   python scripts/diag/repro_bits.py [maxit] [runs]"""
import os, sys, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
import nep_amd as na
from nep_amd.linsolvers import _DeviceRefactor
iarmod = None
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 9956
nep = na.nep_gallery("gun_spmf_scaled", n); nep.dev
kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
out = []
for i in range(nr):
    hist = []
    lam, Q, _ = na.iar(nep, errhist=hist, **kw)
    torch.cuda.synchronize()
    if i == 0: _DeviceRefactor.wait()
    lam = np.asarray(lam); Q = np.asarray(Q)
    order = np.lexsort((lam.imag, lam.real))
    h = lambda a: hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=6).hexdigest()
    out.append({"run": i, "pairs": int(len(lam)), "lam_hash": h(lam), "lam_sorted_hash": h(lam[order]), "Q_hash": h(Q),
                "hist_hash": h(np.concatenate([np.sort(np.asarray(e, dtype=float)) for e in hist])) if hist else None,
                "refine_hint": getattr(nep, "_refine_hint", None), "orth_pass_misses": getattr(na.iar, "orth_pass_misses", None)})
for o in out: print(json.dumps(o))
