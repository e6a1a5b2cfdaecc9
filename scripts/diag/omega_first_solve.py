"""componentwise backward error omega of the FIRST solve (before any refinement sweep), host SuperLU factors against device
static-pivot factors, gun M(0): does the device factorisation cost the refinement sweep every Arnoldi step takes?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nep_amd as na
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf_scaled"); n = nep.n
rng = np.random.default_rng(0)
out = {}
for tag in ("host", "device"):
    s = na.create_linsolver(na.FactorizeLinSolverCreator(max_factorizations=0), nep, 0.0)
    if tag == "host":
        _DeviceRefactor.wait()
    dev = bool(getattr(s.lu, "device_factorized", False))
    s._refine_setup(); W = s._W
    om1 = []; om2 = []
    for t in range(40):
        b = na.to_dev(rng.standard_normal(n) + 1j * rng.standard_normal(n))[0] if t % 2 else na.to_dev(np.ones(n) / (1 + t))[0]
        bd = b.reshape(1, n)
        s.lu.solve(bd, out=W[1].reshape(1, n))
        om1.append(float(s._residual(bd, True)))
        s.lu.solve(W[0].reshape(1, n), out=W[3].reshape(1, n))
        from nep_amd import dense
        dense.axpy(1.0, W[3], W[1], n)
        om2.append(float(s._residual(bd, True)))
    out[tag] = {"device_factorized": dev, "growth": getattr(s.lu, "growth", None), "omega1_median": float(np.median(om1)), "omega1_max": float(np.max(om1)),
                "omega1_min": float(np.min(om1)), "omega2_median": float(np.median(om2)), "omega2_max": float(np.max(om2)), "eps": 2.220446049250313e-16}
print(json.dumps(out, indent=1))
