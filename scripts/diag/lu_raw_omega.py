"""Componentwise backward error of the RAW device triangular solve (no refinement) for the gun M(sigma), under different
schedule choices (NEP_LU_TAIL / NEP_LU_MID / NEP_LU_BLOCK in the environment).  Usage: python scripts/diag/lu_raw_omega.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import nep_amd as na

nep = na.nep_gallery("gun_spmf_scaled")
n = nep.n
M = sp.csr_matrix(nep.compute_Mder(0.0))
lu = na.DeviceLU(M)
rng = np.random.default_rng(0)
om = []
for t in range(5):
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x = na.to_host(lu.solve(na.to_dev(b)))[:, 0]
    r = b - M @ x
    den = abs(M) @ abs(x) + abs(b)
    om.append(float(np.max(abs(r) / den)))
print(dict(tail=lu.tail, mid_rows=lu.mid_rows, mid_block=lu.mid_block, omega_raw=["%.1e" % o for o in om],
           env={k: v for k, v in os.environ.items() if k.startswith("NEP_LU")}))
