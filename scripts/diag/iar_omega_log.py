"""Distribution of the recorded componentwise backward errors omega(x_0), omega(x_1), omega(x_2) of the refined solves of
the headline iar run (gun SPMF, m = 100), device and host numeric LU.  Usage: python scripts/diag/iar_omega_log.py [runs]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
from nep_amd import linsolvers
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
EPS = np.finfo(float).eps
for mode in ("device", "host"):
    if mode == "host":
        os.environ["NEP_LU_DEV"] = "0"
    for rep in range(R):
        linsolvers.FactorizeLinSolver._omega_log = log = []
        lam, Q = na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)[:2]
        W = np.array([w + [np.nan] * (3 - len(w)) for w in log])
        print(mode, rep, "pairs", len(lam), "steps", len(W), "omega0 max %.2e" % np.nanmax(W[:, 0]),
              "omega1/eps: median %.2f max %.2f  (>2eps: %d, >4eps: %d)" % (np.nanmedian(W[:, 1]) / EPS, np.nanmax(W[:, 1]) / EPS,
               int(np.sum(W[:, 1] > 2 * EPS)), int(np.sum(W[:, 1] > 4 * EPS))),
              "omega2/eps max %.2f" % (np.nanmax(W[:, 2]) / EPS if np.any(np.isfinite(W[:, 2])) else np.nan), flush=True)
