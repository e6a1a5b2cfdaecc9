"""diagnostic: the nlar call of tests/test_gpu_solvers.py::test_nlar_gun_twin_vs_oracle with its error history printed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import nep_amd as na
n = 400
nep1 = na.nep_gallery("gun_spmf_scaled", n)
kw = dict(tol=1e-10, lam=0, maxit=100, neigs=2, R=0.01, v=np.ones(n), max_subspace=150, num_restart_ritz_vecs=8)
try:
    D, X, hist = na.nlar(nep1, inner_solver_method=na.IARInnerSolver(), **kw)
    print("converged", D, "iterations", len(hist))
    print("hist tail", [float("%.3e" % h) for h in np.asarray(hist).ravel()[-12:]])
except na.NoConvergenceException as e:
    print("NOCONV", str(e)[:200])
