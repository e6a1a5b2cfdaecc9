import sys, os, time, traceback; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
n = 400
nep1 = na.nep_gallery("gun_spmf_scaled", n)
import nep_amd.nlar as _x
mod = sys.modules[na.nlar.__module__]
orig = mod.inner_solve
def traced(solver, pnep, **kw):
    dd, vv = orig(solver, pnep, **kw)
    print("inner k=%d ->" % pnep.size(1), np.asarray(dd)[:4], "finite", np.all(np.isfinite(np.asarray(vv))), flush=True)
    return dd, vv
mod.inner_solve = traced
try:
    D, X, hist = na.nlar(nep1, tol=1e-10, lam=0, maxit=100, neigs=2, R=0.01, v=np.ones(n), inner_solver_method=na.IARInnerSolver(), max_subspace=150)
except Exception as e:
    traceback.print_exc()

print("main done")
try:
    D2, X2, _ = na.nlar(nep1, tol=1e-10, lam=0, maxit=100, neigs=1, R=0.01, v=np.ones(n), inner_solver_method=na.IARInnerSolver(), eigval_sorter=na.default_eigval_sorter, max_subspace=150)
    print("default sorter", D2)
except Exception:
    traceback.print_exc()
try:
    na.nlar(nep1, tol=1e-20, maxit=3, neigs=3, v=np.ones(n), inner_solver_method=na.IARInnerSolver())
except Exception:
    traceback.print_exc()
