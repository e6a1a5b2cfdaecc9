import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na
# test/nlar.jl:12-44 at full size (gun stand-in n = 9956)
nep = na.nep_gallery("nlevp_native_gun"); n = nep.n
shift, scale = 250.0 ** 2, 330.0 ** 2 - 220.0 ** 2
TOL = 1e-10
t = time.perf_counter()
lref, vref = na.quasinewton(nep, lam=shift + scale * (-0.131403 + 0.00759532j), v=np.ones(n), tol=TOL / 50, maxit=500)
print("quasinewton %.2f s" % (time.perf_counter() - t), lref)
nep1 = na.nep_gallery("gun_spmf_scaled")
t = time.perf_counter()
D, X, hist = na.nlar(nep1, tol=TOL, lam=0, maxit=100, neigs=2, R=0.01, v=np.ones(n), inner_solver_method=na.IARInnerSolver(),
                     num_restart_ritz_vecs=8, max_subspace=150)
print("nlar %.2f s" % (time.perf_counter() - t), shift + scale * D)
Av = nep.get_Av(); fv = nep.get_fv()
for i in range(2):
    lo = shift + scale * D[i]
    r = sum(f.derivs(lo, 1)[0] * (A @ X[:, i]) for f, A in zip(fv, Av))
    print("residual", np.linalg.norm(r), "dist to quasinewton eigenvalue", abs(lo - lref))
