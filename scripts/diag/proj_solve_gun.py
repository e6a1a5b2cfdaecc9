import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na
# proj_solve=true on the gun SPMF (config C2 problem): Ritz extraction by Galerkin projection + inner iar
nep = na.nep_gallery("gun_spmf_scaled"); n = nep.n
for ps in (False, True):
    t = time.perf_counter()
    lam, Q, _ = na.iar(nep, maxit=40, neigs=np.inf, v=np.ones(n), tol=1e-10, check_error_every=10, proj_solve=ps,
                       inner_solver_method=na.IARInnerSolver(maxit=60))
    torch.cuda.synchronize()
    print("proj_solve", ps, "pairs", len(lam), "time %.2f s" % (time.perf_counter() - t))
