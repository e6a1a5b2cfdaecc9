import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na
from nep_amd import dense
# count the DGKS passes of every step of the gun iar run with the synchronous orthogonalisation
nep = na.nep_gallery("gun_spmf_scaled")
passes = []
orig = dense.orthogonalize_and_normalize
def wrapped(*a, **k):
    h, beta, np_ = orig(*a, **k); passes.append(np_); return h, beta, np_
dense.orthogonalize_and_normalize = wrapped
os.environ["NEP_IAR_SYNC"] = "1"
lam, Q, V = na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
print("pairs", len(lam), "steps", len(passes), "mean DGKS passes %.2f" % np.mean(passes), "histogram", np.bincount(passes))
