import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, torch, nep_amd as na
# repeated pipelined iar runs must return the same eigenvalues every time (no race in the event-ordered pipeline)
nep = na.nep_gallery("gun_spmf_scaled"); n = nep.n
ref = None; worst = 0.0; t0 = time.time()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    lam, Q, _ = na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(n), tol=1e-10)
    lam = np.sort_complex(lam)
    if ref is None:
        ref = lam
    assert len(lam) == len(ref), (i, len(lam), len(ref))
    worst = max(worst, float(np.max(abs(lam - ref))))
print("runs ok, pairs", len(ref), "max eigenvalue deviation between runs %.2e" % worst, "time %.1f s" % (time.time() - t0),
      "mem reserved %.2f GB" % (torch.cuda.memory_reserved() / 1e9))
