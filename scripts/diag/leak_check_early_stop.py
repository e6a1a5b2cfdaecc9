"""iar calls that end early (neigs reached: speculative steps and checks are dropped) or with NoConvergenceException, repeated:
device memory must plateau, nothing may hang"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20
out = {"ok": 0, "noconv": 0, "other": 0}
for i in range(121):
    try:
        if i % 3 == 0:
            lam = na.iar(nep, maxit=100, neigs=5, v=np.ones(nep.n), tol=1e-10)[0]          # stops around step 40
        elif i % 3 == 1:
            lam = na.iar(nep, maxit=12, neigs=8, v=np.ones(nep.n), tol=1e-12)[0]            # cannot converge
        else:
            lam = na.iar(nep, maxit=60, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, check_error_every=7)[0]
        out["ok"] += 1
    except na.NoConvergenceException:
        out["noconv"] += 1
    except Exception as e:
        out["other"] += 1; print("EXC", repr(e)[:200], flush=True)
    if i % 20 == 0: print("call %d: %.0f MiB in use %s" % (i, used(), out), flush=True)
