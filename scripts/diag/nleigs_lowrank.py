"""nleigs on gun variant R1: PEP + SPMF (full blocks) against PEP + LowRankFactorizedNEP (compressed blocks) on the
device, and the compressed run against the oracle.  Usage: python scripts/diag/nleigs_lowrank.py [n] [--oracle]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nep_amd as na
from scripts.run_configs import gun_r1, match

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 9956
Sigma, Xi, nodes = gun_r1()
K, M, W1, W2 = na.gallery.gun_matrices(n)
fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -na.gallery.GUN_SIGMA2 ** 2)]
full = na.SumNEP(na.PEP([K, -M]), na.SPMF_NEP([W1, W2], fv))
lowr = na.SumNEP(na.PEP([K, -M]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]), na.LowRankMatrixAndFunction(W2, fv[1])]))
v = np.random.Generator(np.random.Philox(1)).standard_normal(n) + 0j
out = {}
for name, nep in (("full", full), ("lowrank", lowr)):
    nep.dev
    for rep in range(2):
        info = {}
        torch.cuda.synchronize(); t = time.perf_counter()
        lam, X, res = na.nleigs(nep, Sigma, Xi=Xi, maxit=100, v=v, leja=0, nodes=nodes, reusefact=2, tol=1e-10,
                                errmeasure=na.StandardSPMFErrmeasure(nep), info=info)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    out[name] = lam
    E = na.StandardSPMFErrmeasure(full)
    print(json.dumps(dict(case=name, n=n, eigenpairs=len(lam), s=dt, nfact=info["nfact"], N=info["N"],
                          max_backward_error=max([float(na.estimate_error(E, lam[i], X[:, i])) for i in range(len(lam))] + [0.0]))), flush=True)
print("same eigenvalues:", match(out["full"], out["lowrank"], 1e-8))
if "--oracle" in sys.argv:
    from oracle import neps as on, nleigs as onl, solvers as osol
    ofv = [on.f_isqrt(0.0), on.f_isqrt(-na.gallery.GUN_SIGMA2 ** 2)]
    olr = on.SumNEP(on.PEP([K, -M]), on.LowRankFactorizedNEP([on.LowRankMatrixAndFunction(W1, ofv[0]), on.LowRankMatrixAndFunction(W2, ofv[1])]))
    t = time.perf_counter()
    lo, Xo, ro = onl.nleigs(olr, Sigma, Xi=Xi, maxit=100, v=v, leja=0, nodes=nodes, reusefact=2, tol=1e-10,
                            errmeasure=osol.StandardSPMFErrmeasure(olr))
    print(json.dumps(dict(case="oracle lowrank", eigenpairs=len(lo), s=time.perf_counter() - t)), "parity:", match(out["lowrank"], lo, 1e-8))
