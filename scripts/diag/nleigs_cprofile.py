"""cProfile of nleigs on gun R1 (low-rank or full).  Usage: python scripts/diag/nleigs_cprofile.py [lowrank|full]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nep_amd as na
from scripts.run_configs import gun_r1

n = 9956
Sigma, Xi, nodes = gun_r1()
K, M, W1, W2 = na.gallery.gun_matrices(n)
fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -na.gallery.GUN_SIGMA2 ** 2)]
if len(sys.argv) > 1 and sys.argv[1] == "full":
    nep = na.SumNEP(na.PEP([K, -M]), na.SPMF_NEP([W1, W2], fv))
else:
    nep = na.SumNEP(na.PEP([K, -M]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]), na.LowRankMatrixAndFunction(W2, fv[1])]))
v = np.random.Generator(np.random.Philox(1)).standard_normal(n) + 0j
nep.dev
run = lambda: na.nleigs(nep, Sigma, Xi=Xi, maxit=100, v=v, leja=0, nodes=nodes, reusefact=2, tol=1e-10,
                        errmeasure=na.StandardSPMFErrmeasure(nep))
run()
torch.cuda.synchronize(); t = time.perf_counter(); run(); torch.cuda.synchronize(); print("wall %.3f s" % (time.perf_counter() - t))
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
