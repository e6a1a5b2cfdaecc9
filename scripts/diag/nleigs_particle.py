"""The "particle in a canyon" low-rank NEP of test/nleigs/particle_test_utils.jl (n = 16281, 83 terms, r = 162) through
nleigs variants R2 (dynamic) and S (static) on the device; optional oracle parity.
Usage: python scripts/diag/nleigs_particle.py [--oracle]"""
import os, sys, time, json, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nep_amd as na

warnings.simplefilter("ignore")
nep, Sigma, Xi, v, nodes, xmin, xmax = na.gallery.particle_init(2)
nep.dev
variants = {"R2": dict(Xi=Xi, maxdgr=50, minit=30, maxit=100, v=v, nodes=nodes),
            "S": dict(Xi=Xi, maxdgr=50, minit=120, maxit=200, v=v, nodes=nodes, static=True)}
res_dev = {}
for name, kw in variants.items():
    for rep in range(2):
        info = {}
        torch.cuda.synchronize(); t = time.perf_counter()
        lam, X, res = na.nleigs(nep, Sigma, info=info, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    res_dev[name] = lam
    print(json.dumps(dict(variant=name, eigenpairs=len(lam), lam=[str(l) for l in lam], res=[float(r) for r in res], s=dt,
                          N=info["N"], k=info["k"], nfact=info["nfact"], vrows=info["vrows"])), flush=True)
if "--oracle" in sys.argv:
    from oracle import gallery as og, nleigs as onl
    onep = og.particle_init(2)[0]
    for name, kw in variants.items():
        t = time.perf_counter()
        lo, Xo, ro = onl.nleigs(onep, Sigma, **kw)
        print(json.dumps(dict(variant=name, oracle_eigenpairs=len(lo), lam=[str(l) for l in lo], s=time.perf_counter() - t)), flush=True)
