"""device memory over repeated creation / destruction of NEP objects, factorisations and solver objects: should plateau"""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.sparse as sp
import nep_amd as na
def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20
for i in range(41):
    nep = na.nep_gallery("gun_spmf_scaled"); nep.dev                       # uploads the terms
    z = nep.compute_Mlincomb(0.1 + 0.2j, np.ones((nep.n, 3)))
    lu = na.DeviceLU(nep.compute_Mder(0.3))
    x = lu.solve(torch.ones(nep.n, dtype=torch.complex128, device="cuda"))
    s = na.linsolvers.create_linsolver(na.FactorizeLinSolverCreator(), nep, 0.2)
    y = na.lin_solve(s, np.ones(nep.n))
    lam = na.iar(nep, maxit=20, neigs=2, v=np.ones(nep.n), tol=1e-8)[0]
    del nep, lu, x, s, y, z; gc.collect()
    if i % 10 == 0: print("round %d: %.0f MiB in use" % (i, used()), flush=True)
wep = None
for i in range(11):
    wep = na.nep_gallery("WEP", nx=303, nz=299, benchmark_problem="JARLEBRING"); wep.dev
    z = wep.compute_Mlincomb(-3 - 3.5j, np.ones((wep.n, 2)))
    del wep, z; gc.collect()
    if i % 5 == 0: print("WEP round %d: %.0f MiB in use" % (i, used()), flush=True)
