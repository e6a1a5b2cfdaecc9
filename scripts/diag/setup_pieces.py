"""diagnostic: host-side pieces of one FactorizeLinSolver set-up at the steady state (gun, pattern plan ready)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
import nep_amd as na
from nep_amd.linsolvers import _DeviceRefactor, DeviceLU, FactorizeLinSolver
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
sigma = 0.0
for i in range(3):
    s = FactorizeLinSolver(nep, sigma); torch.cuda.synchronize()
    if i == 0:
        _DeviceRefactor.wait()
def T(f, n=20):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return r, 1e3 * float(np.median(ts))
A, t = T(lambda: nep.compute_Mder(sigma)); print("compute_Mder %.3f ms (%s)" % (t, type(A).__name__))
Ac, t = T(lambda: sp.csc_matrix(A, dtype=np.complex128)); print("csc_matrix   %.3f ms" % t)
k, t = T(lambda: _DeviceRefactor.key(Ac, (None, None, None))); print("key hash     %.3f ms" % t)
plan = _DeviceRefactor.lookup(k); print("plan", None if plan is None else plan["state"])
_, t = T(lambda: DeviceLU(A, expected_solves=200)); print("DeviceLU(A)  %.3f ms" % t)
_, t = T(lambda: FactorizeLinSolver(nep, sigma)); print("FactorizeLinSolver %.3f ms" % t)
al = nep.aligned_terms_dev()
if al is not None and plan is not None:
    indptr, indices, D_dev, G = al
    fv = nep.get_fv()
    def terms():
        Cf = np.array([[f.derivs(sigma, 1)[0] for f in fv]], dtype=np.complex128)
        normA = np.sqrt(np.maximum(np.einsum("bs,st,bt->b", Cf.conj(), G, Cf).real, 0.0))
        return _DeviceRefactor.factor_batch_terms(plan, nep.n, D_dev, Cf, normA, expected_solves=200)
    r, t = T(terms); print("factor_batch_terms(B=1) %.3f ms  ok=%s" % (t, r[0] is not None))
