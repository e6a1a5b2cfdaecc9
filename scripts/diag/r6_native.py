"""nep_iar_run (one foreign call) against the step-at-a-time Python pipeline on config C2: same pairs, time per call"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nep_amd as na
import torch
from nep_amd import iar as iar_mod
from nep_amd.linsolvers import _DeviceRefactor

nep = na.nep_gallery("gun_spmf_scaled")
def run(native, reps=12, hist=None, neigs=np.inf, maxit=100):
    os.environ["NEP_IAR_NATIVE_RUN"] = "1" if native else "0"
    ts = []
    out = None
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        creator = na.FactorizeLinSolverCreator(max_factorizations=0)
        h = [] if hist is not None else None
        try:
            out = na.iar(nep, sigma=0.0, gamma=1.0, maxit=maxit, neigs=neigs, v=np.ones(nep.n), tol=1e-10, linsolvercreator=creator, errhist=h)
        except na.NoConvergenceException as e:
            out = (e.lam, e.v, None)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        if r == 0:
            _DeviceRefactor.wait()
        if hist is not None:
            hist[:] = h
    return out, np.array(ts) * 1e3

h0 = []; h1 = []
(l0, Q0, _), t0 = run(False, hist=h0)
(l1, Q1, _), t1 = run(True, hist=h1)
print("python pipeline: %d pairs, ms per call %s" % (len(l0), np.round(t0, 2)))
print("nep_iar_run    : %d pairs, ms per call %s  (native runs %d, misses %d)" % (len(l1), np.round(t1, 2), iar_mod.native_runs, iar_mod.native_run_misses))
print("max |lam diff| %.3e" % np.abs(np.sort_complex(l0) - np.sort_complex(l1)).max())
print("hist lens", len(h0), len(h1), "max rel diff of the 8 smallest errors at step 100: %.3e" % np.max(np.abs(h0[-1][:8] - h1[-1][:8]) / h0[-1][:8]))
for i in range(len(l1)):
    j = np.argmin(np.abs(l0 - l1[i]))
    a = Q0[:, j] / Q0[np.argmax(np.abs(Q0[:, j])), j]; b = Q1[:, i] / Q1[np.argmax(np.abs(Q0[:, j])), i]
    d = np.linalg.norm(a - b) / np.linalg.norm(a)
    if d > 1e-6:
        print("vector", i, "differs", d)
# finite neigs
for ne in (5, 200):
    (a0, _, _), _ = run(False, reps=2, neigs=ne, maxit=60)
    (a1, _, _), _ = run(True, reps=2, neigs=ne, maxit=60)
    print("neigs", ne, len(a0), len(a1), "max diff %.3e" % (np.abs(np.sort_complex(a0) - np.sort_complex(a1)).max() if len(a0) == len(a1) else -1))
