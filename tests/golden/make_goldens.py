"""Generates the golden OUTPUT fixtures of SURVEY.md section 8c from the CPU oracle (NumPy/SciPy restatement of the
reference): run in the build container,

    python tests/golden/make_goldens.py

writes tests/golden/goldens.npz (inputs are regenerated from seeds by the tests, only expected outputs are stored):
  K1   compute_Mlincomb on a 200-row synthetic sparse SPMF with the four gun functions (shift-and-scaled as in config C2),
       k in {1, 2, 7, 33}, generic a with a zero entry
  K6   DGKS: h, beta and the normalised w for a generic block and for a forced re-orthogonalisation case
  K5   solution of M(sigma) x = b for the reduced gun matrix (n = 1310), 3 right-hand sides
  iar  error histories (sorted backward errors per step) for dep0(100) (docstring call, src/method_tiar.jl:37-45) and for the
       gun stand-in at reduced n = 1310, m = 30
  tiar eigenvalues of the docstring call
  Beyn A0, A1 for dep0 (n = 5) with the fixed probe block, N = 64, k = 3, sigma = 0.2, radius = 1
Both the oracle (`-m "not gpu"`: drift of the restatement) and the HIP path (`-m gpu`) are tested against this file
(tests/test_goldens.py)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import gallery as og, neps, solvers as osol     # noqa: E402


def synthetic_spmf(n=200, seed=7):
    """4 sparse real terms of a 200-row problem + the gun functions composed with the C2 shift/scale"""
    rng = np.random.default_rng(seed)
    Av = [sp.random(n, n, density=d, random_state=rng, format="csr") + (sp.identity(n) if i == 0 else 0 * sp.identity(n))
          for i, d in enumerate((0.03, 0.05, 0.01, 0.02))]
    Av = [sp.csr_matrix(A) for A in Av]
    base = [neps.f_one(), neps.f_id(), neps.f_isqrt(0.0), neps.f_isqrt(-108.8774 ** 2)]
    shift, scale = 250.0 ** 2, 330.0 ** 2 - 220.0 ** 2
    fv = [neps.f_compose_affine(f, scale, shift) for f in base]
    return neps.SPMF_NEP(Av, fv)


def k1_inputs(n, k, seed):
    rng = np.random.default_rng(1000 + seed)
    V = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    a = rng.standard_normal(k)
    if k > 2:
        a[1] = 0.0
    return V, a


def dgks_inputs(rows=700, k=9, forced=False, seed=3):
    rng = np.random.default_rng(seed)
    V, _ = np.linalg.qr(rng.standard_normal((rows, k)) + 1j * rng.standard_normal((rows, k)))
    w = rng.standard_normal(rows) + 1j * rng.standard_normal(rows)
    if forced:                      # w almost inside span(V): the first pass cancels 7 digits -> a second pass is required
        w = V @ (rng.standard_normal(k) + 1j * rng.standard_normal(k)) + 1e-7 * w
    return V, w


def main():
    out = {}
    nep = synthetic_spmf()
    lam = 0.013 + 0.002j
    for k in (1, 2, 7, 33):
        V, a = k1_inputs(200, k, k)
        out["k1_k%d" % k] = nep.compute_Mlincomb(lam, V.copy(), a.copy())
    for name, forced in (("generic", False), ("forced", True)):
        V, w = dgks_inputs(forced=forced)
        h = np.zeros(9, dtype=complex)
        beta = osol.dgks(V, w, h)
        out["dgks_%s_h" % name] = h; out["dgks_%s_beta" % name] = np.array(beta); out["dgks_%s_w" % name] = w
    # K5
    A = sp.csc_matrix(og.nlevp_native_gun(1310).compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    rng = np.random.default_rng(21)
    B = rng.standard_normal((1310, 3)) + 1j * rng.standard_normal((1310, 3))
    import scipy.sparse.linalg as spla
    out["k5_X"] = spla.splu(A).solve(B)
    # iar / tiar histories
    d100 = og.dep0(100)
    hist = []
    lam_i, _, _ = osol.iar(d100, v=np.ones(100), tol=1e-5, neigs=3, errhist=hist)
    out["iar_dep0_lam"] = np.sort_complex(lam_i)
    out["iar_dep0_hist_best"] = np.array([h[0] for h in hist])
    lam_t = osol.tiar(d100, v=np.ones(100), tol=1e-5, neigs=3)[0]
    out["tiar_dep0_lam"] = np.sort_complex(lam_t)
    gun = og.gun_spmf_scaled(1310)
    hist = []
    try:
        osol.iar(neps.DerSPMF(gun, 0.0, 30), maxit=30, neigs=np.inf, v=np.ones(1310), tol=1e-10,
                 errmeasure=osol.StandardSPMFErrmeasure(gun), errhist=hist)
    except osol.NoConvergenceException:
        pass
    m = len(hist)
    H = np.full((m, 5), np.nan)
    for i, h in enumerate(hist):
        H[i, :min(5, len(h))] = h[:5]
    out["iar_gun1310_hist5"] = H
    # Beyn moments
    dep = og.dep0()
    info = {}
    osol.contour_beyn(dep, sigma=0.2, radius=1.0, neigs=4, k=3, N=64, sanity_check=False, Vh=osol.probe_block(5, 3), info=info)
    out["beyn_dep0_A0"] = info["A0"]; out["beyn_dep0_A1"] = info["A1"]
    np.savez_compressed(os.path.join(HERE, "goldens.npz"), **out)
    print("wrote", os.path.join(HERE, "goldens.npz"), {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
