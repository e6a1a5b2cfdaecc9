"""Golden OUTPUT fixtures (tests/golden/goldens.npz, generated from the CPU oracle by tests/golden/make_goldens.py --
SURVEY.md section 8c "Fixtures to commit").  The inputs are regenerated from the same seeds; the CPU tests pin the oracle
against drift, the `-m gpu` tests pin the HIP path against the same numbers."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_goldens as mg      # noqa: E402  (input generators shared with the fixture script)
from oracle import gallery as og, neps, solvers as osol     # noqa: E402

G = np.load(os.path.join(HERE, "golden", "goldens.npz"))
LAM = 0.013 + 0.002j


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


# ---- the oracle against its own committed outputs (CPU) ----------------------------------------------------------------------
def test_oracle_k1_dgks_k5_goldens():
    nep = mg.synthetic_spmf()
    for k in (1, 2, 7, 33):
        V, a = mg.k1_inputs(200, k, k)
        assert rel(nep.compute_Mlincomb(LAM, V, a), G["k1_k%d" % k]) < 1e-13
    for name, forced in (("generic", False), ("forced", True)):
        V, w = mg.dgks_inputs(forced=forced)
        h = np.zeros(9, dtype=complex)
        beta = osol.dgks(V, w, h)
        assert rel(h, G["dgks_%s_h" % name]) < 1e-13 and abs(beta - G["dgks_%s_beta" % name]) < 1e-13 * beta
        assert rel(w, G["dgks_%s_w" % name]) < 1e-8            # forced case: w is what is left after 7 cancelled digits
    import scipy.sparse.linalg as spla
    A = sp.csc_matrix(og.nlevp_native_gun(1310).compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    B = np.random.default_rng(21).standard_normal((1310, 3)) + 1j * np.random.default_rng(21).standard_normal((1310, 3))
    rng = np.random.default_rng(21)
    B = rng.standard_normal((1310, 3)) + 1j * rng.standard_normal((1310, 3))
    assert rel(spla.splu(A).solve(B), G["k5_X"]) < 1e-11


def test_oracle_driver_goldens():
    d100 = og.dep0(100)
    hist = []
    lam, _, _ = osol.iar(d100, v=np.ones(100), tol=1e-5, neigs=3, errhist=hist)
    assert np.allclose(np.sort_complex(lam), G["iar_dep0_lam"], rtol=0, atol=1e-12)
    assert np.allclose([h[0] for h in hist], G["iar_dep0_hist_best"], rtol=1e-6)
    lam_t = osol.tiar(d100, v=np.ones(100), tol=1e-5, neigs=3)[0]
    assert np.allclose(np.sort_complex(lam_t), G["tiar_dep0_lam"], rtol=0, atol=1e-12)
    info = {}
    osol.contour_beyn(og.dep0(), sigma=0.2, radius=1.0, neigs=4, k=3, N=64, sanity_check=False, Vh=osol.probe_block(5, 3), info=info)
    assert rel(info["A0"], G["beyn_dep0_A0"]) < 1e-13 and rel(info["A1"], G["beyn_dep0_A1"]) < 1e-13


# ---- the HIP path against the same numbers -----------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def na():
    import nep_amd
    assert nep_amd.device_count() >= 1
    return nep_amd


def _device_synthetic(na):
    onep = mg.synthetic_spmf()
    f = na.funcs
    base = na.SPMF_NEP(onep.get_Av(), [f.one(), f.ident(), f.ISqrt(1.0, 0.0), f.ISqrt(1.0, -108.8774 ** 2)])
    return na.shift_and_scale(base, shift=250.0 ** 2, scale=330.0 ** 2 - 220.0 ** 2)


@pytest.mark.gpu
def test_gpu_k1_goldens(na):
    nep = _device_synthetic(na)
    for k in (1, 2, 7, 33):
        V, a = mg.k1_inputs(200, k, k)
        assert rel(nep.compute_Mlincomb(LAM, V, a), G["k1_k%d" % k]) < 1e-12


@pytest.mark.gpu
def test_gpu_dgks_goldens(na):
    for name, forced in (("generic", False), ("forced", True)):
        V, w = mg.dgks_inputs(forced=forced)
        Vd = na.to_dev(V); wd = na.to_dev(w)[0]
        h, beta, npass = na.orthogonalize_and_normalize(Vd, wd, 9)
        assert npass == (2 if forced else 1)
        assert rel(h, G["dgks_%s_h" % name]) < 1e-12 and abs(beta - G["dgks_%s_beta" % name]) < 1e-9 * beta
        assert rel(na.to_host(wd.reshape(1, -1))[:, 0], G["dgks_%s_w" % name]) < (1e-6 if forced else 1e-12)


@pytest.mark.gpu
def test_gpu_k5_goldens(na):
    A = sp.csc_matrix(og.nlevp_native_gun(1310).compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    rng = np.random.default_rng(21)
    B = rng.standard_normal((1310, 3)) + 1j * rng.standard_normal((1310, 3))
    X = na.to_host(na.DeviceLU(A).solve(na.to_dev(B)))
    assert rel(X, G["k5_X"]) < 1e-9
    ls = na.FactorizeLinSolver(na.nep_gallery("nlevp_native_gun", 1310), 250.0 ** 2 + 1j)      # with UMFPACK-style refinement
    x = na.lin_solve(ls, B[:, 0])
    assert rel(x, G["k5_X"][:, 0]) < 1e-12


@pytest.mark.gpu
def test_gpu_driver_goldens(na):
    d100 = na.nep_gallery("dep0", 100)
    hist = []
    lam, _, _ = na.iar(d100, v=np.ones(100), tol=1e-5, neigs=3, errhist=hist)
    assert np.allclose(np.sort_complex(lam), G["iar_dep0_lam"], rtol=0, atol=1e-10)
    best = np.array([h[0] for h in hist])
    m = min(len(best), len(G["iar_dep0_hist_best"]))
    assert np.all((best[:m] / G["iar_dep0_hist_best"][:m] < 10) & (best[:m] / G["iar_dep0_hist_best"][:m] > 0.1))
    lam_t = na.tiar(d100, v=np.ones(100), tol=1e-5, neigs=3)[0]
    assert np.allclose(np.sort_complex(lam_t), G["tiar_dep0_lam"], rtol=0, atol=1e-10)
    gun = na.nep_gallery("gun_spmf_scaled", 1310)
    hist = []
    try:
        na.iar(gun, maxit=30, neigs=np.inf, v=np.ones(1310), tol=1e-10, errhist=hist)
    except na.NoConvergenceException:
        pass
    H = G["iar_gun1310_hist5"]
    assert len(hist) == H.shape[0]
    for i, h in enumerate(hist):                      # SURVEY.md section 8d rule (iv): within a factor 10 above 1e-12
        for j in range(min(5, len(h))):
            if H[i, j] > 1e-12:
                assert 0.1 < h[j] / H[i, j] < 10, (i, j, h[j], H[i, j])
    info = {}
    na.contour_beyn(na.nep_gallery("dep0"), sigma=0.2, radius=1.0, neigs=4, k=3, N=64, sanity_check=False, Vh=na.probe_block(5, 3), info=info)
    assert rel(info["A0"], G["beyn_dep0_A0"]) < 1e-11 and rel(info["A1"], G["beyn_dep0_A1"]) < 1e-11
