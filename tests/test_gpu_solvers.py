"""GPU parity tests of the drivers: same inputs through the HIP backend and the CPU oracle;
acceptance = the reference's own criterion verify_lambdas (test/runtests.jl:80-89): eigenpair count
+ residual below tol, plus eigenvalue agreement with the oracle run."""
import os

import numpy as np
import scipy.sparse as sp
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


@pytest.fixture(scope="module")
def na():
    import nep_amd
    assert nep_amd.device_count() >= 1, "no GPU visible"
    return nep_amd


def _match(l1, l2, rtol):
    l1 = list(l1); l2 = list(l2)
    assert len(l1) == len(l2)
    for x in l1:
        j = int(np.argmin([abs(x - y) for y in l2]))
        assert abs(x - l2[j]) <= rtol * max(1.0, abs(x)), (x, l2[j])
        l2.pop(j)


def test_iar_dep0_kat(na):
    # test/iar.jl:23-39 ; src/method_tiar.jl:37-45
    from oracle import gallery as og, solvers as osol
    nep = na.nep_gallery("dep0"); onep = og.dep0()
    R = na.ResidualErrmeasure(nep)
    lam, Q, V = na.iar(nep, sigma=1.1, v=np.ones(5), maxit=100, tol=EPS * 100, neigs=5, errmeasure=R)
    assert len(lam) == 5
    oR = osol.ResidualErrmeasure(onep)
    assert all(oR(lam[i], Q[:, i]) < EPS * 100 for i in range(5))        # independent host re-evaluation
    lam, Q, V = na.iar(nep, sigma=1.1, v=np.ones(5), maxit=38, tol=EPS * 100, neigs=np.inf)
    assert len(lam) == 6
    Vh = na.to_host(V)
    assert np.linalg.norm(Vh.conj().T @ Vh - np.eye(Vh.shape[1]), 2) < 1e-6
    nep100 = na.nep_gallery("dep0", 100)
    lam, Q, V = na.iar(nep100, v=np.ones(100), tol=1e-5, neigs=3)
    ref = np.array([-0.07708769561361105, 0.050462487743188206, 0.1503916927814904])
    assert np.allclose(np.sort(lam.real), ref, atol=1e-12)
    with pytest.raises(na.NoConvergenceException):
        na.iar(nep100, sigma=1.1, v=np.ones(100), neigs=6, maxit=7, tol=EPS * 100)


@pytest.mark.parametrize("orth", [0, 1, 2])
def test_iar_orthogonality(na, orth):
    # test/iar.jl:41-63
    nep = na.nep_gallery("dep0")
    lam, Q, V = na.iar(nep, orthmethod=orth, sigma=1.1, v=np.ones(5), maxit=100, tol=EPS * 100, neigs=5,
                       errmeasure=na.ResidualErrmeasure(nep))
    Vh = na.to_host(V)
    assert np.linalg.norm(Vh.conj().T @ Vh - np.eye(Vh.shape[1]), 2) < 1e-6


def test_iar_gun_twin_vs_oracle(na):
    """config C2 at reduced size (n=1310, m=40): identical eigenpair count, eigenvalues and error history."""
    from oracle import gallery as og, solvers as osol
    n, m = 1310, 40
    onep = og.gun_spmf_scaled(n)
    oder = __import__("oracle.neps", fromlist=["DerSPMF"]).DerSPMF(onep, 0.0, m)
    oh = []; gh = []
    lo, Qo, _ = osol.iar(oder, sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10,
                         errmeasure=osol.StandardSPMFErrmeasure(onep), errhist=oh)
    nep = na.nep_gallery("gun_spmf_scaled", n)
    lg, Qg, _ = na.iar(nep, sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10, errhist=gh)
    assert len(lg) == len(lo) and len(lg) >= 1
    _match(lg, lo, 1e-8)
    # independent FP64 host re-evaluation of the backward error of the GPU pairs
    oE = osol.StandardSPMFErrmeasure(onep)
    assert max(oE(lg[i], Qg[:, i]) for i in range(len(lg))) < 1e-10
    # error history agrees within a factor 10 wherever above 1e-12 (SURVEY.md section 8d parity rule iv)
    for eo, eg in zip(oh, gh):
        kk = min(len(eo), len(eg), 5)
        for a, b in zip(eo[:kk], eg[:kk]):
            if a > 1e-12 and b > 1e-12:
                assert 0.1 < a / b < 10


def test_iar_runs_are_bit_reproducible(na):
    """repeated iar calls on the same problem return the SAME eigenvalues to the last bit once the pattern's device-LU plan exists
    (K5's switch from the level sweep to the dense apex used to happen when a query found the background build
    finished, i.e. at a timing-dependent solve; now at a fixed solve of each factor, NEP_ML_APEX_AT).  Compared as a sorted set:
    the ORDER in which converged pairs are returned follows the host-side checks (threaded LAPACK / BLAS on worker threads, error
    estimates whose last bits vary) and may differ between runs (scripts/diag/repro_bits.py: 1 of 21 runs); the vectors are compared
    pair by pair up to the sign / phase free in an eigenvector."""
    from nep_amd.linsolvers import _DeviceRefactor
    n, m = 9956, 60
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    na.iar(nep, **kw); _DeviceRefactor.wait()
    runs = [na.iar(nep, **kw) for _ in range(4)]

    def canon(lam, Q):
        lam = np.asarray(lam); Q = np.asarray(Q)
        o = np.lexsort((lam.imag, lam.real))
        return lam[o], Q[:, o]
    l0, Q0 = canon(runs[0][0], runs[0][1])
    assert len(l0) >= 1
    for lam, Q, _ in runs[1:]:
        l1, Q1 = canon(lam, Q)
        assert np.array_equal(l1.view(np.float64), l0.view(np.float64))
        for j in range(len(l0)):
            c = np.vdot(Q0[:, j], Q1[:, j]) / (np.linalg.norm(Q0[:, j]) * np.linalg.norm(Q1[:, j]))
            assert abs(abs(c) - 1.0) < 1e-10

@pytest.mark.parametrize("neigs", [np.inf, 5])
def test_iar_device_eig_equals_host_lapack_route(na, monkeypatch, neigs):
    """eig(H_k) of every step on the device (csrc/hesseig.hip, batches on their own stream) against LAPACK on host worker
    threads (the round-3 route, NEP_IAR_EIG=host): same eigenpair count, eigenvalues to 1e-10 relative, error histories within
    1 % on the 8 best pairs of every iteration above 1e-12 (src/method_iar.jl:112-116).  neigs = 5: the throttled pipeline
    (batches = whatever is pending) stops at the same step.  No decomposition fell back to the host."""
    from nep_amd.linsolvers import _DeviceRefactor
    n, m = 9956, 70
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=neigs, v=np.ones(n), tol=1e-10)
    na.iar(nep, **kw); _DeviceRefactor.wait()
    res = {}
    for mode in ("host", "dev"):
        monkeypatch.setenv("NEP_IAR_EIG", mode)
        hist = []
        fb0 = na.iar.dev_eig_fallbacks
        lam, Q, V = na.iar(nep, errhist=hist, **kw)
        res[mode] = (np.asarray(lam), hist, na.iar.dev_eig_fallbacks - fb0)
    lh, hh, _ = res["host"]; ld, hd, fb = res["dev"]
    assert fb == 0
    assert len(lh) == len(ld) >= (5 if neigs == 5 else 20)
    assert len(hh) == len(hd)
    for x in ld:
        assert np.min(np.abs(lh - x)) <= 1e-10 * max(1.0, abs(x))
    cnt = 0
    for a, b in zip(hd, hh):
        a = np.sort(a)[:8]; b = np.sort(b)[:8]
        for x, y in zip(a, b):
            if x > 1e-12 and y > 1e-12:
                assert 0.99 < x / y < 1.01; cnt += 1
    assert cnt > 50


@pytest.mark.parametrize("n,m", [(9956, 40), (1310, 20)])
def test_iar_fused_finish_and_coefficient_product_is_bit_identical(na, monkeypatch, n, m):
    """step k's last kernel also forms step k + 1's coefficient product and block shift (k_orth_finish_vc, csrc/orth.hip; K1 of the
    next step is then the SpMV alone) against the separate kernels (NEP_IAR_FUSE_VC=0: k_orth_finish, then k_vc / the fused small-k
    SpMV + nep_iar_shift_scale): the product is summed in k_vc's order, but for k <= 8 the separate form runs the fused small-k SpMV, which
    sums differently -- so the two runs agree to rounding, not bitwise: eigenvalues to 1e-12, the six best backward errors of every
    iteration within 10 %; the fused form itself is bitwise repeatable"""
    from nep_amd.linsolvers import _DeviceRefactor
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    try:
        na.iar(nep, **kw)
    except na.NoConvergenceException:
        pass
    _DeviceRefactor.wait()
    res = {}
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("NEP_IAR_FUSE_VC", mode)
        hist = []
        try:
            lam, Q, V = na.iar(nep, errhist=hist, **kw)
        except na.NoConvergenceException as e:
            lam, Q = e.lam, e.v
        res.setdefault(mode, []).append((np.asarray(lam), np.asarray(na.to_host(Q)) if not isinstance(Q, np.ndarray) else Q, hist))
    (l0, q0, h0), = res["0"]; (l1, q1, h1), (l2, q2, h2) = res["1"]
    assert np.array_equal(l1, l2) and np.array_equal(q1, q2)                       # run to run: bitwise
    assert len(h0) == len(h1) == m and len(l0) == len(l1)
    assert np.abs(np.sort_complex(l0) - np.sort_complex(l1)).max() <= 1e-12 * max(1.0, np.abs(l0).max()) if len(l0) else True
    for a, b in zip(h0, h1):
        a = np.sort(a)[:6]; b = np.sort(b)[:6]
        for x, y in zip(a, b):
            if x > 1e-11 and y > 1e-11:
                assert 0.9 < x / y < 1.1


def test_two_concurrent_iar_calls_on_one_gpu(na):
    """two host threads run iar at the same time on their own NEP objects (one GPU, one eig stream, one check stream): the scratch
    of the device eigen-decompositions is checked out per call (two launches of a batch share it -- with a process-wide block call
    A's inverse iteration could run on call B's matrices, status words clean), so every result equals the single-threaded one"""
    import threading
    import torch
    n, m = 2000, 40
    neps = [na.nep_gallery("gun_spmf_scaled", n) for _ in range(2)]
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    ref = np.sort_complex(np.asarray(na.iar(neps[0], **kw)[0]))
    na.iar(neps[1], **kw)
    bad = []

    def work(nep):
        torch.cuda.set_device(0)
        for _ in range(6):
            try:
                lam = np.sort_complex(np.asarray(na.iar(nep, **kw)[0]))
                if len(lam) != len(ref) or np.abs(lam - ref).max() > 1e-9 * np.abs(ref).max():
                    bad.append("differs")
            except Exception as e:      # noqa: BLE001
                bad.append(repr(e)[:200])
    th = [threading.Thread(target=work, args=(neps[i],)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad
    import sys as _sys
    pool = _sys.modules["nep_amd.iar"]._EIG_WORK.get(torch.cuda.current_device(), [])
    assert len(pool) <= 2                                   # idle blocks kept per device: bounded


def test_iar_device_eig_failure_falls_back_to_lapack(na, monkeypatch):
    """a decomposition that reports a failure (forced here for one step: the QR status word of step 23, the inverse-iteration
    status word of step 31) is redone by LAPACK on the host and the run returns what the all-device run returns"""
    n, m = 2000, 40
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    monkeypatch.setenv("NEP_IAR_EIG", "dev")
    monkeypatch.setenv("NEP_IAR_NATIVE_RUN", "0")        # the step-at-a-time pipeline (the one-call route reports NEP_ERR_RETRY instead: next test)
    lam0, _, _ = na.iar(nep, **kw)
    # a refused LAUNCH of the eigenvalue kernel (a device that does not grant its LDS): every batch goes to the host, same result
    monkeypatch.setenv("NEP_IAR_EIG_LAUNCH_FAIL", "1")
    fb0 = na.iar.dev_eig_fallbacks
    lam1, _, _ = na.iar(nep, **kw)
    monkeypatch.delenv("NEP_IAR_EIG_LAUNCH_FAIL")
    assert na.iar.dev_eig_fallbacks - fb0 == m and len(lam1) == len(lam0)
    assert np.abs(np.sort_complex(np.asarray(lam1)) - np.sort_complex(np.asarray(lam0))).max() <= 1e-10 * np.abs(lam0).max()
    for fail in ("23", "-31"):
        monkeypatch.setenv("NEP_IAR_EIG_FAIL_AT", fail)
        fb0 = na.iar.dev_eig_fallbacks
        lam1, _, _ = na.iar(nep, **kw)
        assert na.iar.dev_eig_fallbacks == fb0 + 1
        assert len(lam1) == len(lam0) and np.allclose(np.sort_complex(lam1), np.sort_complex(lam0), rtol=1e-10, atol=0)


def test_iar_native_run_retry_routes(na, monkeypatch):
    """nep_iar_run (the one-call route) reports NEP_ERR_RETRY when a step's record asks for what the enqueued work did not do; the
    host then takes the route that can: checked solves (reason 1), the step-synchronous DGKS loop (2), the step-at-a-time pipeline
    with its LAPACK fallback (3).  Injected at one step each; the call returns what the clean run returns."""
    n, m = 2000, 40
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    r0 = na.iar.native_runs
    h0 = []
    lam0, Q0, _ = na.iar(nep, errhist=h0, **kw)
    assert na.iar.native_runs == r0 + 1 and len(h0) == m
    for kind, counter in ((1, "refinement_misses"), (2, "orth_pass_misses"), (3, "native_run_misses")):
        monkeypatch.setenv("NEP_IAR_RUN_FAIL_AT", "%d:17" % kind)
        c0 = getattr(na.iar, counter); r0 = na.iar.native_runs
        h1 = []
        lam1, Q1, _ = na.iar(nep, errhist=h1, **kw)
        assert getattr(na.iar, counter) == c0 + 1 and na.iar.native_runs == r0
        assert len(h1) == m and len(lam1) == len(lam0)
        _match(lam1, lam0, 1e-10)
    monkeypatch.delenv("NEP_IAR_RUN_FAIL_AT")
    # a device basis, a device eigenvector block, a finite neigs and a check every 3rd step through the same entry point
    lam2, Qd, V = na.iar(nep, return_device=True, **kw)
    assert Qd.shape == (len(lam0), n) and V.shape[0] == m
    Qh = na.to_host(Qd)                      # (the order of pairs with equal errors is not fixed from run to run: matched by eigenvalue)
    for i, x in enumerate(lam2):
        j = int(np.argmin(abs(lam0 - x)))
        assert abs(lam0[j] - x) <= 1e-10 * max(1.0, abs(x))
        a = Q0[:, j] / np.linalg.norm(Q0[:, j]); b = Qh[:, i] / np.linalg.norm(Qh[:, i])
        assert abs(abs(np.vdot(a, b)) - 1.0) < 1e-8
    h3 = []
    lam3, _, _ = na.iar(nep, sigma=0.0, gamma=1.0, maxit=m, neigs=4, v=np.ones(n), tol=1e-10, check_error_every=3, errhist=h3)
    assert len(lam3) == 4 and len(h3) < m // 3 + 1 and all(len(h) % 3 == 0 or len(h) == m for h in h3)
    with pytest.raises(na.NoConvergenceException) as ei:
        na.iar(nep, sigma=0.0, gamma=1.0, maxit=12, neigs=30, v=np.ones(n), tol=1e-10)
    assert len(ei.value.lam) == 12 and ei.value.v.shape == (n, 12)


def test_iar_native_run_chunks_as_hipgraphs(na, monkeypatch):
    """NEP_IAR_GRAPH=1 (opt-in; measured: no gain): the chunks of Arnoldi steps after the first are captured
    on the caller's (non-NULL) stream and replayed as hipGraphs, their events recorded behind the launch -- same eigenpairs, same
    error history as the plain launches; on the NULL stream, which cannot be captured, the switch changes nothing"""
    import torch
    n, m = 2000, 48
    from nep_amd.linsolvers import _DeviceRefactor
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    na.iar(nep, **kw)
    _DeviceRefactor.wait()
    na.iar(nep, **kw)
    side = torch.cuda.Stream()
    out = []
    for g in ("0", "1"):
        monkeypatch.setenv("NEP_IAR_GRAPH", g)
        h = []
        r0 = na.iar.native_runs
        with torch.cuda.stream(side):
            lam, Q, _ = na.iar(nep, errhist=h, **kw)
        side.synchronize()
        assert na.iar.native_runs == r0 + 1
        out.append((lam, np.concatenate(h)))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    lam2, _, _ = na.iar(nep, **kw)                      # NULL stream, switch still on
    assert np.array_equal(lam2, out[0][0])


@pytest.mark.parametrize("n,m", [(1310, 30), (2000, 64), (333, 40)])
def test_iar_native_run_poisoned_basis_small_sizes(na, monkeypatch, n, m):
    """the slack-only clearing of the Krylov basis (nep_iar_run) at sizes whose column ends fall anywhere inside the tiles of the
    Gram-Schmidt kernels: NaN-poisoned block + slack clear gives the bits of the fully zeroed run"""
    from nep_amd.linsolvers import _DeviceRefactor
    nep = na.nep_gallery("gun_spmf_scaled", n)
    kw = dict(sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    na.iar(nep, **kw)
    _DeviceRefactor.wait()            # (the pattern's device-LU plan is built behind the first call: both runs below must take the same factorisation route)
    na.iar(nep, **kw)
    out = {}
    for mode in ("NEP_IAR_FULL_ZERO", "NEP_IAR_POISON"):
        monkeypatch.setenv(mode, "1")
        r0 = na.iar.native_runs
        h = []
        lam, Q, _ = na.iar(nep, errhist=h, **kw)
        assert na.iar.native_runs == r0 + 1
        monkeypatch.delenv(mode)
        out[mode] = (lam, Q, np.concatenate(h))
    a, b = out["NEP_IAR_FULL_ZERO"], out["NEP_IAR_POISON"]
    assert np.all(np.isfinite(b[2])) and np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_transf_shift_and_scale_iar_qdep0(na):
    """test/transf.jl:44-52 on the device path: the recipe of config C2 (shift_and_scale + iar) on the in-tree sparse SPMF
    qdep0; residuals evaluated by the ORACLE on the original problem < sqrt(eps); eigenvalues equal the oracle's"""
    from oracle import gallery as og, neps as oneps, solvers as osol
    nep3 = na.nep_gallery("qdep0"); o3 = og.qdep0(); n = nep3.n
    sig, al = -3 + 0.3j, 0.9
    tr = na.shift_and_scale(nep3, shift=sig, scale=al)
    lam, V, _ = na.iar(tr, sigma=0, neigs=2, maxit=60, v=np.ones(n))
    for i in range(2):
        assert np.linalg.norm(o3.compute_Mlincomb(al * lam[i] + sig, V[:, i])) < np.sqrt(EPS)
    lo, Vo = osol.iar(oneps.shift_and_scale(o3, shift=sig, scale=al), sigma=0, neigs=2, maxit=60, v=np.ones(n))[:2]
    _match(lam, lo, 1e-9)


def test_augnewton_and_newton_inner_solver(na):
    """augnewton (method_newton.jl:262-345) from the start of test/newton.jl:16-17 reaches the dep0 eigenvalue that resinv
    finds; NewtonInnerSolver on a projected problem (test/inner_solves.jl:31) returns eigenpairs of the projected NEP"""
    from oracle import gallery as og
    dep = na.nep_gallery("dep0"); od = og.dep0()
    lam, v = na.augnewton(dep, lam=0.0, v=np.ones(5), maxit=30)
    assert abs(lam - (-0.15955391823299253)) < 1e-10
    assert np.linalg.norm(od.compute_Mlincomb(lam, v)) / np.linalg.norm(v) < 1e-12
    nep = na.nep_gallery("dep0", 50)
    pnep = na.create_proj_NEP(nep)
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((50, 6)))
    pnep.set_projectmatrices(Q, Q)
    lamv, Vp = na.inner_solve(na.NewtonInnerSolver(), pnep, lamv=np.array([0.0, 1.0]) + 0j, V=np.ones((6, 2)), tol=EPS * 100)
    Mp = lambda l: pnep.compute_Mder(l)
    assert min(np.linalg.norm(Mp(lamv[j]) @ Vp[:, j]) / np.linalg.norm(Vp[:, j]) for j in range(2)) < 1e-9


def test_iar_recorded_refinement_and_miss_fallback(na, monkeypatch):
    """the native iar step refines without reading omega back and records it; (i) the record is reviewed for every step and
    the settled sweep count is what the checked solves find, (ii) a review miss re-runs the call with checked solves and
    returns the same eigenpairs and error history"""
    from nep_amd.linsolvers import FactorizeLinSolver
    n, m = 1310, 30
    nep = na.nep_gallery("gun_spmf_scaled", n)
    monkeypatch.setenv("NEP_IAR_NATIVE_RUN", "0")        # the step-at-a-time pipeline: its host replays the rule in Python
    seen = []
    orig = FactorizeLinSolver.review_recorded

    def spy(self, w, plan, final_recorded=True):
        ok = orig(self, w, plan, final_recorded=final_recorded)
        seen.append((plan, [float(x) for x in w[:plan + 1]], ok, final_recorded))
        return ok
    monkeypatch.setattr(FactorizeLinSolver, "review_recorded", spy)
    h1 = []
    l1, Q1, _ = na.iar(nep, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10, errhist=h1)
    assert len(seen) == m and all(ok for _, _, ok, _ in seen)
    assert all(w[-1] <= 4 * np.finfo(float).eps for _, w, _, _ in seen)       # every kept iterate is converged
    assert seen[0][0] == 2 and seen[-1][0] <= 1                                 # two sweeps to start, then the settled count
    # (ii) force a miss on the 7th review
    count = [0]

    def miss(self, w, plan, final_recorded=True):
        count[0] += 1
        return orig(self, w, plan, final_recorded=final_recorded) and count[0] != 7
    monkeypatch.setattr(FactorizeLinSolver, "review_recorded", miss)
    h2 = []
    l2, Q2, _ = na.iar(nep, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10, errhist=h2)
    assert count[0] >= 7 and len(h2) == len(h1) == m
    assert len(l2) == len(l1)
    _match(l2, l1, 1e-10)
    for a, b in zip(h1, h2):
        for x, y in zip(a[:3], b[:3]):
            if x > 1e-12 and y > 1e-12:
                assert 0.2 < x / y < 5
    # (iii) the settled count travels with the NEP: the next solver starts with it (every step, not only the late ones)
    assert nep._refine_hint == 1
    del seen[:]
    monkeypatch.setattr(FactorizeLinSolver, "review_recorded", spy)
    l3, Q3, _ = na.iar(nep, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    assert len(seen) == m and all(ok and plan == 1 for plan, _, ok, _ in seen)
    # ... and with a settled count the backward error of the KEPT iterate is evaluated in every 8th step only (the policy of
    # FactorizeLinSolver.solve_dev: 7 of 8 solves on trust); omega of x_0 is still recorded in every step
    assert [fr for _, _, _, fr in seen] == [(j % 8 == 0) for j in range(1, m + 1)]
    assert all(w[0] > 4 * np.finfo(float).eps for _, w, _, _ in seen)
    _match(l3, l1, 1e-10)
    monkeypatch.setenv("NEP_REFINE_HINT", "0")
    del seen[:]
    na.iar(nep, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    assert seen[0][0] == 2


def test_iar_chebyshev_and_default_inner_solver(na):
    """iar_chebyshev on the device: the docstring eigenvalues of method_iar_chebyshev.jl:45-56 (1e-12), the SPMF and PEP
    versions of compute_y0_cheb against the oracle, a shifted / scaled run; then test/iar.jl:29-33 exactly: iar(dep0,
    proj_solve=true) with the DEFAULT inner solver (DEP -> iar_chebyshev on the normalised projected DEP), and the
    PEP default (polyeig of the projected problem)"""
    from oracle import gallery as og, solvers as osol, neps as oneps
    nep = na.nep_gallery("dep0", 100); onep = og.dep0(100)
    lam, V, _ = na.iar_chebyshev(nep, v=np.ones(100), tol=1e-5, neigs=3)
    ref = np.array([0.050462487848960284, -0.07708779190301127, 0.1503856540695659])
    assert np.max(abs(lam.real - ref)) < 1e-12 and np.max(abs(lam.imag)) < 1e-13
    lam2, V2, _ = na.iar_chebyshev(nep, v=np.ones(100), tol=1e-9, neigs=3, compute_y0_method="SPMF", a=-1.0, b=0.0)
    lo2 = osol.iar_chebyshev(onep, v=np.ones(100), tol=1e-9, neigs=3, compute_y0_method="SPMF", a=-1.0, b=0.0)[0]
    _match(lam2, lo2, 1e-9)
    lam4, V4, _ = na.iar_chebyshev(nep, v=np.ones(100), tol=1e-9, neigs=2, sigma=0.1, gamma=0.5, maxit=40)
    assert max(np.linalg.norm(onep.compute_Mlincomb(lam4[i], V4[:, i])) / np.linalg.norm(V4[:, i]) for i in range(2)) < 1e-7
    B = [np.array([[1.0, 3], [5, 6]]), np.array([[3.0, 4], [6, 6]]), np.eye(2)]
    lam3, V3, _ = na.iar_chebyshev(na.PEP(B), v=np.ones(2), tol=1e-10, neigs=2, maxit=20)
    lo3 = osol.iar_chebyshev(oneps.PEP(B), v=np.ones(2), tol=1e-10, neigs=2, maxit=20)[0]
    _match(lam3, lo3, 1e-8)
    # proj_solve with the default inner solver
    dep = na.nep_gallery("dep0"); od = og.dep0()
    lam, Q, _ = na.iar(dep, sigma=1.1, neigs=5, v=np.ones(5), maxit=100, tol=EPS * 100, errmeasure=na.ResidualErrmeasure(dep),
                       proj_solve=True)
    assert len(lam) == 5
    assert max(np.linalg.norm(od.compute_Mlincomb(lam[i], Q[:, i])) / np.linalg.norm(Q[:, i]) for i in range(5)) < EPS * 100
    # PEP original -> polyeig of the projected PEP
    rng = np.random.default_rng(4)
    Bp = [rng.standard_normal((30, 30)) for _ in range(3)]
    pep = na.PEP(Bp)
    pnep = na.create_proj_NEP(pep)
    Qb, _ = np.linalg.qr(rng.standard_normal((30, 4)))
    pnep.set_projectmatrices(Qb, Qb)
    lamp, Xp = na.inner_solve(na.DefaultInnerSolver(), pnep)
    Bq = [Qb.T @ Bi @ Qb for Bi in Bp]
    fin = np.isfinite(lamp)
    assert fin.sum() >= 6
    assert max(np.linalg.norm((Bq[0] + l * Bq[1] + l * l * Bq[2]) @ Xp[:, j]) for j, l in enumerate(lamp) if np.isfinite(l) and abs(l) < 1e3) < 1e-8


def test_ilan_vs_oracle(na):
    """infinite Lanczos on the device: the docstring eigenvalues of method_ilan.jl:41-52 are among the converged ones
    (1e-10), the Lanczos coefficients H and omega of the first 8 steps equal the oracle's (1e-8; later steps amplify
    round-off in both), Ritz extraction path, DEP == equivalent SPMF_NEP (test/ilan.jl:45-62), NoConvergence (:33-43)"""
    import warnings
    from oracle import gallery as og, solvers as osol
    nep = na.nep_gallery("dep_symm_double", 10); n = nep.n
    onep = og.dep_symm_double(10)
    out = na.ilan(nep, v=np.ones(n), tol=1e-5, neigs=12)
    lam, W = out[0], out[1]
    ref = np.array([0.03409997385842267, -0.03100798730589012, -0.0367653644764646])
    assert len(lam) == 12 and max(np.min(abs(lam - r)) for r in ref) < 1e-10
    assert max(np.linalg.norm(onep.compute_Mlincomb(lam[i], W[:, i])) / np.linalg.norm(W[:, i]) for i in range(12)) < 1e-2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oo = osol.ilan(onep, v=np.ones(n), tol=1e-5, neigs=12)
    assert np.linalg.norm(out[3][:9, :8] - oo[3][:9, :8]) <= 1e-8 * np.linalg.norm(oo[3][:9, :8])
    assert np.linalg.norm(out[4][:8] - oo[4][:8]) <= 1e-8 * np.linalg.norm(oo[4][:8])
    lam2 = na.ilan(nep, v=np.ones(n), tol=1e-5, neigs=3, proj_solve=False)[0]
    assert len(lam2) == 3 and np.min(abs(lam2 - ref[0])) < 1e-8
    # same problem as DEP and as SPMF_NEP: same Lanczos coefficients
    rng = np.random.default_rng(1)
    m = 60
    def symtri():
        d = rng.random(m); e = rng.random(m - 1)
        A = sp.diags([e, d, e], [-1, 0, 1]); return sp.csc_matrix(A + A.T)
    A1, A2 = symtri(), symtri()
    f = na.funcs
    nep1 = na.DEP([A1, A2], [0.0, 1.0])
    nep2 = na.SPMF_NEP([sp.identity(m, format="csc"), A1, A2], [f.Monomial(1) * (-1.0), f.one(), f.Exp(-1.0)])
    v0 = rng.random(m)
    o1 = na.ilan(nep1, neigs=np.inf, maxit=10, tol=EPS * 100, check_error_every=np.inf, v=v0)
    o2 = na.ilan(nep2, neigs=np.inf, maxit=10, tol=EPS * 100, check_error_every=np.inf, v=v0)
    assert np.linalg.norm(o1[3] - o2[3]) < 1e-6 and np.linalg.norm(o1[2] - o2[2]) < 1e-6
    with pytest.raises(na.NoConvergenceException):
        na.ilan(nep1, neigs=2, maxit=3, tol=EPS * 100, check_error_every=np.inf, v=v0, errmeasure=na.ResidualErrmeasure(nep1))


def test_nlar_gun_twin_vs_oracle(na):
    """test/nlar.jl:12-44 on the device (400-row gun twin, shift_and_scale SPMF, IARInnerSolver, residual sorter): the two
    eigenvalues equal the oracle's (1e-7), residual thresholds of the reference test; default sorter; NoConvergence (:66)"""
    import warnings
    from oracle import gallery as og, solvers as osol, neps as oneps
    n = 400
    onep = og.nlevp_native_gun(n)
    shift, scale = 250.0 ** 2, 330.0 ** 2 - 220.0 ** 2
    o1 = oneps.shift_and_scale(oneps.SPMF_NEP(onep.get_Av(), onep.get_fv()), shift=shift, scale=scale)
    nep1 = na.nep_gallery("gun_spmf_scaled", n)
    TOL = 1e-10
    kw = dict(tol=TOL, lam=0, maxit=100, neigs=2, R=0.01, v=np.ones(n), max_subspace=150, num_restart_ritz_vecs=8)
    D, X, hist = na.nlar(nep1, inner_solver_method=na.IARInnerSolver(), **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Do, Xo = osol.nlar(o1, inner_solver_method=osol.IARInnerSolver(), **kw)
    _match(D, Do, 1e-7)
    for i in range(2):
        lo = shift + scale * D[i]
        assert np.linalg.norm(onep.compute_Mlincomb(lo, X[:, i])) < np.sqrt(TOL) * 50
    D2, X2, _ = na.nlar(nep1, inner_solver_method=na.IARInnerSolver(), eigval_sorter=na.default_eigval_sorter, **dict(kw, neigs=1))
    assert np.linalg.norm(onep.compute_Mlincomb(shift + scale * D2[0], X2[:, 0])) < np.sqrt(TOL) * 50
    with pytest.raises(na.NoConvergenceException):
        na.nlar(nep1, tol=1e-20, maxit=3, neigs=3, v=np.ones(n), inner_solver_method=na.IARInnerSolver())


def test_jd_betcke_vs_oracle(na):
    """test/jd.jl:15-60 on the device with in-tree problems: random quadratic PEP (n=60, default inner solver = polyeig of
    the projected PEP), dep0(40) with the default (Chebyshev) inner solver, Galerkin projection, restart from a converged
    pair, the error cases; eigenvalues against the oracle"""
    import warnings
    from oracle import gallery as og, solvers as osol, neps as oneps
    rng = np.random.default_rng(0)
    n = 60
    B = [rng.standard_normal((n, n)) for _ in range(3)]
    pep = na.PEP(B); opep = oneps.PEP(B)
    lam, u = na.jd_betcke(pep, tol=1e-11, maxit=55, neigs=2, v=np.ones(n), lam=0, errmeasure=na.ResidualErrmeasure(pep))
    assert max(np.linalg.norm(opep.compute_Mlincomb(lam[i], u[:, i])) / np.linalg.norm(u[:, i]) for i in range(2)) < 1e-10
    dep = na.nep_gallery("dep0", 40); odep = og.dep0(40)
    lam, u = na.jd_betcke(dep, tol=1e-10, maxit=30, v=np.ones(40), lam=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lo, uo = osol.jd_betcke(odep, tol=1e-10, maxit=30, v=np.ones(40), lam=0)
    assert osol.DefaultErrmeasure(odep)(lam[0], u[:, 0]) < 1e-10
    _match(lam, lo, 1e-7)
    lam2, u2 = na.jd_betcke(dep, tol=1e-10, maxit=25, neigs=1, lam=lam[0], v=u[:, 0])          # converged before starting
    assert abs(lam2[0] - lam[0]) < 1e-12
    lam3, u3 = na.jd_betcke(dep, tol=1e-10, maxit=30, v=np.ones(40), lam=0, projtype="Galerkin", inner_solver_method=na.IARInnerSolver())
    assert osol.DefaultErrmeasure(odep)(lam3[0], u3[:, 0]) < 1e-10
    small = na.PEP([b[:10, :10] for b in B])
    with pytest.raises(ValueError):
        na.jd_betcke(small, tol=1e-10, maxit=60, v=np.ones(10))
    with pytest.raises(ValueError):
        na.jd_betcke(small, tol=1e-10, maxit=4, projtype="MYNOTDEFINED", v=np.ones(10))
    with pytest.raises(na.NoConvergenceException):
        na.jd_betcke(small, tol=1e-10, maxit=4, lam=10.0, neigs=1000, v=np.ones(10))


def test_projection_and_proj_solve(na):
    """Proj_SPMF_NEP (NEPTypes.jl:724-790): set / expand project matrices against NumPy on a sparse SPMF; then
    proj_solve=true in tiar (test/tiar.jl:70-84 at n=200) and iar (test/iar.jl:29-33) with IARInnerSolver: same eigenvalues
    as the oracle, residuals below the reference's thresholds"""
    from oracle import gallery as og, solvers as osol
    nep = na.nep_gallery("qdep0"); n = nep.n
    rng = np.random.default_rng(1)
    V = rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4)); W = rng.standard_normal((n, 4)) + 0j
    pnep = na.create_proj_NEP(nep)
    pnep.set_projectmatrices(W[:, :3], V[:, :3])
    Bref = [W[:, :3].conj().T @ (A @ V[:, :3]) for A in nep.get_Av()]
    for B, Br in zip(pnep.get_Av(), Bref):
        assert np.linalg.norm(B - Br) <= 1e-12 * np.linalg.norm(Br)
    pnep.expand_projectmatrices(W, V)
    Bref = [W.conj().T @ (A @ V) for A in nep.get_Av()]
    for B, Br in zip(pnep.get_Av(), Bref):
        assert B.shape == (4, 4) and np.linalg.norm(B - Br) <= 1e-12 * np.linalg.norm(Br)
    lam0 = 0.3 + 0.1j
    M = sum(f.derivs(lam0, 1)[0] * A for f, A in zip(nep.get_fv(), nep.get_Av()))
    assert np.linalg.norm(pnep.compute_Mder(lam0) - W.conj().T @ (M @ V)) <= 1e-11 * np.linalg.norm(M.toarray() if hasattr(M, "toarray") else M)
    # tiar with projected extraction
    m = 200
    depp = na.nep_gallery("dep0", m); odep = og.dep0(m)
    nn = np.linalg.norm(odep.compute_Mder(0), 2)
    errm = lambda l, v: np.linalg.norm(odep.compute_Mlincomb(l, np.asarray(v))) / nn
    kw = dict(sigma=0, gamma=3, neigs=3, v=np.ones(m), maxit=50, tol=np.sqrt(EPS), check_error_every=3, proj_solve=True)
    lam, Q = na.tiar(depp, inner_solver_method=na.IARInnerSolver(), errmeasure=errm, **kw)[:2]
    lo, Qo = osol.tiar(odep, inner_solver_method=osol.IARInnerSolver(), errmeasure=errm, **kw)[:2]
    assert len(lam) == 3 and errm(lam[0], Q[:, 0]) < np.sqrt(EPS) * 10
    _match(lam, lo, 1e-8)
    # iar with projected extraction
    dep = na.nep_gallery("dep0"); od = og.dep0()
    lam, Q, _ = na.iar(dep, sigma=1.1, neigs=5, v=np.ones(5), maxit=100, tol=EPS * 100, errmeasure=na.ResidualErrmeasure(dep),
                       proj_solve=True, inner_solver_method=na.IARInnerSolver())
    assert len(lam) == 5
    assert max(np.linalg.norm(od.compute_Mlincomb(lam[i], Q[:, i])) / np.linalg.norm(Q[:, i]) for i in range(5)) < EPS * 100


def test_tiar_dep0_kat(na):
    # test/tiar.jl:23-39,59-69,86-90 ; src/method_tiar.jl:37-45
    from oracle import gallery as og, solvers as osol
    nep = na.nep_gallery("dep0", 100); onep = og.dep0(100)
    R = na.ResidualErrmeasure(nep); oR = osol.ResidualErrmeasure(onep)
    lam, Q, Z, _ = na.tiar(nep, sigma=1.1, gamma=3, neigs=2, v=np.ones(100), maxit=50, tol=EPS * 100, errmeasure=R)
    assert len(lam) == 2
    for fused in (True, False):
        lam, Q, Z, _ = na.tiar(nep, sigma=1.1, gamma=3, neigs=np.inf, v=np.ones(100), maxit=50, tol=EPS * 100,
                               errmeasure=R, fused=fused)
        assert len(lam) == 7
        assert max(oR(lam[i], Q[:, i]) for i in range(7)) < EPS * 100
        Zh = na.to_host(Z)
        assert np.linalg.norm(Zh.conj().T @ Zh - np.eye(Zh.shape[1]), 2) < 1e-6
    lam, Q, _, _ = na.tiar(nep, v=np.ones(100), tol=1e-5, neigs=3)
    ref = np.array([-0.07708769561361105, 0.050462487743188206, 0.1503916927814904])
    assert np.allclose(np.sort(lam.real), ref, atol=1e-12)
    # tiar == iar
    kw = dict(sigma=1.1, gamma=3, neigs=3, v=np.ones(100), maxit=50, tol=1e-10)
    l1 = na.tiar(nep, **kw)[0]; l2 = na.iar(nep, **kw)[0]
    _match(l1, l2, 1e-6)
    with pytest.raises(na.NoConvergenceException):
        na.tiar(nep, sigma=2.0, gamma=3, neigs=4, v=np.ones(100), maxit=5, tol=EPS * 100)
    with pytest.raises(na.LostOrthogonalityException):
        na.tiar(na.nep_gallery("dep0"), maxit=30, v=np.ones(5))


def test_tiar_gun_twin_vs_oracle(na):
    from oracle import gallery as og, solvers as osol, neps as oneps
    n, m = 1310, 30
    onep = og.gun_spmf_scaled(n)
    lo, Qo, _, _ = osol.tiar(oneps.DerSPMF(onep, 0.0, m), maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10,
                             errmeasure=osol.StandardSPMFErrmeasure(onep))
    nep = na.nep_gallery("gun_spmf_scaled", n)
    lg, Qg, _, _ = na.tiar(nep, maxit=m, neigs=np.inf, v=np.ones(n), tol=1e-10)
    assert len(lg) == len(lo) and len(lg) >= 1
    _match(lg, lo, 1e-8)
    oE = osol.StandardSPMFErrmeasure(onep)
    assert max(oE(lg[i], Qg[:, i]) for i in range(len(lg))) < 1e-10


def test_resinv_dep0_c1(na):
    """config C1: nep_gallery("dep0") n=5, resinv(lam=0, v=ones).  From this start the inner scalar
    Newton iteration of compute_rf has no nearby root and wanders chaotically for its 80 allowed steps
    (compute_rf_wrapper.jl:31-39, bad_solution_allowed=true), so iterates are not comparable between any
    two floating-point implementations; the converged pair is."""
    from oracle import gallery as og, solvers as osol
    nep = na.nep_gallery("dep0"); onep = og.dep0()
    lam, v = na.resinv(nep, lam=0, v=np.ones(5))
    assert lam.real == pytest.approx(-0.1595539182329811, rel=1e-10) and abs(lam.imag) < 1e-12
    assert osol.DefaultErrmeasure(onep)(lam, v) < EPS * 100
    # well-conditioned start (near the eigenpair): iterate-by-iterate agreement with the oracle
    lo, vo = osol.resinv(onep, lam=0, v=np.ones(5))
    v0 = vo + 0.05 * np.arange(1, 6)
    hg = []; ho = []
    l1, v1 = na.resinv(nep, lam=lo + 0.05, v=v0, hist=hg)
    l2, v2 = osol.resinv(onep, lam=lo + 0.05, v=v0, hist=ho)
    assert len(hg) == len(ho) and len(hg) > 3
    for (k1, e1, a), (k2, e2, b) in zip(hg, ho):
        assert abs(a - b) <= 1e-9 * max(1, abs(b))
        if e2 > 1e-12:
            assert e1 == pytest.approx(e2, rel=1e-4)
    assert abs(l1 - l2) < 1e-12
    # armijo path (method_newton.jl:598-609)
    lam2, v2 = na.resinv(nep, lam=lo + 0.05, v=v0, armijo_factor=0.5)
    assert abs(lam2 - l1) < 1e-10


def test_quasinewton_qdep0_history(na):
    # src/errmeasure.jl:156-169: sparse compute_Mlincomb (startder 0 and 1), StandardSPMFErrmeasure and a reused
    # sparse factorisation -- the reference's printed 9-step history
    nep = na.nep_gallery("qdep0")
    hist = []
    lam, v = na.quasinewton(nep, lam=-1, v=np.ones(1000), errmeasure=na.StandardSPMFErrmeasure(nep), tol=1e-10,
                            hist=hist)
    ref = [(0.022010375110869937, -1.0), (0.002515422247048546, -0.7063330111559607),
           (0.000892354247568813, -0.8919579082730457), (5.445678793151584e-5, -1.0097584042560848),
           (6.649967517409105e-7, -1.0023823873044), (1.0557281809769784e-8, -1.0024660870524031),
           (6.420125566431444e-9, -1.0024677891861997), (3.181093707909799e-10, -1.0024669496893164),
           (2.6368050026394416e-11, -1.0024669918249076)]
    assert len(hist) == 9
    for (k, err, l), (eref, lref) in zip(hist, ref):
        assert l.real == pytest.approx(lref, rel=1e-10)
        assert err == pytest.approx(eref, rel=1e-4)
    assert hist[0][1] == pytest.approx(ref[0][0], rel=1e-13)


@pytest.mark.parametrize("sched", ["block", "old"])
def test_lu_schedule_shortens_the_dependency_chain(na, sched, monkeypatch):
    from oracle import gallery as og
    import scipy.sparse as sp
    if sched == "old":
        monkeypatch.setenv("NEP_LU_SCHED", "old")
    A = sp.csc_matrix(og.gun_spmf_scaled(2620).compute_Mder(0.0), dtype=complex)
    lu = na.DeviceLU(A, permc_spec="MMD_AT_PLUS_A")
    if sched == "old":
        assert not lu.block_schedule and lu.tail >= 64 and lu.levL < lu.levL_full and lu.levU < lu.levU_full
    else:
        assert lu.block_schedule and lu.levels <= 6 and lu.blocks >= 10 and lu.mid_block <= 256
    rng = np.random.default_rng(0)
    b = rng.standard_normal(2620) + 1j * rng.standard_normal(2620)
    x = na.to_host(lu.solve(na.to_dev(b)))[:, 0]
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b) * 10
    assert lu.launches_last_solve() < (200 if sched == "old" else 20)


def test_beyn_dep0_kat(na):
    # test/beyn.jl:15-46 (N=1000 default)
    from oracle import gallery as og
    nep = na.nep_gallery("dep0"); onep = og.dep0()
    lam, V = na.contour_beyn(nep, radius=1, neigs=1, sanity_check=False)
    M = onep.compute_Mder(lam[0])
    assert np.linalg.svd(M, compute_uv=False).min() < EPS * 1000
    assert np.linalg.norm(onep.compute_Mlincomb(lam[0], V[:, 0])) < EPS * 500
    lam, V = na.contour_beyn(nep, sigma=0.2, radius=1.0, neigs=4, sanity_check=False)
    assert len(lam) == 3


def test_beyn_gun_twin_vs_oracle(na):
    """config C4 at reduced size: same probe block, same count, eigenvalues agree, residuals below tol."""
    from oracle import gallery as og, solvers as osol
    n, k, N = 1310, 16, 32
    onep = og.gun_spmf(n)
    nep = na.nep_gallery("gun_spmf", n)
    Vh = na.probe_block(n, k)
    kw = dict(sigma=250.0 ** 2, radius=1.2e4, N=N, k=k, neigs=10 ** 6, tol=1e-6, sanity_check=True)
    io = {}; ig = {}
    lo, Vo = osol.contour_beyn(onep, Vh=Vh, info=io, **kw)
    lg, Vg = na.contour_beyn(nep, Vh=Vh, info=ig, **kw)
    assert ig["p"] == io["p"]
    assert len(lg) == len(lo) and len(lg) >= 1
    _match(lg, lo, 1e-7)
    oE = osol.StandardSPMFErrmeasure(onep)
    assert max(oE(lg[i], Vg[:, i]) for i in range(len(lg))) < 1e-6


def test_beyn_device_tail_equals_host_tail(na, monkeypatch):
    """the dense tail of contour_beyn with the moments left on the device (A0 = Q R by DGKS on the device, svd of the k x k R, the
    k x k Gram block Q^H A1 from the device, eigenvectors by K7; contour._beyn_tail_device) against the reference's own order of
    operations on the host (svd of the n x k block, method_beyncontour.jl:114-128): same rank, singular values to 1e-10 relative
    (above the rank threshold), same eigenvalues to 1e-9, eigenvectors with residuals below tol; A0 has rank < k here (the block
    the QR orthonormalises has columns of pure noise)"""
    n, k, N = 1310, 16, 32
    nep = na.nep_gallery("gun_spmf", n)
    Vh = na.probe_block(n, k)
    kw = dict(sigma=250.0 ** 2, radius=1.2e4, N=N, k=k, neigs=10 ** 6, tol=1e-6, sanity_check=True)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NEP_BEYN_DEVICE_TAIL", mode)
        info = {"moments": mode == "0"}
        lam, V = na.contour_beyn(nep, Vh=Vh, info=info, **kw)
        res[mode] = (np.asarray(lam), V, info)
    (l0, V0, i0), (l1, V1, i1) = res["0"], res["1"]
    assert i1["A0"] is None and i0["A0"] is not None              # the device tail downloaded no n x k block
    assert i0["p"] == i1["p"] < k
    p = i0["p"]
    assert np.abs(i0["S"][:p] - i1["S"][:p]).max() <= 1e-10 * i0["S"][0]
    assert len(l0) == len(l1) >= 1
    _match(l1, l0, 1e-9)
    from oracle import gallery as og, solvers as osol
    oE = osol.StandardSPMFErrmeasure(og.gun_spmf(n))
    assert max(oE(l1[i], V1[:, i]) for i in range(len(l1))) < 1e-6


def test_beyn_first_call_seeds_the_device_plan(na, monkeypatch):
    """a process that ONLY runs contour_beyn.  Round 6: the FIRST call on a pattern factorises one node on the host, builds the
    pattern's device-factorisation plan at once and factorises the other N - 1 nodes on the GPU in one batch (cold call of C4
    0.47 -> 0.17 s); NEP_BEYN_COLD_PLAN=0 is the earlier route (all nodes of the first call in the host workers, the first of those
    factorisations seeds the plan, the second call is the first batched one).  Same eigenvalues every way."""
    from nep_amd.linsolvers import _DeviceRefactor
    if not _DeviceRefactor.enabled():
        pytest.skip("device numeric factorisation switched off")
    n, k, N = 1310, 16, 16
    nep = na.nep_gallery("gun_spmf", n)
    Vh = na.probe_block(n, k)
    kw = dict(sigma=250.0 ** 2, radius=1.2e4, N=N, k=k, neigs=10 ** 6, tol=1e-6)
    ref = None
    for cold in ("1", "0"):
        monkeypatch.setenv("NEP_BEYN_COLD_PLAN", cold)
        _DeviceRefactor.clear()
        l1, V1 = na.contour_beyn(nep, Vh=Vh, **kw)
        _DeviceRefactor.wait()
        plans = [p for p in _DeviceRefactor.plans.values() if p["state"] == "ready"]
        assert len(plans) == 1
        first = plans[0]["uses"] + plans[0]["fails"]
        assert first == (N - 1 if cold == "1" else 0), (cold, first)
        l2, V2 = na.contour_beyn(nep, Vh=Vh, **kw)
        assert plans[0]["uses"] + plans[0]["fails"] == first + N and plans[0]["uses"] >= first + N - 2
        assert len(l1) == len(l2) and len(l1) >= 1
        _match(l2, l1, 1e-8)
        if ref is None:
            ref = l1
        else:
            _match(l1, ref, 1e-8)


def test_block_SS_dep0_kat_and_gun_twin(na):
    """test/contour_block_SS.jl:9-24 on the device path (three variants), then a sparse gun twin against the oracle with
    the same probe blocks: same numerical rank, eigenvalues inside the contour agree to 1e-7 relative."""
    from oracle import gallery as og, solvers as osol
    nep = na.nep_gallery("dep0", 3); onep = og.dep0(3)
    for kw in (dict(radius=1.0, K=3), dict(radius=[1.0, 2.0], K=3), dict(radius=1.0, K=4, Shat_mode="JSIAM")):
        lam, V = na.contour_block_SS(nep, N=1000, sigma=0.1, k=3, **kw)
        assert np.linalg.norm(onep.compute_Mlincomb(lam[0], V[:, 0])) < np.sqrt(EPS)
        lo, Vo = osol.contour_block_SS(onep, N=1000, sigma=0.1, k=3, **kw)
        _match(np.sort_complex(lam), np.sort_complex(lo), 1e-9)
    # radius-normalised moments (JSIAM mode): the rank gap of the Hankel matrix is 1e-4 -> 1e-14 on this problem
    n, L, K, N = 1310, 8, 4, 64
    onep = og.gun_spmf(n); nep = na.nep_gallery("gun_spmf", n)
    U, V = na.contour.probe_block_uniform(n, L)
    kw = dict(sigma=250.0 ** 2, radius=1.2e4, N=N, k=L, K=K, rank_drop_tol=1e-10, Shat_mode="JSIAM")
    io = {}; ig = {}
    lo, Vo = osol.contour_block_SS(onep, U=U, V=V, info=io, **kw)
    lg, Vg = na.contour_block_SS(nep, U=U, V=V, info=ig, **kw)
    assert ig["mprime"] == io["mprime"] and len(lg) == len(lo)
    oE = osol.StandardSPMFErrmeasure(onep)
    eg = np.array([oE(lg[i], Vg[:, i]) for i in range(len(lg))])
    eo = np.array([oE(lo[i], Vo[:, i]) for i in range(len(lo))])
    assert io["mprime"] == 7 and eg.max() < 1e-12 and eo.max() < 1e-12
    _match(lg, lo, 1e-10)


def test_wep_mlincomb_vs_oracle(na):
    # test/wep_small.jl:13-22 (SPMF == WEP_FD); here device WEP == oracle WEP_FD == oracle literal SPMF
    from oracle import wep as ow
    nep = na.nep_gallery("WEP", nx=11, nz=7, benchmark_problem="TAUSCH")
    o = ow.WEP_FD(11, 7, "TAUSCH")
    lam = -1.3 - 0.31j
    v1 = nep.compute_Mlincomb(lam, np.ones(nep.n)); v2 = o.compute_Mlincomb(lam, np.ones(o.n))
    assert np.linalg.norm(v1 - v2) / np.linalg.norm(v2) < 1e-13
    rng = np.random.default_rng(0)
    V = rng.standard_normal((nep.n, 5)) + 1j * rng.standard_normal((nep.n, 5))
    a = np.array([1.0, 0.5, 0.0, -2.0, 0.3])
    z1 = nep.compute_Mlincomb(lam, V, a); z2 = o.compute_Mlincomb(lam, V, a)
    assert np.linalg.norm(z1 - z2) / np.linalg.norm(z2) < 1e-12
    z1 = nep.compute_Mlincomb(lam, V[:, :2], a[:2], 1); z2 = o.compute_Mlincomb(lam, V[:, :2], a[:2], 1)
    assert np.linalg.norm(z1 - z2) / np.linalg.norm(z2) < 1e-12
    # residual error measure incl. corner term
    e = na.estimate_error(na.ResidualErrmeasure(nep), lam, V[:, 0])
    assert e == pytest.approx(np.linalg.norm(o.compute_Mlincomb(lam, V[:, 0])) / np.linalg.norm(V[:, 0]), rel=1e-11)


def test_wep_reference_eigenvalue(na):
    """test/wep_small.jl:31-36,73-76: JARLEBRING nx=109 nz=105, iar(sigma=-3-3.5i, neigs=3, maxit=100, tol=1e-8)
    finds lambda_ref = -2.743228671961724-3.1439375599649972i to 1e-10; tiar finds it too."""
    lref = -2.743228671961724 - 3.1439375599649972j
    nep = na.nep_gallery("WEP", nx=109, nz=105, benchmark_problem="JARLEBRING")
    n = nep.n
    v0 = np.ones(n) / np.sqrt(n)
    lam, Q, _ = na.iar(nep, sigma=-3 - 3.5j, neigs=3, maxit=100, v=v0, tol=1e-8)
    assert len(lam) == 3 and min(abs(lref - lam)) < 1e-10
    from oracle import wep as ow, solvers as osol
    o = ow.WEP_FD(109, 105, "JARLEBRING")
    R = osol.ResidualErrmeasure(o)
    assert max(R(lam[i], Q[:, i]) for i in range(3)) < 1e-8          # independent host re-evaluation
    lam2, Q2, _, _ = na.tiar(nep, sigma=-3 - 3.5j, neigs=3, maxit=100, v=v0, tol=1e-8)
    assert len(lam2) == 3 and min(abs(lref - lam2)) < 1e-10
    assert max(R(lam2[i], Q2[:, i]) for i in range(3)) < 1e-8


def _gun_r1_setup(n):
    """test/rk_helper/gun_test_utils.jl:6-32 (target set, nodes, pole candidates) for the gun problem"""
    gam = 300.0 ** 2 - 200.0 ** 2; mu = 250.0 ** 2; sigma2 = 108.8774
    xmin = gam * (-1) + mu; xmax = gam * 1 + mu
    npts = 1000
    th = np.linspace(0, np.pi, int(round(np.pi / 2 * npts)) + 2)
    halfcircle = xmin + (xmax - xmin) * (np.exp(1j * th) / 2 + .5)
    Sigma = np.concatenate([halfcircle, [xmin]])
    Z = np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3])
    nodes = gam * Z + mu
    Xi = -10.0 ** np.linspace(-8, 8, 10000) + sigma2 ** 2
    v = np.random.Generator(np.random.Philox(1)).standard_normal(n) + 0j
    return Sigma, Xi, nodes, v


def test_nleigs_basic_kat(na):
    # test/nleigs/nleigs_basic.jl:11-19,42-47 ; src/method_nleigs.jl:44-50
    from oracle import neps as oneps, nleigs as onl, gallery as og
    B = [np.array([[1., 3], [5, 6]]), np.array([[3., 4], [6, 6]]), np.eye(2)]
    Sigma = np.array([-10 - 2j, 10 - 2j, 10 + 2j, -10 + 2j])
    info = {}
    lam, X, res = na.nleigs(na.PEP(B), Sigma, maxit=10, v=np.ones(2) + 0j, blksize=5, info=info)
    lo, Xo, ro = onl.nleigs(oneps.PEP(B), Sigma, maxit=10, v=np.ones(2) + 0j, blksize=5)
    assert len(lam) == 4 and len(lo) == 4
    _match(lam, lo, 1e-9)
    o = oneps.PEP(B)
    assert max(np.linalg.norm(o.compute_Mlincomb(lam[i], X[:, i])) for i in range(4)) < 1e-5
    assert info["kconv"] == 6 and info["nfact"] == 4
    cB = [b + 1j * np.eye(2) for b in B]
    lam, X, res = na.nleigs(na.PEP(cB), Sigma, maxit=10, v=np.ones(2) + 0j, blksize=5)
    assert len(lam) == 3
    d = na.nep_gallery("dep0"); od = og.dep0()
    lam, X, res = na.nleigs(d, np.array([1 + 1j, 1 - 1j, -1 - 1j, -1 + 1j]), v=np.ones(5) + 0j)
    assert len(lam) >= 2
    assert max(np.linalg.norm(od.compute_Mlincomb(lam[i], X[:, i])) for i in range(len(lam))) < 1e-10


def test_nleigs_scalar_isfunm_false(na):
    """test/nleigs/nleigs_scalar.jl:9-34 on the device path (n = 1, user functions given as callables, divided
    differences by differencing): 1 eigenvalue with the polynomial approach, 3 with poles; values against the oracle KAT"""
    import scipy.linalg as sla
    fsqrt = lambda S: np.sqrt(S + 0j) if np.ndim(S) == 0 else sla.sqrtm(np.asarray(S, dtype=complex))
    fsin = lambda S: np.sin(2 * (S + 0j)) if np.ndim(S) == 0 else sla.sinm(2 * np.asarray(S, dtype=complex))
    nep = na.SPMF_NEP([np.array([[0.2]]), np.array([[-0.6]])], [fsqrt, fsin])
    Sig = np.array([0.01, 4], dtype=complex)
    lam, X, res = na.nleigs(nep, Sig, maxit=100, v=np.ones(1) + 0j, leja=2, isfunm=False)
    assert len(lam) == 1 and abs(lam[0] - 1.37036708) < 1e-7
    lam, X, res = na.nleigs(nep, Sig, Xi=-np.logspace(-6, 5, 10000), maxit=100, v=np.ones(1) + 0j, leja=2, isfunm=False)
    assert len(lam) == 3 and np.allclose(np.sort(lam.real), [0.02780643, 1.37036708, 3.47695453], atol=1e-7)
    assert max(abs(0.2 * np.sqrt(l) - 0.6 * np.sin(2 * l)) for l in lam) < 1e-10


def test_nleigs_nep_types(na):
    """test/nleigs/nleigs_nep_types.jl:31-46: the same quadratic problem as SPMF_NEP, PEP, PEP + SPMF and
    PEP + LowRankFactorizedNEP -> 4 eigenvalues each, all equal (the custom non-SPMF NEP type is out of scope)"""
    import scipy.sparse as sp
    f = na.funcs
    B = [np.array([[1.0, 3], [5, 6]]), np.array([[3.0, 4], [6, 6]])]; Cm = [np.eye(2)]
    Sig = [-10.0 - 2j, 10 - 2j, 10 + 2j, -10 + 2j]
    problems = [
        na.SPMF_NEP(B + Cm, [f.one(), f.ident(), f.Monomial(2)]),
        na.PEP(B + Cm),
        na.SumNEP(na.PEP(B), na.SPMF_NEP(Cm, [f.Monomial(2)])),
        na.SumNEP(na.PEP(B), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(sp.csc_matrix(Cm[0]), f.Monomial(2))])),
    ]
    ref = None
    for nep in problems:
        lam, X, res = na.nleigs(nep, Sig, maxit=10, v=np.ones(2) + 0j, blksize=5)
        assert len(lam) == 4
        M = lambda l: B[0] + l * B[1] + l * l * Cm[0]
        assert max(np.linalg.norm(M(lam[i]) @ X[:, i]) for i in range(4)) < 1e-5
        if ref is None:
            ref = lam
        _match(lam, ref, 1e-9)


def test_nleigs_static_and_details_vs_oracle(na):
    """test/nleigs/nleigs_basic.jl:28-73 on the device path: static variant (warning + 4 eigenvalues), return_details
    (0 / 3 / 4 eigenvalues, history consistent with the returned values); each case against the oracle (1e-9); then the
    static variant on a sparse gun twin against the oracle."""
    import warnings
    from oracle import nleigs as onl, neps as oneps, gallery as og
    B = [np.array([[1.0, 3], [5, 6]]), np.array([[3.0, 4], [6, 6]]), np.eye(2)]
    Sig = [-10.0 - 2j, 10 - 2j, 10 + 2j, -10 + 2j]
    pep = na.PEP(B); opep = oneps.PEP(B)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lam, X, _ = na.nleigs(pep, Sig, maxit=10, v=np.ones(2) + 0j, maxdgr=5, blksize=5, static=True)
        assert any("Linearization not converged" in str(x.message) for x in w)
    lo, Xo, _ = onl.nleigs(opep, Sig, maxit=10, v=np.ones(2) + 0j, maxdgr=5, blksize=5, static=True)
    assert len(lam) == 4
    _match(lam, lo, 1e-9)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        lam, X, _, d = na.nleigs(pep, Sig, maxit=5, v=np.ones(2) + 0j, blksize=5, return_details=True)
    assert len(lam) == 0
    Bc = [b + 1j * np.eye(2) for b in B]
    lam, X, _, d = na.nleigs(na.PEP(Bc), Sig, maxit=10, v=np.ones(2) + 0j, blksize=5, return_details=True)
    lo, Xo, _, do = onl.nleigs(oneps.PEP(Bc), Sig, maxit=10, v=np.ones(2) + 0j, blksize=5, return_details=True)
    assert len(lam) == 3
    _match(lam, lo, 1e-9)
    lam, X, res, d = na.nleigs(pep, Sig, maxit=10, v=np.ones(2) + 0j, blksize=5, return_details=True)
    lo, Xo, reso, do = onl.nleigs(opep, Sig, maxit=10, v=np.ones(2) + 0j, blksize=5, return_details=True)
    assert len(lam) == 4 and d.Lam.shape == do.Lam.shape and d.kconv == do.kconv
    l2 = d.Lam[:, -1]; r2 = d.Res[:, -1]
    conv = (r2 < 1e-12) & onl.in_Sigma(l2, np.asarray(Sig, dtype=complex), 0)
    assert conv.sum() == 4 and len(set(np.round(np.concatenate([lam, l2[conv]]), 8))) == 4
    # static variant on a sparse gun twin (leja nodes in both phases are replaced by given nodes: variant S set-up)
    n = 400
    onep = og.nlevp_native_gun(n); nep = na.nep_gallery("nlevp_native_gun", n)
    gam, mu = 300.0 ** 2 - 200.0 ** 2, 250.0 ** 2
    xmin, xmax = mu - gam, mu + gam
    half = xmin + (xmax - xmin) * (np.exp(1j * np.linspace(0, np.pi, 202)) / 2 + 0.5)
    Sg = np.concatenate([half, [xmin]])
    nodes = gam * np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3]) + mu
    Xi = -np.logspace(-8, 8, 2000) + 108.8774 ** 2
    v0 = np.cos(np.arange(n)) + 0j
    kw = dict(Xi=Xi, minit=30, maxit=50, v=v0, nodes=nodes, static=True, tol=1e-8)
    io = {}; ig = {}
    lo, Xo, ro = onl.nleigs(onep, Sg, info=io, **kw)
    lg, Xg, rg = na.nleigs(nep, Sg, info=ig, **kw)
    assert ig["kconv"] == io["kconv"] and len(lg) == len(lo) and len(lg) >= 1
    _match(lg, lo, 1e-7)


def test_nleigs_gun_twin_r1_vs_oracle(na):
    """config C3 at reduced size: gun in native PEP+SPMF form, variant R1 (leja=0, 5 cyclic nodes, reusefact=2)"""
    from oracle import gallery as og, nleigs as onl, solvers as osol
    n, maxit = 1310, 40
    Sigma, Xi, nodes, v = _gun_r1_setup(n)
    onep = og.nlevp_native_gun(n)
    io = {}; ig = {}
    oE = osol.StandardSPMFErrmeasure(onep)
    lo, Xo, ro = onl.nleigs(onep, Sigma, Xi=Xi, maxit=maxit, v=v, leja=0, nodes=nodes, reusefact=2, errmeasure=oE,
                            tol=1e-10, info=io)
    nep = na.nep_gallery("nlevp_native_gun", n)
    lg, Xg, rg = na.nleigs(nep, Sigma, Xi=Xi, maxit=maxit, v=v, leja=0, nodes=nodes, reusefact=2,
                           errmeasure=na.StandardSPMFErrmeasure(nep), tol=1e-10, info=ig)
    assert ig["nfact"] == io["nfact"] == 5                    # 5 cached factorisations (linsolvercache.jl)
    assert len(lg) == len(lo) and len(lg) >= 1
    _match(lg, lo, 1e-8)
    assert max(oE(lg[i], Xg[:, i]) for i in range(len(lg))) < 1e-10


def test_gmres_linsolver(na):
    """docs/src/tutorial_linsolve.md / test/newlinsolve.jl: tridiagonal SPMF n=100, GMRES with the diagonal of M(lam0)
    as left preconditioner vs the factorised solver."""
    import scipy.sparse as sp
    n = 100; alpha = 0.01
    A = sp.diags([np.ones(n), alpha * np.ones(n - 1), alpha * np.ones(n - 1)], [0, 1, -1], format="csc")
    B = sp.identity(n, format="csc"); Cm = sp.diags(np.arange(1, n + 1) / n, format="csc")
    nep = na.SPMF_NEP([A, B, Cm], [na.funcs.one(), na.funcs.ident(), na.funcs.Exp(1.0)])
    lam0 = -1.02
    M = sp.csc_matrix(nep.compute_Mder(lam0))
    b = np.arange(1, n + 1) + 1j
    creator = na.GMRESLinSolverCreator(Pl=M.diagonal(), tol=1e-12)
    s = na.create_linsolver(creator, nep, lam0)
    x = na.lin_solve(s, b)
    assert np.linalg.norm(M @ x - b) <= 1e-10 * np.linalg.norm(b)
    assert 0 < s.iterations <= n
    # no preconditioner: full GMRES at the (nearly singular) shift, restarted GMRES(5) at a well-conditioned one
    x2 = na.lin_solve(na.create_linsolver(na.GMRESLinSolverCreator(tol=1e-12, restart=n), nep, lam0), b)
    assert np.linalg.norm(M @ x2 - b) <= 1e-8 * np.linalg.norm(b)
    M3 = sp.csc_matrix(nep.compute_Mder(3.0))
    s3 = na.create_linsolver(na.GMRESLinSolverCreator(tol=1e-12, restart=5), nep, 3.0)
    x3 = na.lin_solve(s3, b)
    assert np.linalg.norm(M3 @ x3 - b) <= 1e-10 * np.linalg.norm(b) and s3.iterations > 5
    # as the linear solver of a NEP driver
    l1, v1 = na.quasinewton(nep, lam=lam0, v=np.ones(n), tol=1e-12)
    l2, v2 = na.quasinewton(nep, lam=lam0, v=np.ones(n), tol=1e-12, linsolvercreator=na.GMRESLinSolverCreator(Pl=M.diagonal(), tol=1e-12))
    assert abs(l1 - l2) < 1e-10


def test_nleigs_lowrank_gun_vs_oracle(na):
    """gun_nep() of test/rk_helper/gun_test_utils.jl:37-43 (PEP + LowRankFactorizedNEP, ranks 19 + 65) through variant
    R1 (test/nleigs/nleigs_gun_variant_r1.jl:15) on a reduced gun problem: the compressed device run (Krylov vectors of
    n + 84 N rows) against the compressed oracle run (eigenvalues 1e-8 relative) and against the device run with full blocks"""
    from oracle import neps as on, nleigs as onl, solvers as osol
    n = 1310
    K, M, W1, W2 = na.gallery.gun_matrices(n)
    s2 = na.gallery.GUN_SIGMA2
    fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -s2 ** 2)]
    ofv = [on.f_isqrt(0.0), on.f_isqrt(-s2 ** 2)]
    full = na.SumNEP(na.PEP([K, -M]), na.SPMF_NEP([W1, W2], fv))
    lowr = na.SumNEP(na.PEP([K, -M]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]), na.LowRankMatrixAndFunction(W2, fv[1])]))
    olr = on.SumNEP(on.PEP([K, -M]), on.LowRankFactorizedNEP([on.LowRankMatrixAndFunction(W1, ofv[0]), on.LowRankMatrixAndFunction(W2, ofv[1])]))
    assert lowr.nep2.rank == 84 and olr.nep2.rank == 84
    gam = 300.0 ** 2 - 200.0 ** 2; mu = 250.0 ** 2
    th = np.linspace(0, np.pi, int(round(np.pi / 2 * 1000)) + 2)
    Sig = np.concatenate([(mu - gam) + 2 * gam * (np.exp(1j * th) / 2 + .5), [mu - gam]])
    nodes = gam * np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3]) + mu
    Xi = -10.0 ** np.linspace(-8, 8, 10000) + s2 ** 2
    v = np.random.default_rng(1).standard_normal(n) + 0j
    kw = dict(Xi=Xi, maxit=60, v=v, leja=0, nodes=nodes, reusefact=2)
    info = {}
    lam, X, res = na.nleigs(lowr, Sig, errmeasure=na.StandardSPMFErrmeasure(lowr), info=info, **kw)
    assert info["vrows"] == n + 84 * 61 and info["nfact"] == 5
    lo, Xo, ro = onl.nleigs(olr, Sig, errmeasure=osol.StandardSPMFErrmeasure(olr), **kw)
    assert len(lam) >= 5
    _match(lam, lo, 1e-8)
    lf, Xf, rf = na.nleigs(full, Sig, errmeasure=na.StandardSPMFErrmeasure(full), **kw)
    _match(lam, lf, 1e-8)
    E = osol.StandardSPMFErrmeasure(olr)
    assert max(E(lam[i], X[:, i]) for i in range(len(lam))) < 1e-10


def test_nleigs_first_call_seeds_the_device_plan(na):
    """a process that ONLY runs nleigs: the shifts of the first call are factorised on the host (prefetch thread) and the first of
    those factorisations seeds the pattern's device-factorisation plan; the second call factorises its shifts on the GPU and
    returns the same eigenvalues"""
    from nep_amd.linsolvers import _DeviceRefactor
    if not _DeviceRefactor.enabled():
        pytest.skip("device numeric factorisation switched off")
    n = 1310
    K, M, W1, W2 = na.gallery.gun_matrices(n)
    s2 = na.gallery.GUN_SIGMA2
    fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -s2 ** 2)]
    lowr = na.SumNEP(na.PEP([K, -M]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]), na.LowRankMatrixAndFunction(W2, fv[1])]))
    gam = 300.0 ** 2 - 200.0 ** 2; mu = 250.0 ** 2
    th = np.linspace(0, np.pi, int(round(np.pi / 2 * 1000)) + 2)
    Sig = np.concatenate([(mu - gam) + 2 * gam * (np.exp(1j * th) / 2 + .5), [mu - gam]])
    nodes = gam * np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3]) + mu
    Xi = -10.0 ** np.linspace(-8, 8, 10000) + s2 ** 2
    v = np.random.default_rng(1).standard_normal(n) + 0j
    kw = dict(Xi=Xi, maxit=60, v=v, leja=0, nodes=nodes, reusefact=2)
    _DeviceRefactor.clear()
    l1 = na.nleigs(lowr, Sig, errmeasure=na.StandardSPMFErrmeasure(lowr), **kw)[0]
    _DeviceRefactor.wait()
    plans = [p for p in _DeviceRefactor.plans.values() if p["state"] == "ready"]
    assert len(plans) == 1
    u0 = plans[0]["uses"]
    l2 = na.nleigs(lowr, Sig, errmeasure=na.StandardSPMFErrmeasure(lowr), **kw)[0]
    assert plans[0]["uses"] - u0 == 5 and plans[0]["fails"] == 0          # the five shifts of variant R1
    assert len(l1) == len(l2) and len(l1) >= 5
    _match(l2, l1, 1e-8)


def test_nleigs_lowrank_degree2_vs_oracle(na):
    """polynomial part of degree 2 + two low-rank exponential terms: the n-row recurrences of blocks 1..p-1, the UU^H seams
    at block p and the corrected first-block-row term (oracle/nleigs.py backslash) on the device; dynamic and static
    variants against the oracle (1e-8) and against the full-block device run"""
    import warnings
    from oracle import neps as on, nleigs as onl
    try:
        from test_oracle_kat import _lowrank_p2_problem
    except ImportError:
        from tests.test_oracle_kat import _lowrank_p2_problem
    n, B, C, Sigma = _lowrank_p2_problem(None)
    fv = [na.funcs.Exp(-1.0), na.funcs.Exp(-0.5)]; ofv = [on.f_exp(-1.0), on.f_exp(-0.5)]
    full = na.SumNEP(na.PEP(B), na.SPMF_NEP(C, fv))
    lowr = na.SumNEP(na.PEP(B), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(C[i], fv[i]) for i in range(2)]))
    olr = on.SumNEP(on.PEP(B), on.LowRankFactorizedNEP([on.LowRankMatrixAndFunction(C[i], ofv[i]) for i in range(2)]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, _, _ = na.nleigs(full, Sigma, maxit=60, v=np.ones(n) + 0j)
        assert len(ref) == 8
        for static in (False, True):
            lam, X, res = na.nleigs(lowr, Sigma, maxit=60, v=np.ones(n) + 0j, static=static)
            lo, Xo, ro = onl.nleigs(olr, Sigma, maxit=60, v=np.ones(n) + 0j, static=static)
            _match(lam, lo, 1e-8)
            _match(lam, ref, 1e-7)
    with pytest.raises(ValueError):                     # p = 3: the reference reads a block that does not exist yet
        na.nleigs(na.SumNEP(na.PEP(B + [0.01 * np.eye(n)]), lowr.nep2), Sigma, maxit=10, v=np.ones(n) + 0j)


@pytest.mark.parametrize("variant", ["P", "R2", "S", "naive"])
def test_nleigs_gun_variants_lowrank_vs_oracle(na, variant):
    """the other gun variants of test/nleigs (nleigs_gun_variant_p.jl: polynomial, no poles; _r2.jl: Leja-Bagby nodes +
    cyclic shifts after convergence, minit=60; _s.jl: static, minit=70; nleigs_gun_naive.jl: defaults on a square) on the
    reduced low-rank gun problem, device against oracle: same eigenvalue count, eigenvalues to 1e-7 relative"""
    import warnings
    from oracle import neps as on, nleigs as onl, solvers as osol
    n = 1310
    K, M, W1, W2 = na.gallery.gun_matrices(n)
    s2 = na.gallery.GUN_SIGMA2
    fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -s2 ** 2)]
    ofv = [on.f_isqrt(0.0), on.f_isqrt(-s2 ** 2)]
    lowr = na.SumNEP(na.PEP([K, -M]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]), na.LowRankMatrixAndFunction(W2, fv[1])]))
    olr = on.SumNEP(on.PEP([K, -M]), on.LowRankFactorizedNEP([on.LowRankMatrixAndFunction(W1, ofv[0]), on.LowRankMatrixAndFunction(W2, ofv[1])]))
    gam = 300.0 ** 2 - 200.0 ** 2; mu = 250.0 ** 2
    th = np.linspace(0, np.pi, int(round(np.pi / 2 * 1000)) + 2)
    Sig = np.concatenate([(mu - gam) + 2 * gam * (np.exp(1j * th) / 2 + .5), [mu - gam]])
    nodes = gam * np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3]) + mu
    Xi = -10.0 ** np.linspace(-8, 8, 10000) + s2 ** 2
    v = np.random.default_rng(1).standard_normal(n) + 0j
    kw = {"P": dict(maxit=100, v=v, leja=0, nodes=nodes, reusefact=2),
          "R2": dict(Xi=Xi, minit=60, maxit=100, v=v, nodes=nodes),
          "S": dict(Xi=Xi, minit=70, maxit=100, v=v, nodes=nodes, static=True),
          "naive": dict(v=v, maxit=60)}[variant]
    if variant == "naive":
        Sig = 150.0 ** 2 + 200.0 * np.array([-1 - 1j, -1 + 1j, 1 + 1j, 1 - 1j])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lam, X, res = na.nleigs(lowr, Sig, errmeasure=na.StandardSPMFErrmeasure(lowr), **kw)
        lo, Xo, ro = onl.nleigs(olr, Sig, errmeasure=osol.StandardSPMFErrmeasure(olr), **kw)
    assert len(lam) >= 1 or variant == "naive"       # the stand-in has no eigenvalue in the naive square; counts must agree
    _match(lam, lo, 1e-7)


def test_nleigs_particle_lowrank_static(na):
    """test/nleigs/nleigs_particle_variant_s.jl on the device: 83 terms (the stacked CSR carries up to 128), low-rank
    blocks of r = 162 rows, static variant with the reference's settings and start vector -> the 2 eigenvalues the
    reference's verify_lambdas(2, ...) expects, equal to the oracle's to 1e-9, residuals below 1e-5"""
    import warnings
    from oracle import gallery as og, nleigs as onl, solvers as osol
    nep, Sigma, Xi, v, nodes, xmin, xmax = na.gallery.particle_init(2)
    onep = og.particle_init(2)[0]
    kw = dict(Xi=Xi, maxdgr=50, minit=120, maxit=200, v=v, nodes=nodes, static=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        info = {}
        lam, X, res = na.nleigs(nep, Sigma, info=info, **kw)
        lo, Xo, ro = onl.nleigs(onep, Sigma, **kw)
    assert len(lam) == 2 and info["lowrank_r"] == 162
    _match(lam, lo, 1e-9)
    E = osol.ResidualErrmeasure(onep)
    assert max(E(lam[i], X[:, i]) for i in range(2)) < 1e-5


def test_nleigs_particle_lowrank_dynamic(na):
    """test/nleigs/nleigs_particle_variant_r2.jl on the device (dynamic variant: leja points in the expansion phase,
    repeated nodes after the freeze) with the reference's settings and DEFAULT tolerance: with the seeded start vector
    of the oracle KAT the run ends with exactly the 2 eigenvalues verify_lambdas(2, ...) expects, residuals < 1e-10,
    equal to the oracle's; with the reference's own start vector device and oracle level off alike (tol = 1e-7; see the
    oracle KAT for why)"""
    import warnings
    from oracle import gallery as og, nleigs as onl, solvers as osol
    nep, Sigma, Xi, v, nodes, xmin, xmax = na.gallery.particle_init(2)
    onep = og.particle_init(2)[0]
    E = osol.ResidualErrmeasure(onep)
    v1 = np.random.default_rng(1).standard_normal(len(v)) + 0j
    for vv, tol in ((v1, 1e-10), (v, 1e-7)):
        kw = dict(Xi=Xi, maxdgr=50, minit=30, maxit=100, v=vv, nodes=nodes, tol=tol)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lam, X, res = na.nleigs(nep, Sigma, **kw)
            lo, Xo, ro = onl.nleigs(onep, Sigma, **kw)
        assert len(lam) == 2 and len(lo) == 2
        _match(lam, lo, 1e-8)
        assert max(E(lam[i], X[:, i]) for i in range(2)) < tol


def test_wep_linsolvers_small(na):
    """test/wep_small.jl:24-28 on the device: the Sylvester-SMW preconditioner with one grid point per region (N = nz)
    inverts SchurMatVec exactly; device SchurMatVec / assembled Schur complement / dense-transform Sylvester solve against
    the oracle (1e-13); the Schur-complement lin_solve == M(lam)^{-1} for the three inner solvers; error behaviour"""
    import torch
    from oracle import wep as ow, wep_linsolvers as owl
    nep = na.nep_gallery("WEP", nx=11, nz=7, benchmark_problem="TAUSCH")
    onep = ow.WEP_FD(11, 7, "TAUSCH")
    lam = -1.3 - 0.31j
    rng = np.random.default_rng(0)
    b1 = rng.random(77) + 1j * rng.random(77)
    wl = na.wep_linsolvers
    ops = wl.SchurOps(nep, lam)
    out = torch.empty(77, dtype=torch.complex128, device="cuda")
    Sb = ops.matvec(na.to_dev(b1)[0], out).cpu().numpy().copy()
    So = owl.SchurMatVec(onep, lam)(b1)
    assert np.linalg.norm(Sb - So) < 1e-13 * np.linalg.norm(So)
    assert np.linalg.norm(na.construct_WEP_schur_complement(nep, lam) @ b1 - So) < 1e-13 * np.linalg.norm(So)
    P = na.wep_generate_preconditioner(nep, 7, lam)
    r = na.to_dev(Sb)[0]
    b2 = P(r).cpu().numpy()
    assert np.linalg.norm(b1 - b2) / np.linalg.norm(b1) < 1e-13
    X = rng.random((7, 11)) + 1j * rng.random((7, 11))
    C = onep._A(lam) @ X + (onep.wd.Dxx.T @ X.T).T
    Cd = na.to_dev(C)
    assert np.linalg.norm(na.to_host(P.linv(Cd)) - X) < 1e-12 * np.linalg.norm(X)
    x = rng.random(nep.n) + 1j * rng.random(nep.n)
    M = onep.compute_Mder(lam)
    for st in ("backslash", "factorized", "gmres"):
        kw = (("Pl", P), ("reltol", 1e-12)) if st == "gmres" else ()
        solver = na.create_linsolver(na.WEPLinSolverCreator(solver_type=st, kwargs=kw), nep, lam)
        y = na.lin_solve(solver, x)
        assert np.linalg.norm(M @ y - x) < 1e-11 * np.linalg.norm(x), st
    with pytest.raises(ValueError):
        na.wep_generate_preconditioner(na.nep_gallery("WEP", nx=11, nz=9), 3, lam)
    with pytest.raises(ValueError):
        na.wep_generate_preconditioner(nep, 2, lam)
    with pytest.raises(ValueError):
        na.create_linsolver(na.WEPLinSolverCreator(solver_type="qr"), nep, lam)
    with pytest.raises(TypeError):
        na.create_linsolver(na.WEPLinSolverCreator(), na.nep_gallery("dep0"), lam)
    for drv in (na.nleigs, na.iar_chebyshev, na.ilan):               # drivers that bypass the NEP's own operator refuse the corner term
        with pytest.raises(NotImplementedError):
            drv(nep)


@pytest.mark.parametrize("solver_type", ["factorized", "backslash", "gmres"])
def test_wep_linsolvers_resinv(na, solver_type):
    """test/wep_small.jl:30-61 on the device: resinv on the 109 x 105 JARLEBRING waveguide with WEPLinSolverCreator()
    (factorized Schur complement), :backslash and :gmres (N = 21 preconditioner, reltol 1e-7) -> residual < 1e-10 at the
    reference eigenvalue"""
    from oracle import wep as ow
    nep = na.nep_gallery("WEP", nx=109, nz=105, benchmark_problem="JARLEBRING")
    onep = ow.WEP_FD(109, 105, "JARLEBRING")
    n = nep.n; lam0 = -3 - 3.5j; v0 = np.ones(n) / np.sqrt(n)
    lref = -2.743228671961724 - 3.1439375599649972j
    E = lambda l, v: abs(l - lref) / abs(lref)
    kw = (("Pl", na.wep_generate_preconditioner(nep, 21, lam0)), ("reltol", 1e-7)) if solver_type == "gmres" else ()
    cr = na.WEPLinSolverCreator(solver_type=solver_type, kwargs=kw)
    lam, v = na.resinv(nep, lam=lam0, v=v0, errmeasure=E, tol=1e-12, linsolvercreator=cr)
    assert np.linalg.norm(onep.compute_Mlincomb(lam, v)) / np.linalg.norm(v) < 1e-10 and abs(lam - lref) < 1e-10


def test_wep_nep_solvers_with_schur_linsolver(na):
    """test/wep_small.jl:63-77: quasinewton and iar on the 109 x 105 JARLEBRING waveguide with
    WEPLinSolverCreator(solver_type=:factorized) -- residual < 1e-10, resp. the reference eigenvalue to 1e-10; and tiar with
    the matrix-free GMRES solver + refinement sweeps finds it too"""
    from oracle import wep as ow
    nep = na.nep_gallery("WEP", nx=109, nz=105, benchmark_problem="JARLEBRING")
    onep = ow.WEP_FD(109, 105, "JARLEBRING")
    n = nep.n; lam0 = -3 - 3.5j; v0 = np.ones(n) / np.sqrt(n)
    lref = -2.743228671961724 - 3.1439375599649972j
    E = lambda l, v: abs(l - lref) / abs(lref)
    cr = na.WEPLinSolverCreator(solver_type="factorized")
    lam, v = na.quasinewton(nep, lam=lam0, v=v0, errmeasure=E, tol=1e-12, linsolvercreator=cr)
    assert np.linalg.norm(onep.compute_Mlincomb(lam, v)) / np.linalg.norm(v) < 1e-10
    lams, Q, _ = na.iar(nep, sigma=lam0, neigs=3, maxit=100, v=v0, tol=1e-8, linsolvercreator=cr)
    assert min(abs(lref - lams)) < 1e-10
    P = na.wep_generate_preconditioner(nep, 21, lam0)
    crg = na.WEPLinSolverCreator(solver_type="gmres", kwargs=(("Pl", P), ("reltol", 1e-6), ("restart", 60), ("maxiter", 300)),
                                 refinements=10)
    lam2, Q2, _, _ = na.tiar(nep, sigma=lam0, neigs=3, maxit=100, v=v0, tol=1e-8, linsolvercreator=crg)
    assert min(abs(lref - lam2)) < 1e-10


def test_sharded_beyn_through_the_c_abi_one_rank_rccl(na):
    """the multi-rank exchange of the C ABI (nep_comm_unique_id / nep_comm_create / nep_allgather_sum, csrc/comm.hip) with a
    real RCCL communicator of one rank: the all-gather + fixed-order sum is the identity, in place and out of place, and
    contour_beyn through MatrixTrapezoidalSharded (nodes i = r mod P) returns the eigenpairs of the unsharded integrator
    bit for bit"""
    import torch
    uid = na.DeviceComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = na.DeviceComm(0, 1, uid)
    try:
        import ctypes as C
        info = (C.c_int32 * 2)()
        assert na._lib.lib.nep_comm_info(comm.h, info) == 0 and list(info) == [0, 1]
        S = torch.randn(3, 7, 129, dtype=torch.float64, device="cuda").to(torch.complex128)
        ref = S.clone()
        out = torch.empty_like(S)
        comm.allgather_sum(S, out)
        comm.allgather_sum(S)                              # in place
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and torch.equal(S, ref)
        nep = na.nep_gallery("dep0")
        lam0, V0 = na.contour_beyn(nep, na.MatrixTrapezoidal, sigma=0.2, radius=1.0, neigs=4, N=200, sanity_check=False)
        na.MatrixTrapezoidalSharded.comm = comm
        info = {}
        lam1, V1 = na.contour_beyn(nep, na.MatrixTrapezoidalSharded, sigma=0.2, radius=1.0, neigs=4, N=200, sanity_check=False, info=info)
        assert info["world"] == 1 and info["nodes"] == 200
        assert np.array_equal(lam0, lam1) and np.array_equal(V0, V1)
        # argument errors never reach RCCL
        assert na._lib.lib.nep_allgather_sum(comm.h, None, 5, None, None) == -2
        assert na._lib.lib.nep_comm_create(2, 2, None, None) == -2
    finally:
        na.MatrixTrapezoidalSharded.comm = None
        comm.close()


@pytest.mark.parametrize("nx,nz", [(19, 15), (115, 111), (15, 11), (29, 25)])
def test_wep_schur_matvec_stencil_equals_assembled(na, nx, nz):
    """SchurMatVec (Waveguide.jl:394-425) in its matrix-free form (nep_wep_schur_matvec: five-point stencil + the boundary functional
    gathered inside the P^{-1} kernel) against the assembled route (K1 on the three stacked sparse terms, P^{-1}, C1) and against the
    host Schur complement of construct_WEP_schur_complement (Waveguide.jl:523-550); nz = 15, 111 take the symmetric-half P^{-1}
    kernel, nz = 11 (prime) and 25 (no coprime factors) the plain one"""
    import torch
    from nep_amd import wep_linsolvers as wl
    nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING")
    lam = -1.3 - 0.7j
    ops = wl.SchurOps(nep, lam)
    assert ops.stencil is not None
    rng = np.random.default_rng(nx)
    v = rng.standard_normal(nx * nz) + 1j * rng.standard_normal(nx * nz)
    vd = na.to_dev(v)[0]
    o1 = torch.empty_like(vd); o2 = torch.empty_like(vd)
    ops.matvec(vd, o1)
    st, ops.stencil = ops.stencil, None
    ops.matvec(vd, o2)
    ops.stencil = st
    torch.cuda.synchronize()
    a = na.to_host(o1).ravel(); b = na.to_host(o2).ravel()
    ref = wl.construct_WEP_schur_complement(nep, lam) @ v
    assert np.linalg.norm(a - b) <= 1e-13 * np.linalg.norm(b)
    assert np.linalg.norm(a - ref) <= 1e-12 * np.linalg.norm(ref)


@pytest.mark.parametrize("nz,N", [(15, 15), (15, 5), (105, 21), (111, 37), (999, 37)])
def test_wep_preconditioner_three_transform_form(na, nz, N):
    """solve_smw (waveguide_preconditioner.jl:323-421) through nep_wep_smw_apply -- three transforms, region means taken in mode
    space, the expansion formed inside the transform's loader, the second solve subtracted in mode space -- against the piecewise
    route (four transforms; NEP_WEP_SMW_FUSED=0) on the same vector, and, with one region per grid row (N = nz), against the
    property that the preconditioner then inverts SchurMatVec exactly"""
    import torch
    from nep_amd import wep_linsolvers as wl
    nx = nz + 4
    nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING")
    lam = -1.3 - 0.7j
    P = na.wep_generate_preconditioner(nep, N, lam)
    # the SMW matrix itself: mode-space columns (nep_wep_smw_matrix_modes, interior regions batched) against the piecewise columns
    os.environ["NEP_WEP_SMW_FUSED"] = "0"
    try:
        P0 = na.wep_generate_preconditioner(nep, N, lam)
    finally:
        del os.environ["NEP_WEP_SMW_FUSED"]
    assert P0._fused is False and P._G is not None
    assert np.linalg.norm(P._M - P0._M) <= 1e-12 * np.linalg.norm(P0._M), np.linalg.norm(P._M - P0._M) / np.linalg.norm(P0._M)
    rng = np.random.default_rng(nz + N)
    v = rng.standard_normal(nx * nz) + 1j * rng.standard_normal(nx * nz)
    r1 = na.to_dev(v)[0].clone(); r2 = na.to_dev(v)[0].clone()
    P(r1)
    assert P._fused is True, "the three-transform form was not taken"
    P._fused = False
    P(r2)
    P._fused = None
    torch.cuda.synchronize()
    a = na.to_host(r1.reshape(1, -1))[:, 0]; b = na.to_host(r2.reshape(1, -1))[:, 0]
    assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
    r3 = na.to_dev(v)[0].clone()
    P(r3); torch.cuda.synchronize()
    assert np.array_equal(na.to_host(r3.reshape(1, -1))[:, 0], a)            # same bits on a repeat
    if N == nz:
        ops = wl.SchurOps(nep, lam)
        out = torch.empty_like(r1)
        ops.matvec(na.to_dev(v)[0], out)
        w = P(out); torch.cuda.synchronize()
        assert np.linalg.norm(na.to_host(w.reshape(1, -1))[:, 0] - v) <= 1e-11 * np.linalg.norm(v)


@pytest.mark.parametrize("nz,nx", [(7, 11), (105, 109), (299, 303), (60, 64), (111, 115), (999, 1003)])
def test_wep_sylvester_solve_pfa_vs_numpy(na, nz, nx):
    """nep_wep_sylv_solve (prime-factor DFT along z + per-mode tridiagonal scans along x, csrc/wep.hip) against the dense
    diagonalisation it replaces, X = F (G .* (F^H C W)) W with the DFT matrix F and the sine matrix W
    (waveguide_preconditioner.jl:120-219): sizes with N2 = 1 (7 prime), 105 = 15 * 7, 299 = 13 * 23, 60 = 15 * 4 (an even factor: the
    plain dense stages), 111 = 37 * 3 and the full-size 999 = 37 * 27 (odd factors: the symmetric-half stages); also the
    region means / expansion kernels against their indicator-matrix products"""
    import ctypes as C
    import torch
    L_ = na._lib.lib
    rng = np.random.default_rng(nz)
    hx, hz, sigma, kbar = 0.37, 0.21, -3 - 3.5j, 2.1 + 0.3j
    v = np.zeros(nz, dtype=complex); v[0] = -2; v[1 % nz] += 1; v[nz - 1] += 1; v /= hz ** 2
    w = np.zeros(nz, dtype=complex); w[1 % nz] += 1; w[nz - 1] += -1; w *= sigma / hz
    D = np.fft.fft(v + w) + (sigma ** 2 + kbar)
    S = -(4.0 / hx ** 2) * np.sin(np.pi * np.arange(1, nx + 1) / (2 * (nx + 1))) ** 2
    F = np.fft.fft(np.eye(nz), axis=0) / np.sqrt(nz)
    jx = np.arange(1, nx + 1)
    W = np.sqrt(2.0 / (nx + 1)) * np.sin(np.pi * np.outer(jx, jx) / (nx + 1))
    Cm = rng.standard_normal((nz, nx)) + 1j * rng.standard_normal((nz, nx))
    ref = F @ ((F.conj().T @ Cm @ W) / (D[:, None] + S[None, :])) @ W
    h = C.c_void_p()
    Dc = np.ascontiguousarray(D)
    assert L_.nep_wep_sylv_create(nz, nx, na._lib.hptr(Dc), 1.0 / hx ** 2, C.byref(h)) == 0
    info = (C.c_int32 * 4)()
    assert L_.nep_wep_sylv_info(h, info) == 0 and info[0] * info[1] == nz and np.gcd(info[0], info[1]) == 1
    Xd = na.to_dev(Cm)                                   # (nx, nz) tensor = column-major nz x nx
    for _ in range(2):                                   # the work block is reused: repeatable
        Xd = na.to_dev(Cm)
        assert L_.nep_wep_sylv_solve(h, C.c_void_p(Xd.data_ptr()), None) == 0
    X = na.to_host(Xd)
    assert np.linalg.norm(X - ref) <= 1e-12 * np.linalg.norm(ref)
    assert L_.nep_wep_sylv_destroy(h) == 0
    if nx == nz + 4:
        for N in [d for d in (1, 3, 5, 7, 13) if nz % d == 0][:3]:
            Lr = nz // N
            Bz = np.kron(np.eye(N), np.ones((Lr, 1)))
            Bx = np.zeros((nx, N + 4)); Bx[0, 0] = Bx[1, 1] = Bx[nx - 2, N + 2] = Bx[nx - 1, N + 3] = 1.0
            for j in range(N):
                Bx[2 + j * Lr:2 + (j + 1) * Lr, 2 + j] = 1.0
            wx = np.ones(N + 4); wx[2:N + 2] = 1.0 / Lr
            means_ref = (Bz.T / Lr) @ Cm @ (Bx * wx[None, :])
            out = torch.empty((N + 4, N), dtype=torch.complex128, device="cuda")
            assert L_.nep_wep_region_means(nz, nx, N, C.c_void_p(na.to_dev(Cm).data_ptr()), C.c_void_p(out.data_ptr()), None) == 0
            assert np.linalg.norm(na.to_host(out) - means_ref) <= 1e-13 * np.linalg.norm(means_ref)
            al = rng.standard_normal((N, N + 4)) + 1j * rng.standard_normal((N, N + 4))
            K = rng.standard_normal((nz, nx)) + 1j * rng.standard_normal((nz, nx))
            Y = torch.empty((nx, nz), dtype=torch.complex128, device="cuda"); eb = torch.empty((2, nz), dtype=torch.complex128, device="cuda")
            assert L_.nep_wep_region_expand(nz, nx, N, C.c_void_p(na.to_dev(al).data_ptr()), C.c_void_p(na.to_dev(K).data_ptr()), 0.7, -0.3,
                                            C.c_void_p(Y.data_ptr()), C.c_void_p(eb.data_ptr()), None) == 0
            tz = Bz @ al
            assert np.linalg.norm(na.to_host(Y) - (tz @ Bx.T) * K) <= 1e-13 * np.linalg.norm(K)
            cb = np.zeros((N + 4, 2)); cb[0, 0] = 0.7; cb[1, 0] = -0.3; cb[N + 2, 1] = -0.3; cb[N + 3, 1] = 0.7
            assert np.linalg.norm(na.to_host(eb) - tz @ cb) <= 1e-13 * np.linalg.norm(tz)


@pytest.mark.parametrize("nz", [7, 105, 299])
def test_wep_pinv_dft_vs_dense(na, nz):
    """nep_wep_pinv_apply (two prime-factor DFTs per half in one launch) against blkdiag(R, R) diag(sinv) blkdiag(R, R)^H x with
    the dense R = reverse(bb .* fft(.)) of Waveguide.jl:53-65"""
    import ctypes as C
    L_ = na._lib.lib
    rng = np.random.default_rng(nz + 1)
    p = 0.37
    bb = np.exp(-2j * np.pi * np.arange(nz) * (-p) / nz)
    R = (bb[:, None] * np.fft.fft(np.eye(nz), axis=0))[::-1, :]
    sinv = rng.standard_normal(2 * nz) + 1j * rng.standard_normal(2 * nz)
    x = rng.standard_normal(2 * nz) + 1j * rng.standard_normal(2 * nz)
    ref = np.concatenate([R @ (sinv[:nz] * (R.conj().T @ x[:nz])), R @ (sinv[nz:] * (R.conj().T @ x[nz:]))])
    h = C.c_void_p()
    assert L_.nep_wep_pinv_create(nz, na._lib.hptr(np.ascontiguousarray(bb)), C.byref(h)) == 0
    xd = na.to_dev(x)[0]; sd = na.to_dev(sinv)[0]; od = na.to_dev(np.zeros(2 * nz, dtype=complex))[0]
    assert L_.nep_wep_pinv_apply(h, C.c_void_p(sd.data_ptr()), C.c_void_p(xd.data_ptr()), C.c_void_p(od.data_ptr()), None) == 0
    out = na.to_host(od.reshape(1, -1))[:, 0]
    assert np.linalg.norm(out - ref) <= 1e-12 * np.linalg.norm(ref)
    assert L_.nep_wep_pinv_apply(h, C.c_void_p(sd.data_ptr()), C.c_void_p(xd.data_ptr()), C.c_void_p(xd.data_ptr()), None) == 0   # in place
    assert np.linalg.norm(na.to_host(xd.reshape(1, -1))[:, 0] - ref) <= 1e-12 * np.linalg.norm(ref)
    assert L_.nep_wep_pinv_destroy(h) == 0


def test_nleigs_custom_nep_type(na):
    """test/nleigs/nleigs_nep_types.jl "Custom NEP type": a NEP given only by lam -> M(lam) (na.Mder_NEP) through nleigs with
    matrix-valued divided differences (the D_j become the terms of an SPMF in the rational Newton basis on the device): the 4
    eigenvalues of the underlying PEP, equal to the oracle's run and to the PEP run; plus a sparse n = 655 case against the
    SPMF formulation of the same problem"""
    from oracle import neps as on, nleigs as onl
    B = [np.array([[1.0, 3], [5, 6]]), np.array([[3.0, 4], [6, 6]])]
    opep = on.PEP(B + [np.eye(2)])
    Sigma = np.array([-10.0 - 2j, 10 - 2j, 10 + 2j, -10 + 2j])
    custom = na.Mder_NEP(2, lambda lam: opep.compute_Mder(lam))
    lam, X, res = na.nleigs(custom, Sigma, maxit=10, v=np.ones(2) + 0j, blksize=5)
    lo, Xo, ro = onl.nleigs(on.Mder_NEP(2, lambda lam: opep.compute_Mder(lam)), Sigma, maxit=10, v=np.ones(2) + 0j, blksize=5)
    assert len(lam) == len(lo) == 4
    assert max(np.min(abs(lo - l)) for l in lam) < 1e-9
    for i in range(4):                                           # verify_lambdas tolerance of the reference test
        assert np.linalg.norm(opep.compute_Mlincomb(lam[i], X[:, i])) / np.linalg.norm(X[:, i]) < 1e-5
    # compute_Mlincomb of the function-handle NEP runs on the device
    v = np.array([1.0 + 2j, -0.5j])
    assert np.linalg.norm(custom.compute_Mlincomb(0.3 + 0.1j, v) - opep.compute_Mder(0.3 + 0.1j) @ v) < 1e-13
    # sparse problem (n = 300 quadratic PEP) as a function handle against its PEP form
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    n = 300
    A0 = sp.csr_matrix(0.2 * sp.random(n, n, 0.02, random_state=rng) + sp.diags(np.linspace(-40.0, 40.0, n)))
    A1 = sp.csr_matrix(0.3 * sp.random(n, n, 0.02, random_state=rng))
    A2 = sp.csr_matrix(-sp.identity(n))
    pep = na.PEP([A0, A1, A2])
    fun = na.Mder_NEP(n, lambda lam: A0 + lam * A1 + lam ** 2 * A2)
    Sig = np.array([-0.6 - 0.6j, 0.6 - 0.6j, 0.6 + 0.6j, -0.6 + 0.6j])
    v0 = np.random.Generator(np.random.Philox(3)).standard_normal(n) + 0j
    kw = dict(maxit=60, v=v0, tol=1e-9)
    l1, _, _ = na.nleigs(pep, Sig, **kw)
    l2, X2, _ = na.nleigs(fun, Sig, **kw)
    assert len(l1) == len(l2) >= 2
    assert max(np.min(abs(l1 - l)) for l in l2) < 1e-8
    for i in range(len(l2)):
        M = A0 + l2[i] * A1 + l2[i] ** 2 * A2
        assert np.linalg.norm(M @ X2[:, i]) / np.linalg.norm(X2[:, i]) < 1e-8


def test_lu_batch_from_terms_equals_batch_from_values(na):
    """nep_lu_factor_dev_batch_terms (values of M(lam_b) = sum_t f_t(lam_b) A_t formed inside the scatter kernel from the
    device-resident term block) against nep_lu_factor_dev_batch on host-assembled values: same solves to round-off, and
    both against the host factorisation"""
    from nep_amd.linsolvers import _DeviceRefactor, DeviceLU
    if not _DeviceRefactor.enabled():
        pytest.skip("device numeric factorisation switched off")
    import torch
    n = 1310
    nep = na.nep_gallery("gun_spmf", n)
    lams = [250.0 ** 2 + 1.2e4 * np.exp(2j * np.pi * (j + 0.5) / 6) for j in range(6)]
    _DeviceRefactor.clear()
    A0 = sp.csc_matrix(nep.compute_Mder(lams[0]), dtype=np.complex128)
    DeviceLU(A0)                                           # host factorisation, seeds the plan
    _DeviceRefactor.wait()
    indptr, indices, D_dev, G = nep.aligned_terms_dev()
    plan = _DeviceRefactor.lookup(_DeviceRefactor.key(A0, (None, None, None)))
    assert plan is not None
    _, _, vals = nep.compute_Mder_batch(lams)
    Cf = np.array([[f.derivs(l, 1)[0] for f in nep.get_fv()] for l in lams], dtype=np.complex128)
    normA = np.sqrt(np.einsum("bs,st,bt->b", Cf.conj(), G, Cf).real)
    assert np.allclose(normA, np.linalg.norm(vals, axis=1), rtol=1e-12)
    la = _DeviceRefactor.factor_batch(plan, n, vals)
    lb = _DeviceRefactor.factor_batch_terms(plan, n, D_dev, Cf, normA)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((4, n)) + 1j * rng.standard_normal((4, n))
    bd = torch.from_numpy(b).to("cuda")
    for j, lam in enumerate(lams):
        assert la[j] is not None and lb[j] is not None
        xa = la[j].solve(bd).cpu().numpy(); xb = lb[j].solve(bd).cpu().numpy()
        M = nep.compute_Mder(lam)
        assert np.abs(xa - xb).max() <= 1e-10 * np.abs(xa).max()
        assert np.linalg.norm(M @ xb.T - b.T) <= 1e-8 * np.linalg.norm(b)


@pytest.mark.skipif(not os.environ.get("NEPMI_GUN_DIR"), reason="needs the physical gun_K.txt / gun_M.txt (absent from the "
                    "reference checkout, .MISSING_LARGE_BLOBS): set NEPMI_GUN_DIR")
def test_physical_gun_known_answers(na):
    """the known answers the reference holds for the PHYSICAL gun problem, runnable as soon as the blobs are supplied:
    ||K||_1, ||M||_1 (test/rk_helper/gun_test_utils.jl:50-51, enforced by the loader), the eigenvalue
    22345.116783765+0.644998598i found by quasinewton from 150^2+i (test/gun_native.jl:9-19), the derivative check of
    test/gun_native.jl:22-33, and the 21 eigenvalues of nleigs variant R1 (test/nleigs/nleigs_gun_variant_r1.jl:17)"""
    from nep_amd import gallery
    K, M, W1, W2 = gallery.gun_matrices()
    assert abs(gallery._onenorm(K) - 1.474544889815002e+05) <= 1e-12 * 1.474544889815002e+05
    assert abs(gallery._onenorm(M) - 2.726114618171165e-02) <= 1e-12 * 2.726114618171165e-02
    nep = na.nep_gallery("nlevp_native_gun"); n = nep.n
    tol = 1e-11
    lam, v = na.quasinewton(nep, lam=150.0 ** 2 + 1j, v=np.ones(n), tol=tol, maxit=500)
    v = v / np.linalg.norm(v)
    assert np.linalg.norm(na.to_host(nep.compute_Mlincomb(lam, v))) < np.sqrt(tol)
    assert abs(lam - (22345.116783765 + 0.644998598j)) < np.sqrt(tol) * 100
    l0 = 150.0 ** 2 + 2j; ee = 1e-4
    vv = np.random.default_rng(0).standard_normal(n)
    z1 = na.to_host(nep.compute_Mlincomb(l0, vv.reshape(-1, 1), a=np.array([1.0]), startder=1))
    z2 = (na.to_host(nep.compute_Mlincomb(l0 + ee, vv)) - na.to_host(nep.compute_Mlincomb(l0 - ee, vv))) / (2 * ee)
    assert np.linalg.norm(z2.ravel() - z1.ravel()) < ee ** 2 * 1000
    Sigma, Xi, nodes, v0 = _gun_r1_setup(n)
    lam_r1, X, res = na.nleigs(nep, Sigma, Xi=Xi, maxit=100, v=v0, leja=0, nodes=nodes, reusefact=2,
                               errmeasure=na.StandardSPMFErrmeasure(nep))[:3]
    assert len(lam_r1) == 21


@pytest.mark.parametrize("nx,nz", [(109, 105), (303, 299)])
def test_wep_residual_batch_split_vs_oracle(na, nx, nz, monkeypatch):
    """ResidualErrmeasure on the waveguide problem for a batch of GENERIC pairs (residuals of order one, k below and above
    the panel widths of the tiled K2 kernel): the one-pass form (nep_resid_split_dev: norms over the interior rows + the
    2 nz boundary rows of the residual, corner term added there) equals the full-block form (NEP_WEP_RESID_SPLIT=0) and the
    oracle's matrix-free operator to 1e-12"""
    import torch
    from oracle import wep as ow
    nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); n = nep.n
    o = ow.WEP_FD(nx, nz, "JARLEBRING")
    rng = np.random.default_rng(nx)
    for k in (1, 5, 13):
        Q = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
        lams = (-2.7 - 3.1j) + 0.3 * (rng.standard_normal(k) + 1j * rng.standard_normal(k))
        ref = np.array([np.linalg.norm(o._mlincomb(complex(lams[s]), Q[:, s:s + 1], np.ones(1, dtype=complex))) / np.linalg.norm(Q[:, s])
                        for s in range(k)])
        QT = torch.from_numpy(np.ascontiguousarray(Q)).to("cuda")
        E = na.ResidualErrmeasure(nep)
        got = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("NEP_WEP_RESID_SPLIT", flag)
            got[flag] = np.asarray(E.batch(list(lams), QT))
            assert np.allclose(got[flag], ref, rtol=1e-12), (flag, k)
        assert np.allclose(got["1"], got["0"], rtol=1e-13)


def test_tiar_column_major_ritz_blocks(na, monkeypatch):
    """tiar on a waveguide problem large enough for the column-major Ritz path (n >= 32768: dense.ColMajorBlock -> tiled K2 with
    contiguous column loads, WEP corner term on the column-major tail) returns the eigenpairs of the row-major path
    (NEP_K2_CM=0) and of the oracle's matrix-free residual; also through a user-supplied (callable) error measure"""
    from oracle import wep as ow
    nx, nz = 183, 179
    nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); n = nep.n
    assert n >= 32768
    v0 = np.ones(n) / np.sqrt(n)
    res = {}
    monkeypatch.setenv("NEP_K2_CM", "1")
    assert nep.prefers_colmajor_ritz()
    for flag in ("1", "0"):
        monkeypatch.setenv("NEP_K2_CM", flag)
        assert nep.prefers_colmajor_ritz() == (flag == "1")
        h = []
        lam, Q, _, _ = na.tiar(nep, sigma=-3 - 3.5j, gamma=1.0, maxit=40, neigs=np.inf, v=v0, tol=1e-8, errhist=h)
        res[flag] = (lam, Q, h)
    (l1, Q1, h1), (l0, Q0, h0) = res["1"], res["0"]
    assert len(l1) == len(l0) >= 2 and len(h1) == len(h0)
    assert max(np.min(abs(l0 - x)) for x in l1) < 1e-9
    for a, b in zip(h1, h0):
        kk = min(len(a), len(b), 4)
        for x, y in zip(np.sort(a)[:kk], np.sort(b)[:kk]):
            if x > 1e-10 and y > 1e-10:
                assert 0.5 < x / y < 2.0
    o = ow.WEP_FD(nx, nz, "JARLEBRING")
    for i in range(len(l1)):
        r = o._mlincomb(complex(l1[i]), Q1[:, i:i + 1], np.ones(1, dtype=complex))
        assert np.linalg.norm(r) / np.linalg.norm(Q1[:, i]) < 1e-8
    monkeypatch.setenv("NEP_K2_CM", "1")
    R = na.ResidualErrmeasure(nep)
    lam_c, Qc, _, _ = na.tiar(nep, sigma=-3 - 3.5j, gamma=1.0, maxit=25, neigs=np.inf, v=v0, tol=1e-8,
                              errmeasure=lambda l, v: float(na.estimate_error(R, l, v)))
    assert len(lam_c) >= 1 and max(np.min(abs(l1 - x)) for x in lam_c) < 1e-8


def test_iar_reruns_when_a_dgks_pass_is_missing(na):
    """the asynchronous DGKS enqueues two passes ("twice is enough"); the reference's DGKS repeats without a bound.  When the
    device reports that the criterion still held after the last enqueued pass, iar re-runs the call with the synchronous loop
    (exact DGKS).  Forced here by enqueuing ONE pass only (NEP_ORTH_DEV_PASSES=1, read once per process -> subprocess): the
    call must notice (orth_pass_misses = 1) and return the eigenvalues of the normal run"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); import nep_amd as na; "
            "nep = na.nep_gallery('gun_spmf_scaled', 1310); "
            "lam, Q, _ = na.iar(nep, sigma=0.0, gamma=1.0, maxit=40, neigs=np.inf, v=np.ones(nep.n), tol=1e-10); "
            "print(json.dumps({'misses': na.iar.orth_pass_misses, 're': list(np.sort_complex(lam).real), 'im': list(np.sort_complex(lam).imag)}))") % root
    out = {}
    for passes in ("1", "2"):
        env = dict(os.environ, NEP_ORTH_DEV_PASSES=passes)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        out[passes] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["2"]["misses"] == 0 and out["1"]["misses"] == 1
    a = np.array(out["1"]["re"]) + 1j * np.array(out["1"]["im"]); b = np.array(out["2"]["re"]) + 1j * np.array(out["2"]["im"])
    assert len(a) == len(b) >= 1 and np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())


def test_generic_fallbacks_from_MM_and_from_Mder(na):
    """src/NEPCore.jl:218-270: a user NEP type that implements only compute_MM (resp. only compute_Mder) gets compute_Mlincomb
    through compute_Mlincomb_from_MM (resp. _from_Mder) and compute_Mder through compute_Mder_from_MM; checked as test/core.jl:
    16-32 checks them -- against the type's own compute_Mlincomb / compute_Mder -- on dep0 (dense, n = 5) and a sparse SPMF with
    the gun functions, with zeros in `a` and with `startder`, for host and device V"""
    rng = np.random.default_rng(5)

    class OnlyMM(na.NEP):                      # knows nothing but compute_MM (delegated to a full type of this backend)
        def __init__(self, full):
            self.full, self.n = full, full.n

        def compute_MM(self, S, V):
            return self.full.compute_MM(S, V)
        compute_Mlincomb = na.NEP.compute_Mlincomb_from_MM
        compute_Mder = na.NEP.compute_Mder_from_MM

    class OnlyMder(na.NEP):
        def __init__(self, full):
            self.full, self.n = full, full.n

        def compute_Mder(self, lam, i=0):
            return self.full.compute_Mder(lam, i)
        compute_Mlincomb = na.NEP.compute_Mlincomb_from_Mder

    dep = na.nep_gallery("dep0")
    gun = na.nep_gallery("nlevp_native_gun", 300)
    for full, lam in ((dep, -0.3 + 0.2j), (gun, 250.0 ** 2 + 40.0j)):
        n = full.n
        um, ud = OnlyMM(full), OnlyMder(full)
        V = rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4))
        for a in (None, np.array([1.0, -2.0, 0.5, 3.0]), np.array([0.0, 0.0, 1.0, 2.0])):
            ref = full.compute_Mlincomb(lam, V, a) if a is not None else full.compute_Mlincomb(lam, V)
            z1 = um.compute_Mlincomb(lam, V, a)
            z2 = ud.compute_Mlincomb(lam, V, a)
            sc = np.linalg.norm(ref)
            assert np.linalg.norm(z1 - ref) <= 1e-9 * sc and np.linalg.norm(z2 - ref) <= 1e-10 * sc
        # startder: [0, 0, 1] == startder = 2 (test/core.jl:60-70)
        z3 = um.compute_Mlincomb(lam, V[:, :1], np.ones(1), 2)
        ref3 = full.compute_Mlincomb(lam, V[:, :1], np.ones(1), 2)
        assert np.linalg.norm(z3 - ref3) <= 1e-9 * max(np.linalg.norm(ref3), 1e-300)
        # device V in, device vector out
        zd = um.compute_Mlincomb(lam, na.to_dev(V))
        refd = full.compute_Mlincomb(lam, V)
        assert np.linalg.norm(na.to_host(zd.reshape(1, -1))[:, 0] - refd) <= 1e-9 * np.linalg.norm(refd)
    # compute_Mder_from_MM (dense, small: S has size n (i + 1))
    um = OnlyMM(dep)
    for i in (0, 1, 2):
        D = um.compute_Mder(-0.3 + 0.2j, i); R = np.asarray(dep.compute_Mder(-0.3 + 0.2j, i))
        assert np.linalg.norm(D - R) <= 1e-10 * max(np.linalg.norm(R), 1e-300)
