"""The BASELINE.json configurations C2-C5 as functions: device run, CPU-oracle twin, and the parity
rule of section 8d (same count, eigenvalues as multisets to 1e-8 relative, every pair below the driver's tolerance when
re-evaluated on the host).  Measurement / test infrastructure: used by bench.py, scripts/run_configs.py and
tests/test_gpu_fullsize.py -- never by the product."""
import os
import time

import numpy as np

GUN_N = 9956


def match(l1, l2, rtol):
    """eigenvalues as multisets: every entry of l1 has a partner in l2 within rtol * max(1, |lambda|)"""
    l2 = list(l2); worst = 0.0
    if len(l1) != len(l2):
        return False, None
    for x in l1:
        j = int(np.argmin([abs(x - y) for y in l2])); worst = max(worst, abs(x - l2[j]) / max(1.0, abs(x))); l2.pop(j)
    return worst <= rtol, worst


def host_backward_errors(Av, fv, lam, Q):
    """StandardSPMFErrmeasure (src/errmeasure.jl:186-190) re-evaluated in FP64 on the host from the returned pairs"""
    fro = [np.sqrt(abs(A.multiply(A.conj()).sum())) if hasattr(A, "multiply") else np.linalg.norm(A) for A in Av]
    out = []
    for s in range(len(lam)):
        fl = [f(lam[s]) for f in fv]
        r = sum(c * (A @ Q[:, s]) for A, c in zip(Av, fl))
        den = sum(n_ * abs(c) for n_, c in zip(fro, fl)) * np.linalg.norm(Q[:, s])
        out.append(float(np.linalg.norm(r) / den))
    return out


# ---- C2: gun SPMF, shift-and-scaled, iar m = 100 ------------------------------------------------------------------------
def c2_device(na, nep, maxit=100, permc=None, timers=None, hist=None, return_device=True):
    creator = na.FactorizeLinSolverCreator(permc_spec=permc, max_factorizations=0)
    lam, Q, V = na.iar(nep, sigma=0.0, gamma=1.0, maxit=maxit, neigs=np.inf, v=np.ones(nep.n), tol=1e-10,
                       linsolvercreator=creator, timers=timers, errhist=hist, return_device=return_device)
    return lam, Q


def c2_oracle(n=GUN_N, maxit=100, permc="MMD_AT_PLUS_A", timers=None, hist=None):
    from oracle import gallery as og, solvers as osol, neps as oneps
    onep = og.gun_spmf_scaled(n)
    der = oneps.DerSPMF(onep, 0.0, maxit)
    lam, Q, _ = osol.iar(der, sigma=0.0, gamma=1.0, maxit=maxit, neigs=np.inf, v=np.ones(n), tol=1e-10,
                         errmeasure=osol.StandardSPMFErrmeasure(onep),
                         linsolvercreator=osol.FactorizeLinSolverCreator(permc_spec=permc), timers=timers, errhist=hist)
    return lam, Q


def history_agreement(h_dev, h_ora, floor=1e-12, lead=8):
    """SURVEY.md section 8d parity rule (iv): per iteration the `lead` smallest error estimates of the two runs agree within a
    factor 10 wherever both are above `floor`.  Returns (ok, worst ratio >= 1, number of compared entries, iterations)."""
    worst = 1.0; cnt = 0
    for eg, eo in zip(h_dev, h_ora):
        a = np.sort(np.asarray(eg, dtype=float)); b = np.sort(np.asarray(eo, dtype=float))
        kk = min(len(a), len(b), lead)
        for x, y in zip(a[:kk], b[:kk]):
            if x > floor and y > floor:
                worst = max(worst, x / y, y / x); cnt += 1
    return bool(worst < 10.0 and len(h_dev) == len(h_ora)), float(worst), int(cnt), int(min(len(h_dev), len(h_ora)))


def c2_parity(lam_dev, hist_dev, lam_ora, hist_ora):
    """parity object of the headline configuration (SURVEY.md section 8d rules i, iii, iv): same count, eigenvalue multiset
    to 1e-8 relative, error histories within a factor 10 above 1e-12"""
    ok, worst = match(lam_dev, lam_ora, 1e-8)
    hok, hworst, hcnt, hit = history_agreement(hist_dev, hist_ora)
    return {"oracle_eigenpairs": int(len(lam_ora)), "same_count": bool(len(lam_dev) == len(lam_ora)),
            "eigenvalues_match_1e-8": bool(ok), "max_rel_eig_diff": worst, "history_within_x10_above_1e-12": hok,
            "history_worst_ratio": hworst, "history_entries_compared": hcnt, "history_iterations": hit}


# ---- C3: gun nleigs, variant R1 (test/nleigs/nleigs_gun_variant_r1.jl, test/rk_helper/gun_test_utils.jl) ---------------
def gun_r1_sets():
    gam = 300.0 ** 2 - 200.0 ** 2; mu = 250.0 ** 2; sigma2 = 108.8774
    xmin = gam * (-1) + mu; xmax = gam + mu
    th = np.linspace(0, np.pi, int(round(np.pi / 2 * 1000)) + 2)
    Sigma = np.concatenate([xmin + (xmax - xmin) * (np.exp(1j * th) / 2 + .5), [xmin]])
    nodes = gam * np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3]) + mu
    Xi = -10.0 ** np.linspace(-8, 8, 10000) + sigma2 ** 2
    return Sigma, Xi, nodes


def c3_kwargs(n, maxit=100):
    Sigma, Xi, nodes = gun_r1_sets()
    v = np.random.Generator(np.random.Philox(1)).standard_normal(n) + 0j
    return Sigma, dict(Xi=Xi, maxit=maxit, v=v, leja=0, nodes=nodes, reusefact=2, tol=1e-10)


def c3_device_nep(na, n=GUN_N):
    """gun_nep() of test/rk_helper/gun_test_utils.jl:37-43: PEP + LowRankFactorizedNEP (ranks 19 + 65)"""
    Kg, Mg, W1, W2 = na.gallery.gun_matrices(n)
    fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -na.gallery.GUN_SIGMA2 ** 2)]
    nep = na.SumNEP(na.PEP([Kg, -Mg]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]),
                                                                  na.LowRankMatrixAndFunction(W2, fv[1])]))
    nep.dev
    return nep


def c3_device(na, nep, maxit=100, info=None):
    Sigma, kw = c3_kwargs(nep.n, maxit)
    return na.nleigs(nep, Sigma, errmeasure=na.StandardSPMFErrmeasure(nep), info=info, **kw)


def c3_oracle(na, n=GUN_N, maxit=100):
    from oracle import neps as on, nleigs as onl, solvers as osol
    Kg, Mg, W1, W2 = na.gallery.gun_matrices(n)
    ofv = [on.f_isqrt(0.0), on.f_isqrt(-na.gallery.GUN_SIGMA2 ** 2)]
    olr = on.SumNEP(on.PEP([Kg, -Mg]), on.LowRankFactorizedNEP([on.LowRankMatrixAndFunction(W1, ofv[0]),
                                                                 on.LowRankMatrixAndFunction(W2, ofv[1])]))
    Sigma, kw = c3_kwargs(n, maxit)
    return onl.nleigs(olr, Sigma, errmeasure=osol.StandardSPMFErrmeasure(olr), **kw)


def c3_host_errors(n, lam, X):
    from oracle import gallery as og, solvers as osol
    onep = og.nlevp_native_gun(n)
    oE = osol.StandardSPMFErrmeasure(onep)
    return [float(oE(lam[i], X[:, i])) for i in range(len(lam))]


# ---- C4: contour_beyn on the unscaled gun SPMF, N = 64 nodes, k = 32 ------------------------------------------------------
C4_KW = dict(sigma=250.0 ** 2, radius=1e4, N=64, k=32, neigs=10 ** 6, tol=1e-6, sanity_check=True)


def c4_device(na, nep, integrator=None, Vh=None, info=None, **over):
    kw = dict(C4_KW, **over)
    if Vh is None:
        Vh = na.probe_block(nep.n, kw["k"])
    args = (nep,) if integrator is None else (nep, integrator)
    if info is not None:
        info.setdefault("moments", False)      # the n x k moment blocks stay on the device (contour_beyn downloads them for callers that ask)
    return na.contour_beyn(*args, Vh=Vh, info=info, **kw)


def c4_oracle(na, n=GUN_N, info=None, **over):
    from oracle import gallery as og, solvers as osol
    kw = dict(C4_KW, **over)
    onep = og.gun_spmf(n)
    return osol.contour_beyn(onep, Vh=na.probe_block(n, kw["k"]), info=info, **kw)


def c4_host_errors(n, lam, V):
    from oracle import gallery as og, solvers as osol
    onep = og.gun_spmf(n)
    oE = osol.StandardSPMFErrmeasure(onep)
    return [float(oE(lam[i], V[:, i])) for i in range(len(lam))]


# ---- C5: waveguide (WEP, JARLEBRING), tiar m = 60 ----------------------------------------------------------------------------
def c5_device(na, nx=1003, nz=999, solver="gmres", N=37, reltol=1e-9, refine=1, maxit=60, timers=None, restart=60, sweep_reltol=1e-6):
    """returns (lam, Q, residuals, info).  solver: "lu" = FactorizeLinSolver on the assembled M(sigma) (host SuperLU of an
    n = nx*nz + 2nz matrix), "gmres" = the reference's own solver for this problem (Schur complement + Sylvester-SMW
    preconditioned GMRES, Waveguide.jl:394-567).  reltol / refine: inner GMRES tolerance and refinement sweeps around it
    (measured at n = 1e6: (1e-6, 10) 2.9 s, (1e-9, 1) 2.2 s, same 7 eigenpairs and residuals; without a sweep the left-
    preconditioned residual GMRES controls is not the true one and no pair converges).  sweep_reltol: inner tolerance of the
    sweep (round 4: 1e-6 finds the same 7 pairs to 5e-12 in 22 -> 14 iterations per sweep; 1e-4 loses one pair)"""
    import torch
    t0 = time.perf_counter()
    nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); n = nep.n; nep.dev
    torch.cuda.synchronize()
    info = dict(n=n, generate_s=time.perf_counter() - t0, linsolver=solver)
    v0 = np.ones(n) / np.sqrt(n)
    kw = {}
    t1 = time.perf_counter()
    if solver != "lu":
        skw = ()
        if solver == "gmres":
            P = na.wep_generate_preconditioner(nep, N, -3 - 3.5j)
            torch.cuda.synchronize()
            info.update(preconditioner_N=N, preconditioner_setup_s=time.perf_counter() - t1)
            if os.environ.get("NEP_C5_SWEEP_RELTOL"):
                sweep_reltol = float(os.environ["NEP_C5_SWEEP_RELTOL"])
            if os.environ.get("NEP_C5_RELTOL"):
                reltol = float(os.environ["NEP_C5_RELTOL"])
            skw = (("Pl", P), ("reltol", reltol), ("restart", restart), ("maxiter", 300), ("sweep_reltol", sweep_reltol),
                   ("orth_meth", os.environ.get("NEP_GMRES_ORTH_FORCE", "cgs")))
            info.update(reltol=reltol, sweep_reltol=sweep_reltol)
        kw["linsolvercreator"] = na.WEPLinSolverCreator(solver_type=solver, kwargs=skw, refinements=refine)
    out = na.tiar(nep, sigma=-3 - 3.5j, gamma=1.0, maxit=maxit, neigs=np.inf, v=v0, tol=1e-8, timers=timers, **kw)
    torch.cuda.synchronize()
    info["solve_s"] = time.perf_counter() - t1
    lam, Q = out[0], out[1]
    R = na.ResidualErrmeasure(nep)
    res = [float(na.estimate_error(R, lam[i], Q[:, i])) for i in range(len(lam))]
    return lam, Q, res, info


def c5_host_residuals(nx, nz, lam, Q):
    """||M(lam) v|| / ||v|| of every returned pair re-evaluated in FP64 ON THE HOST by the oracle's matrix-free waveguide
    operator (oracle/wep.py WEP_FD: sparse stencils + FFT corner term, Waveguide.jl:204-379) -- independent of the device's K1"""
    from oracle import wep as ow
    o = ow.WEP_FD(nx, nz, "JARLEBRING")
    out = []
    for i in range(len(lam)):
        v = np.asarray(Q[:, i], dtype=complex)
        r = o._mlincomb(complex(lam[i]), v.reshape(-1, 1), np.ones(1, dtype=complex))
        out.append(float(np.linalg.norm(r) / np.linalg.norm(v)))
    return out


def c5_oracle_twin(nx=303, nz=299, maxit=60):
    """CPU oracle of config C5 on the reduced twin (assembled M(sigma) + SuperLU, oracle tiar, same start vector and
    tolerances).  Returns (lam, Q, seconds)."""
    from oracle import wep as ow, solvers as osol
    o = ow.WEP_FD(nx, nz, "JARLEBRING")
    n = o.n
    t0 = time.perf_counter()
    out = osol.tiar(o, sigma=-3 - 3.5j, gamma=1.0, maxit=maxit, neigs=np.inf, v=np.ones(n) / np.sqrt(n), tol=1e-8,
                    errmeasure=osol.ResidualErrmeasure(o))
    return out[0], out[1], time.perf_counter() - t0


class _SweptOracleCreator:
    """the oracle's matrix-free waveguide solver with `refine` residual-correction sweeps around it -- what the device
    configuration runs (c5_device: refinements=refine); the sweeps use the oracle's own M(sigma) product"""

    def __init__(self, inner, refine, progress=None):
        self.inner, self.refine, self.progress = inner, refine, progress

    def create_linsolver(self, nep, lam):
        s = self.inner.create_linsolver(nep, lam)
        self.last = s.iterations
        one = np.ones(1, dtype=complex)
        outer = self

        class _S:
            iterations = s.iterations

            def lin_solve(self, b, tol=0):
                b = np.asarray(b, dtype=complex).ravel()
                x = s.lin_solve(b)
                for _ in range(outer.refine):
                    r = b - nep._mlincomb(complex(lam), x.reshape(-1, 1), one).ravel()
                    x = x + s.lin_solve(r)
                if outer.progress is not None:
                    outer.progress(len(s.iterations), list(s.iterations[-(outer.refine + 1):]))
                return x
        return _S()


def c5_oracle_full(nx=1003, nz=999, N=37, reltol=1e-9, refine=1, maxit=60, restart=60, progress=None):
    """CPU oracle of config C5 by the reference's own route for this problem at ANY size (Waveguide.jl:427-567 +
    waveguide_preconditioner.jl:36-421 as restated in oracle/wep_linsolvers.py): Schur complement, Sylvester-SMW
    preconditioner with N x (N+4) regions, restarted GMRES, oracle tiar with the device configuration's start vector,
    tolerances, inner tolerance and refinement sweeps.  Returns (lam, Q, info) with the wall seconds of each part."""
    from oracle import wep as ow, solvers as osol, wep_linsolvers as owl
    info = {}
    t0 = time.perf_counter()
    o = ow.WEP_FD(nx, nz, "JARLEBRING")
    n = o.n
    info["generate_s"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    P = owl.wep_generate_preconditioner(o, N, -3 - 3.5j)
    info["preconditioner_setup_s"] = time.perf_counter() - t1
    inner = owl.WEPLinSolverCreator("gmres", (("Pl", P), ("reltol", reltol), ("restart", restart), ("maxiter", 300)))
    cr = _SweptOracleCreator(inner, refine, progress)
    t2 = time.perf_counter()
    out = osol.tiar(o, sigma=-3 - 3.5j, gamma=1.0, maxit=maxit, neigs=np.inf, v=np.ones(n) / np.sqrt(n), tol=1e-8,
                    errmeasure=osol.ResidualErrmeasure(o), linsolvercreator=cr)
    info["tiar_s"] = time.perf_counter() - t2
    info["seconds_solver"] = time.perf_counter() - t1           # what c5_device's solve_s covers: preconditioner + tiar
    info["n"] = n
    info["gmres_iterations"] = [int(i) for i in getattr(cr, "last", []) ]
    return out[0], out[1], info
