#!/usr/bin/env python
"""Runs the five BASELINE.json configurations on the device backend and, where cheap
enough, through the CPU oracle on the same inputs; prints one JSON line per configuration with eigenpair counts,
residuals, wall times and parity against the oracle.  Usage:  python scripts/run_configs.py [c1 c2 c3 c4 c5] [--oracle]
[--wep-nx NX --wep-nz NZ]"""
import argparse
import json
import os
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # small host BLAS only; big pools stall the launch thread
    os.environ.setdefault(_v, "8")
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

EPS = np.finfo(float).eps


def emit(**kw):
    print(json.dumps(kw, default=lambda o: float(o) if isinstance(o, (np.floating,)) else str(o)), flush=True)


def match(l1, l2, rtol):
    l2 = list(l2); worst = 0.0
    if len(l1) != len(l2):
        return False, None
    for x in l1:
        j = int(np.argmin([abs(x - y) for y in l2])); worst = max(worst, abs(x - l2[j]) / max(1.0, abs(x))); l2.pop(j)
    return worst <= rtol, worst


def gun_r1():
    gam = 300.0 ** 2 - 200.0 ** 2; mu = 250.0 ** 2; sigma2 = 108.8774
    xmin = gam * (-1) + mu; xmax = gam + mu
    th = np.linspace(0, np.pi, int(round(np.pi / 2 * 1000)) + 2)
    Sigma = np.concatenate([xmin + (xmax - xmin) * (np.exp(1j * th) / 2 + .5), [xmin]])
    nodes = gam * np.array([2 / 3, (1 + 1j) / 3, 0, (-1 + 1j) / 3, -2 / 3]) + mu
    Xi = -10.0 ** np.linspace(-8, 8, 10000) + sigma2 ** 2
    return Sigma, Xi, nodes


def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, time.perf_counter() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["c1", "c2", "c3", "c4", "c5"], help="c1..c5 and/or x (widened drivers)")
    ap.add_argument("--oracle", action="store_true", help="also run the (slow) CPU oracle for C2/C3")
    ap.add_argument("--wep-solver", default="lu", choices=["lu", "factorized", "backslash", "gmres"],
                    help="c5: lu = FactorizeLinSolver on the assembled M(sigma); others = WEPLinSolverCreator types")
    ap.add_argument("--wep-N", type=int, default=37, help="c5 gmres: regions per direction of the Sylvester-SMW preconditioner")
    ap.add_argument("--wep-reltol", type=float, default=1e-6)
    ap.add_argument("--wep-refine", type=int, default=10, help="c5 gmres: refinement sweeps around the GMRES solve (0 = reference behaviour)")
    ap.add_argument("--wep-nx", type=int, default=303)
    ap.add_argument("--wep-nz", type=int, default=299)
    args = ap.parse_args()
    import nep_amd as na
    from oracle import gallery as og, solvers as osol, neps as oneps

    if "x" in args.which:
        extras(na)

    if "c1" in args.which:
        nep = na.nep_gallery("dep0"); onep = og.dep0()
        (lam, v), t_first = timed(lambda: na.resinv(nep, lam=0, v=np.ones(5)))     # includes one-time library start-up
        (lam, v), t = timed(lambda: na.resinv(nep, lam=0, v=np.ones(5)))
        t0 = time.perf_counter(); lo, vo = osol.resinv(onep, lam=0, v=np.ones(5)); to = time.perf_counter() - t0
        emit(config="C1 dep0 n=5 resinv", gpu_lambda=[lam.real, lam.imag], cpu_lambda=[lo.real, lo.imag],
             backward_error=osol.DefaultErrmeasure(onep)(lam, v), gpu_s=t, gpu_first_call_s=t_first, cpu_s=to, parity=bool(abs(lam - lo) < 1e-10))

    if "c2" in args.which:
        nep = na.nep_gallery("gun_spmf_scaled"); n = nep.n; nep.dev
        run = lambda: na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(n), tol=1e-10)
        run()
        (lam, Q, _), t = timed(run)
        onep = og.gun_spmf_scaled()
        oE = osol.StandardSPMFErrmeasure(onep)
        res = max(oE(lam[i], Q[:, i]) for i in range(len(lam)))
        out = dict(config="C2 gun SPMF iar m=100", n=n, eigenpairs=len(lam), max_backward_error=res, gpu_s=t,
                   eigenpairs_per_s=len(lam) / t)
        if args.oracle:
            t0 = time.perf_counter()
            lo, Qo, _ = osol.iar(oneps.DerSPMF(onep, 0.0, 100), maxit=100, neigs=np.inf, v=np.ones(n), tol=1e-10, errmeasure=oE,
                                 linsolvercreator=osol.FactorizeLinSolverCreator(permc_spec="MMD_AT_PLUS_A"))
            to = time.perf_counter() - t0
            ok, worst = match(lam, lo, 1e-8)
            out.update(cpu_eigenpairs=len(lo), cpu_s=to, cpu_eigenpairs_per_s=len(lo) / to, parity=ok, max_rel_eig_diff=worst)
        emit(**out)

    if "c3" in args.which:
        # gun_nep() of test/rk_helper/gun_test_utils.jl:37-43: PEP + LowRankFactorizedNEP (ranks 19 + 65), as the
        # reference's variant R1 runs it; the same problem with full blocks (PEP + SPMF) is timed beside it
        Sigma, Xi, nodes = gun_r1()
        Kg, Mg, W1, W2 = na.gallery.gun_matrices()
        fv = [na.funcs.ISqrt(1.0, 0.0), na.funcs.ISqrt(1.0, -na.gallery.GUN_SIGMA2 ** 2)]
        nep = na.SumNEP(na.PEP([Kg, -Mg]), na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(W1, fv[0]),
                                                                      na.LowRankMatrixAndFunction(W2, fv[1])]))
        nep_full = na.nep_gallery("nlevp_native_gun"); n = nep.n; nep.dev; nep_full.dev
        v = np.random.Generator(np.random.Philox(1)).standard_normal(n) + 0j
        info = {}
        kw = dict(Xi=Xi, maxit=100, v=v, leja=0, nodes=nodes, reusefact=2, tol=1e-10)
        run = lambda: na.nleigs(nep, Sigma, errmeasure=na.StandardSPMFErrmeasure(nep), info=info, **kw)
        run()
        (lam, X, res), t = timed(run)
        (lamf, Xf, resf), tf = timed(lambda: na.nleigs(nep_full, Sigma, errmeasure=na.StandardSPMFErrmeasure(nep_full), **kw))
        onep = og.nlevp_native_gun()
        oE = osol.StandardSPMFErrmeasure(onep)
        out = dict(config="C3 gun nleigs variant R1 maxit=100 (PEP + LowRankFactorizedNEP, r = %d)" % info["lowrank_r"], n=n,
                   eigenpairs=len(lam), factorizations=info["nfact"], krylov_rows=info["vrows"],
                   max_backward_error=max([oE(lam[i], X[:, i]) for i in range(len(lam))] + [0.0]), gpu_s=t,
                   eigenpairs_per_s=len(lam) / t, ritz_in_sigma=info["nblamin"], full_blocks_gpu_s=tf,
                   full_blocks_same_eigenvalues=bool(match(lam, lamf, 1e-8)[0]))
        if args.oracle:
            from oracle import neps as on, nleigs as onl
            ofv = [on.f_isqrt(0.0), on.f_isqrt(-na.gallery.GUN_SIGMA2 ** 2)]
            olr = on.SumNEP(on.PEP([Kg, -Mg]), on.LowRankFactorizedNEP([on.LowRankMatrixAndFunction(W1, ofv[0]),
                                                                         on.LowRankMatrixAndFunction(W2, ofv[1])]))
            t0 = time.perf_counter()
            lo, Xo, ro = onl.nleigs(olr, Sigma, errmeasure=osol.StandardSPMFErrmeasure(olr), **kw)
            to = time.perf_counter() - t0
            ok, worst = match(lam, lo, 1e-8)
            out.update(cpu_eigenpairs=len(lo), cpu_s=to, parity=ok, max_rel_eig_diff=worst)
        emit(**out)

    if "c4" in args.which:
        nep = na.nep_gallery("gun_spmf"); n = nep.n; nep.dev
        onep = og.gun_spmf()
        Vh = na.probe_block(n, 32)
        kw = dict(sigma=250.0 ** 2, radius=1e4, N=64, k=32, neigs=10 ** 6, tol=1e-6, sanity_check=True)
        ig = {}
        run = lambda: na.contour_beyn(nep, Vh=Vh, info=ig, **kw)
        (lam, V), t = timed(run)
        oE = osol.StandardSPMFErrmeasure(onep)
        t0 = time.perf_counter(); io = {}
        lo, Vo = osol.contour_beyn(onep, Vh=Vh, info=io, **kw)
        to = time.perf_counter() - t0
        ok, worst = match(lam, lo, 1e-7)
        emit(config="C4 gun contour_beyn N=64 k=32 radius=1e4 (1 GPU)", n=n, eigenpairs=len(lam), rank_p=ig["p"],
             cpu_rank_p=io["p"], max_backward_error=max([oE(lam[i], V[:, i]) for i in range(len(lam))] + [0.0]), gpu_s=t,
             eigenpairs_per_s=len(lam) / t, cpu_eigenpairs=len(lo), cpu_s=to, cpu_eigenpairs_per_s=len(lo) / to, parity=ok,
             max_rel_eig_diff=worst)

    if "c5" in args.which:
        nx, nz = args.wep_nx, args.wep_nz
        t0 = time.perf_counter()
        nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); n = nep.n; nep.dev
        tgen = time.perf_counter() - t0
        v0 = np.ones(n) / np.sqrt(n)
        tm = {}
        kw = {}
        extra = {}
        if args.wep_solver != "lu":
            # the reference's own solver for this problem: Schur complement of the boundary block (Waveguide.jl:552-567),
            # optionally matrix-free GMRES with the Sylvester-SMW preconditioner (waveguide_preconditioner.jl)
            skw = ()
            if args.wep_solver == "gmres":
                tp = time.perf_counter()
                P = na.wep_generate_preconditioner(nep, args.wep_N, -3 - 3.5j)
                torch.cuda.synchronize()
                extra = dict(preconditioner_N=args.wep_N, preconditioner_setup_s=time.perf_counter() - tp, smw_cond=P.cond)
                skw = (("Pl", P), ("reltol", args.wep_reltol), ("restart", 60), ("maxiter", 300), ("orth_meth", "dgks"))
            kw["linsolvercreator"] = na.WEPLinSolverCreator(solver_type=args.wep_solver, kwargs=skw, refinements=args.wep_refine)
            extra["refinements"] = args.wep_refine
        run = lambda: na.tiar(nep, sigma=-3 - 3.5j, gamma=1.0, maxit=60, neigs=np.inf, v=v0, tol=1e-8, timers=tm, **kw)
        (out4, t) = timed(run)
        t += extra.get("preconditioner_setup_s", 0.0)
        extra["linsolver"] = args.wep_solver
        lam, Q = out4[0], out4[1]
        R = na.ResidualErrmeasure(nep)
        res = [na.estimate_error(R, lam[i], Q[:, i]) for i in range(len(lam))]
        emit(config="C5 WEP JARLEBRING tiar m=60", nx=nx, nz=nz, n=n, eigenpairs=len(lam), max_residual=max(res + [0.0]),
             gpu_s=t, eigenpairs_per_s=len(lam) / t, generate_s=tgen, phases_s={k_: round(v_, 4) for k_, v_ in tm.items()},
             eigenvalues=[[l.real, l.imag] for l in lam[:6]], **extra)


def extras(na):
    """the widened drivers on their reference examples: wall time and the reference's own check"""
    import warnings
    shift, scale = 250.0 ** 2, 330.0 ** 2 - 220.0 ** 2
    gun = na.nep_gallery("nlevp_native_gun"); n = gun.n
    nep1 = na.nep_gallery("gun_spmf_scaled"); nep1.dev
    Av, fv = gun.get_Av(), gun.get_fv()
    res = lambda lo, x: float(np.linalg.norm(sum(f.derivs(lo, 1)[0] * (A @ x) for f, A in zip(fv, Av))))
    (D, X, _), t = timed(lambda: na.nlar(nep1, tol=1e-10, lam=0, maxit=100, neigs=2, R=0.01, v=np.ones(n),
                                         inner_solver_method=na.IARInnerSolver(), num_restart_ritz_vecs=8, max_subspace=150))
    emit(config="X1 gun nlar (test/nlar.jl:28-31)", n=n, eigenpairs=2, gpu_s=t,
         residuals=[res(shift + scale * D[i], X[:, i]) for i in range(2)], threshold=float(np.sqrt(1e-10) * 50))
    (lam, Q, _), t = timed(lambda: na.iar(nep1, maxit=40, neigs=np.inf, v=np.ones(n), tol=1e-10, check_error_every=10,
                                          proj_solve=True, inner_solver_method=na.IARInnerSolver(maxit=60)))
    emit(config="X2 gun iar m=40 proj_solve=true (checks every 10)", n=n, eigenpairs=len(lam), gpu_s=t)
    d100 = na.nep_gallery("dep0", 100)
    (lam, V, _), t = timed(lambda: na.iar_chebyshev(d100, v=np.ones(100), tol=1e-5, neigs=3))
    ref = np.array([0.050462487848960284, -0.07708779190301127, 0.1503856540695659])
    emit(config="X3 dep0(100) iar_chebyshev docstring (method_iar_chebyshev.jl:45-56)", eigenpairs=len(lam), gpu_s=t,
         max_abs_diff_to_docstring=float(np.max(np.abs(lam.real - ref))))
    ds = na.nep_gallery("dep_symm_double", 10)
    out, t = timed(lambda: na.ilan(ds, v=np.ones(100), tol=1e-5, neigs=12))
    refl = np.array([0.03409997385842267, -0.03100798730589012, -0.0367653644764646])
    emit(config="X4 dep_symm_double(10) ilan docstring (method_ilan.jl:41-52)", eigenpairs=len(out[0]), gpu_s=t,
         max_dist_docstring_eigs=float(max(np.min(np.abs(out[0] - r)) for r in refl)))
    g = na.nep_gallery("gun_spmf"); g.dev
    na.HostLUPool.warm()
    (lam, V), t = timed(lambda: na.contour_block_SS(g, sigma=250.0 ** 2, radius=1e4, N=64, k=8, K=4, rank_drop_tol=1e-10,
                                                    Shat_mode="JSIAM"))
    E = na.StandardSPMFErrmeasure(g)
    errs = np.array([na.estimate_error(E, lam[i], V[:, i]) for i in range(len(lam))])
    emit(config="X5 gun contour_block_SS N=64 L=8 K=4 (JSIAM moments)", ritz_values=len(lam),
         eigenpairs_backward_error_below_1em6=int(np.sum(errs < 1e-6)), gpu_s=t)


if __name__ == "__main__":
    main()
