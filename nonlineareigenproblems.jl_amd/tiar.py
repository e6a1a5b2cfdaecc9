"""Tensor infinite Arnoldi (tiar) on the device backend -- keyword surface of src/method_tiar.jl:53-70.

Device-resident: the basis Z (n x (m+1)).  Host: the (m+1)^3 coefficient tensor `a` and the O(k^3)
tensor algebra of method_tiar.jl:131-183 (3.6 MB at m=60).

Reference step (method_tiar.jl:116-239)               device realisation
  y[:,2:k+1]=Z[:,1:k]*transpose(a[1:k,k,1:k]) ./(1:k)'   fused=True : no GEMM at all -- the k x k matrix is
  y[:,1]=compute_Mlincomb!(nep,s,y[:,1:k+1],alpha)          multiplied into the k x m_t coefficient block on the
                                                            host, K1 then runs directly on Z  (z = sum_i A_i Z (B c_i))
                                                          fused=False: K7 nep_gemm_ts then K1, as the reference
  y[:,1]=-lin_solve(M0inv,y[:,1]); Z[:,k+1]=y[:,1]        K5 nep_lu_solve(scale=-1) straight into Z[:,k+1]
  orthogonalize_and_normalize!(Z[:,1:k],Z[:,k+1],t,DGKS)  K6 nep_orth
  VV=Z[:,1:k]*transpose(a[1,1:k,1:k]); Q=VV*W              ONE K7 GEMM with the k x k product formed on the host
  err[k,s]=estimate_error(...)                             K2 nep_resid_batch
"""
import os
import time

import numpy as np
import scipy.linalg as sla
import torch

from . import dense
from .errmeasure import DefaultErrmeasure, estimate_errors
from .exceptions import NoConvergenceException, LostOrthogonalityException
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, to_host, to_host_cm

EPS = np.finfo(float).eps


def tiar(nep, orthmethod=dense.DGKS, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6,
         errmeasure=None, sigma=0.0, gamma=1.0, v=None, logger=0, check_error_every=1, proj_solve=False,
         errhist=None, timers=None, fused=True, return_device=False, inner_solver_method=None):
    n = nep.size(1); m = int(maxit)
    sigma = complex(sigma); gamma = complex(gamma)
    if n < m:
        raise LostOrthogonalityException("Loss of orthogonality in the matrix Z. The problem size is too "
                                         "small, use iar instead.")
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if v is None:
        v = np.random.randn(n)
    tm = timers if timers is not None else {}
    for key in ("mlincomb", "solve", "orth", "host_tensor", "host_eig", "ritz", "resid"):
        tm.setdefault(key, 0.0)
    sync = torch.cuda.synchronize if timers is not None else (lambda: None)

    a = np.zeros((m + 1, m + 1, m + 1), dtype=np.complex128)
    Z = torch.zeros((m + 1, n), dtype=CDT, device="cuda")
    t = np.zeros(m + 1, dtype=np.complex128)
    g = np.zeros((m + 1, m + 1), dtype=np.complex128)
    H = np.zeros((m + 1, m), dtype=np.complex128)
    alpha = gamma ** np.arange(m + 1); alpha[0] = 0
    t_ls = time.perf_counter()
    M0inv = create_linsolver(linsolvercreator, nep, sigma)
    sync(); tm["linsolver_setup"] = tm.get("linsolver_setup", 0.0) + time.perf_counter() - t_ls
    if timers is not None and hasattr(M0inv, "lu"):
        tm["host_factorization"] = tm.get("host_factorization", 0.0) + M0inv.lu.t_factor
    err = np.full((m + 1, m + 1), np.nan)
    v0 = np.asarray(v, dtype=np.complex128)
    Z[0] = torch.from_numpy(v0 / np.linalg.norm(v0)).to("cuda")
    a[0, 0, 0] = 1
    tab = nep.derivative_table(sigma, m)
    z = torch.empty(n, dtype=CDT, device="cuda")
    conv_eig_hist = np.zeros(m + 1, dtype=int)
    lam = np.zeros(0, dtype=np.complex128); QT = None; idx = np.zeros(0, dtype=int)
    # large sparse problems keep the Ritz block of a check column-major (tiled K2 with contiguous column loads)
    ritz_cm = bool(getattr(nep, "prefers_colmajor_ritz", lambda: False)())
    pnep = None
    if proj_solve:                                           # method_tiar.jl:104-106
        from .projection import create_proj_NEP, inner_solve, DefaultInnerSolver
        pnep = create_proj_NEP(nep, maxsize=min(n, m + 1))
        if inner_solver_method is None:
            inner_solver_method = DefaultInnerSolver()
        err = np.full((m + 1, m + 4), np.nan)
    # neigs = Inf: no check can end the iteration, so the checks are DEFERRED -- eig(H_k) goes to a worker thread as soon as column k
    # of H exists (LAPACK runs without the GIL), Ritz block + residual batch of all steps are issued back to back behind the
    # recurrence with the host preparing check k + 1 while the device works on check k.  Same arithmetic, same history, same
    # results; the recurrence no longer waits twice per step for a host eigen-decomposition and two read-backs (config C5:
    # 60 steps x ~2.5 ms).  Instrumented runs (timers), proj_solve and NEP_TIAR_DEFER=0 keep the step-synchronous order.
    defer = bool(np.isinf(neigs)) and timers is None and not proj_solve and os.environ.get("NEP_TIAR_DEFER", "1") != "0"
    deferred = []
    eig_pool = None
    if defer:
        from concurrent.futures import ThreadPoolExecutor
        eig_pool = ThreadPoolExecutor(max_workers=2)

    def record_check(kk, laml, QTl, e):
        nonlocal lam, QT, idx, conv_eig
        ne = len(e)
        idxl = np.argsort(e, kind="stable")
        err[kk - 1, :ne] = e[idxl]
        conv_eig = int(np.sum(e < tol))
        if errhist is not None:
            errhist.append(err[kk - 1, :ne].copy())
        lam, QT, idx = laml, QTl, idxl
        if kk == m or conv_eig >= neigs:
            nrof = int(min(len(lam), neigs))
            lam = lam[idx[:nrof]]
            idx = idx[:nrof]
        conv_eig_hist[kk - 1] = conv_eig

    k = 1; conv_eig = 0
    while k <= m and conv_eig < neigs:
        t0 = time.perf_counter()
        Bs = a[:k, k - 1, :k].T / np.arange(1, k + 1)[None, :]          # k x k, column j scaled by 1/(j+1)
        if fused:
            # z = sum_i A_i Z (Bs diag(alpha) fD_i): coefficient matrix G = Bs * alpha (k x k), no GEMM on Z
            nep.lincomb_general(tab, Bs * alpha[None, 1:k + 1], Z, k, n, z)
        else:
            Y = dense.gemm_ts(Z, Bs, k=k, rows=n, ldz=n)
            nep.lincomb_general(tab, np.diag(alpha[1:k + 1]), Y, k, n, z)
        sync(); t1 = time.perf_counter()
        M0inv.solve_dev(z, out=Z[k].reshape(1, n), scale=-1.0)
        sync(); t2 = time.perf_counter()
        h_, beta_, _ = dense.orthogonalize_and_normalize(Z, Z[k], k, rows=n, ldv=n, method=orthmethod)
        t[:k] = h_; t[k] = beta_
        sync(); t3 = time.perf_counter()
        # ---- host tensor algebra (method_tiar.jl:131-183); f and ff alias g as in the reference
        for l in range(k + 1):
            g[1:k + 1, l] = a[:k, k - 1, l] / np.arange(1, k + 1)
            g[0, l] = t[l]
        h = np.zeros(m + 1, dtype=np.complex128)
        for l in range(k):
            h[:k] += a[:k, :k, l].conj().T @ g[:k, l]
        f = g
        for l in range(k):
            f[:k + 1, l] -= a[:k + 1, :k, l] @ h[:k]
        hh = np.zeros(m + 1, dtype=np.complex128)
        for l in range(k):
            hh[:k] += a[:k, :k, l].conj().T @ f[:k, l]
        for l in range(k):
            f[:k + 1, l] -= a[:k + 1, :k, l] @ hh[:k]
        h = h + hh
        beta = np.linalg.norm(f[:k + 1, :k + 1])
        H[:k, k - 1] = h[:k]; H[k, k - 1] = beta
        a[:k + 1, k, :k + 1] = f[:k + 1, :k + 1] / beta
        t4 = time.perf_counter()
        tm["mlincomb"] += t1 - t0; tm["solve"] += t2 - t1; tm["orth"] += t3 - t2; tm["host_tensor"] += t4 - t3
        if defer and ((k % check_error_every == 0) or (k == m)):
            deferred.append((k, eig_pool.submit(sla.eig, H[:k, :k].copy())))
        elif (k % check_error_every == 0) or (k == m):
            D, W = sla.eig(H[:k, :k])
            t5 = time.perf_counter()
            lam = sigma + gamma / D
            if proj_solve:
                # method_tiar.jl:192-207: Galerkin projection on span(Z_k) (Z is orthonormal), inner solve, lift back
                pnep.set_projectmatrices(Z[:k], Z[:k])
                lamp, Qp = inner_solve(inner_solver_method, pnep, lamv=lam.copy(), neigs=len(lam) + 3, sigma=sigma,
                                       tol=tol / 10)
                II = np.argsort(abs(lamp - sigma), kind="stable")
                lam = np.asarray(lamp)[II]; Qp = np.asarray(Qp)[:, II]
                QT = dense.gemm_ts(Z, Qp, rowmajor=not ritz_cm, k=k, rows=n, ldz=n)
            else:
                QT = dense.gemm_ts(Z, a[0, :k, :k].T @ W, rowmajor=not ritz_cm, k=k, rows=n, ldz=n)
            if ritz_cm:
                QT = dense.ColMajorBlock(QT)
            sync(); t6 = time.perf_counter()
            e = estimate_errors(errmeasure, lam, QT) if len(lam) else np.zeros(0)
            t7 = time.perf_counter()
            tm["host_eig"] += t5 - t4; tm["ritz"] += t6 - t5; tm["resid"] += t7 - t6
            ne = len(e)
            err[k - 1, :ne] = e
            conv_eig = int(np.sum(e < tol))
            idx = np.argsort(e, kind="stable")
            err[k - 1, :ne] = e[idx]
            if errhist is not None:
                errhist.append(err[k - 1, :ne].copy())
            if k == m or conv_eig >= neigs:
                nrof = int(min(len(lam), neigs))
                lam = lam[idx[:nrof]]
                idx = idx[:nrof]
            conv_eig_hist[k - 1] = conv_eig
        k += 1
    k -= 1
    if defer:
        from .errmeasure import estimate_errors_async
        prev = None
        for kk, fut in deferred:
            D, W = fut.result()
            laml = sigma + gamma / D
            QTl = dense.gemm_ts(Z, a[0, :kk, :kk].T @ W, rowmajor=not ritz_cm, k=kk, rows=n, ldz=n)
            if ritz_cm:
                QTl = dense.ColMajorBlock(QTl)
            pend = estimate_errors_async(errmeasure, laml, QTl) if len(laml) else None
            if prev is not None:
                record_check(prev[0], prev[1], prev[2], prev[3].get() if prev[3] is not None else np.zeros(0))
            prev = (kk, laml, QTl, pend)
        if prev is not None:
            record_check(prev[0], prev[1], prev[2], prev[3].get() if prev[3] is not None else np.zeros(0))
        eig_pool.shutdown(wait=False)
    if conv_eig < neigs and neigs != np.inf:
        Q = to_host(dense.rowmajor_to_cols(QT, idx[:len(lam)])) if QT is not None else None
        msg = "Number of iterations exceeded. maxit=%d." % maxit
        if conv_eig < 3:
            msg += " Check that σ is not an eigenvalue."
        raise NoConvergenceException(lam, Q, err[k - 1, :len(lam)], msg)
    nc = min(len(lam), conv_eig)
    lam = lam[:nc]
    Qd = dense.rowmajor_to_cols(QT, idx[:nc])
    if return_device:
        return lam, Qd, Z[:k], conv_eig_hist
    return lam, to_host_cm(Qd), Z[:k], conv_eig_hist
