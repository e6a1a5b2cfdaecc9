"""Infinite Lanczos for symmetric NEPs on the device backend -- keyword surface of src/method_ilan.jl:56-75.

Per step (method_ilan.jl:120-166), all blocks n x (k+1) column-major and device resident:
    Qn[:,1:k+1] = Q[:,0:k] ./ (1:k)          nep_iar_shift_scale (one pass)
    Qn[:,0] = -M(sigma)^{-1} sum_j a_j M^(j)(sigma) Qn[:,j]      K1 (coefficient block from the derivative table) + K5
    Z = sum_t A_t (Qn (G o FDH_t))           Bmult, SPMF version (:370-378): K7 GEMMs into one row-major block +
                                             ONE SpMM over the stacked CSR (nep_spmm_terms), transposed back
    alpha, beta, eta = mat_sum(...)          bilinear (non-conjugated) sums: nep_coldotsu on the contiguous blocks
    three-term recurrence, norm, scaling     nep_axpy, nep_nrm2, nep_scal
    orthogonalisation of the first blocks    K6
Extraction by Galerkin projection on span(V) + inner solve (proj_solve=true, the reference default) or by Ritz pairs.
The reference's DEP version of Bmult is the same product in factored form (test/ilan.jl:45-62 checks that the two give the
same iterates); the SPMF version is used for every SPMF-type NEP here.
"""

import numpy as np
import torch

from . import dense, _lib
from ._lib import lib, check, hptr, c_vp
from .errmeasure import DefaultErrmeasure, estimate_errors
from .exceptions import NoConvergenceException
from .iar import _hosteig
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, to_dev, to_host, stream_ptr

EPS = np.finfo(float).eps


def symmetrizer_coefficients(m):
    """method_ilan.jl:419-427"""
    G = np.zeros((m + 1, m + 1), dtype=np.complex128)
    G[:, 0] = 1.0 / np.arange(1, m + 2)
    for j in range(m):
        for i in range(m + 1):
            G[i, j + 1] = G[i, j] * (j + 1) / (i + j + 2)
    return G


def _dotu(x, y, length):
    out = np.empty(1, dtype=np.complex128)
    check(lib.nep_coldotsu(length, 1, c_vp(x.data_ptr()), length, c_vp(y.data_ptr()), length, hptr(out), stream_ptr()))
    return out[0]


def ilan(nep, orthmethod=dense.DGKS, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6, errmeasure=None, sigma=0.0,
         gamma=1.0, v=None, logger=0, check_error_every=30, inner_solver_method=None, proj_solve=True, inner_logger=0):
    from .nep import require_pure_spmf
    require_pure_spmf(nep, "ilan")
    n = nep.size(1); m = int(maxit)
    sigma = complex(sigma); gamma = complex(gamma)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if v is None:
        v = np.random.randn(n)
    fv = nep.get_fv(); mt = len(fv)
    st = stream_ptr
    V = torch.zeros((m + 1, n), dtype=CDT, device="cuda")
    Q = torch.zeros((m + 1, n), dtype=CDT, device="cuda")
    Qp = torch.zeros_like(Q); Qn = torch.zeros_like(Q); Z = torch.zeros_like(Q)
    QQ = None if proj_solve else torch.zeros_like(Q)
    H = np.zeros((m + 1, m), dtype=np.complex128); HH = np.zeros((m + 1, m), dtype=np.complex128)
    om = np.zeros(m + 1, dtype=np.complex128)
    M0inv = create_linsolver(linsolvercreator, nep, sigma)
    # precompute_data (:322-343): FDH_t[i,j] = gamma^(i+j+1) f_t^(i+j+1)(sigma); K1 table a_j f^(j)(sigma), a = gamma^j, a_0 = 0
    fD = np.column_stack([f.derivs(sigma, 2 * m + 2, gamma) for f in fv])              # (2m+2) x mt
    FDH = [np.array([[fD[i + j + 1, t] for j in range(m + 1)] for i in range(m + 1)]) for t in range(mt)]
    G = symmetrizer_coefficients(m)
    a = gamma ** np.arange(2 * m + 3); a[0] = 0
    v0 = np.asarray(v, dtype=np.complex128)
    Q[0] = to_dev(v0 / np.linalg.norm(v0))[0]
    w0 = nep.compute_Mlincomb(0.0, torch.stack([Q[0], Q[0]]), [0, 1])                     # M'(0) q_0
    out1 = np.empty(1, dtype=np.complex128)
    check(lib.nep_coldots(n, 1, c_vp(Q[0].data_ptr()), n, c_vp(w0.data_ptr()), n, hptr(out1), st()))
    om[0] = out1[0]
    dense.copy(Q[0], V[0], n)
    z = torch.empty(n, dtype=CDT, device="cuda")
    XT = torch.empty((n, mt * (m + 1)), dtype=CDT, device="cuda")
    ZT = torch.empty((n, m + 1), dtype=CDT, device="cuda")
    err = np.full((m, m + 1), np.nan)
    lam = np.zeros(0, dtype=np.complex128); WT = None; idx = np.zeros(0, dtype=int)
    k = 1; conv_eig = 0
    while k <= m and conv_eig < neigs:
        if not proj_solve:
            dense.copy(Q[0], QQ[k - 1], n)
        check(lib.nep_iar_shift_scale(n, k, c_vp(Q.data_ptr()), c_vp(Qn.data_ptr()), st()))       # Qn[:,j+1] = Q[:,j]/(j+1)
        Cm = nep.coeff_block(sigma, a[:k + 1], 0) if gamma == 1 else None
        if Cm is None:                                    # a_j f^(j)(sigma) with a = gamma^j: derivative table with scale
            Cm = np.column_stack([f.derivs(sigma, k + 1, gamma) for f in fv]); Cm[0, :] = 0
        nep.dev.mlincomb(Cm, Qn[:k + 1], z)
        M0inv.solve_dev(z, out=Qn[0].reshape(1, n), scale=-1.0)
        # ---- Bmult: Z[:, :k+1] = sum_t A_t (Qn[:, :k+1] (G o FDH_t))
        kk = k + 1
        for t in range(mt):
            F = _lib.as_c128(G[:kk, :kk] * FDH[t][:kk, :kk], "F")
            check(lib.nep_gemm_ts(c_vp(Qn.data_ptr()), n, n, kk, hptr(F), kk, kk, c_vp(XT.data_ptr() + 16 * t * kk), mt * kk, 1, st()))
        check(lib.nep_spmm_terms(nep.dev.h, kk, c_vp(XT.data_ptr()), mt * kk, c_vp(ZT.data_ptr()), kk, st()))
        check(lib.nep_rowmajor_to_colmajor(n, kk, c_vp(ZT.data_ptr()), kk, None, kk, c_vp(Z.data_ptr()), n, st()))
        # ---- three-term recurrence (:137-165)
        beta = _dotu(Z, Qp, k * n) if k > 1 else 0.0
        alpha = _dotu(Z, Q, k * n)
        eta = _dotu(Z, Qn, kk * n)
        H[k - 1, k - 1] = alpha / om[k - 1]
        if k > 1:
            H[k - 2, k - 1] = beta / om[k - 2]
        dense.axpy(-H[k - 1, k - 1], Q, Qn, k * n)
        if k > 1:
            dense.axpy(-H[k - 2, k - 1], Qp, Qn, k * n)
        H[k, k - 1] = dense.nrm2(Qn, kk * n)
        dense.scal(Qn, 1.0 / H[k, k - 1], kk * n)
        om[k] = eta - 2 * alpha * H[k - 1, k - 1] + om[k - 1] * H[k - 1, k - 1] ** 2
        if k > 1:
            om[k] = om[k] - 2 * beta * H[k - 2, k - 1] + om[k - 2] * H[k - 2, k - 1] ** 2
        om[k] = om[k] / H[k, k - 1] ** 2
        dense.copy(Qn[0], V[k], n)
        h, hb, _ = dense.orthogonalize_and_normalize(V, V[k], k, rows=n, ldv=n, method=orthmethod)
        HH[:k, k - 1] = h
        # ---- extraction
        if (check_error_every != np.inf and k % check_error_every == 0) or k == m:
            if not proj_solve:
                D, WR = _hosteig.eig(H[:k, :k].copy())
                WT = dense.gemm_ts(QQ, WR, rowmajor=True, k=k, rows=n, ldz=n)
                lam = sigma + gamma / D
            else:
                from .projection import create_proj_NEP, inner_solve, DefaultInnerSolver
                pnep = create_proj_NEP(nep, kk)
                pnep.set_projectmatrices(V[:kk], V[:kk])
                lamp, Wp = inner_solve(inner_solver_method or DefaultInnerSolver(), pnep, neigs=m, tol=tol)
                q = min(len(lamp), m)
                lam = np.asarray(lamp)[:q]
                WT = dense.gemm_ts(V, np.asarray(Wp)[:, :q], rowmajor=True, k=kk, rows=n, ldz=n) if q else None
            e = estimate_errors(errmeasure, lam, WT) if len(lam) else np.zeros(0)
            ne = len(e)
            err[k - 1, :ne] = e
            conv_eig = int(np.sum(e < tol))
            idx = np.argsort(err[k - 1, :k], kind="stable")
            if k == m or conv_eig >= neigs:
                nrof = int(min(conv_eig, neigs))
                idx = idx[:nrof]
                lam = lam[idx]
        k += 1
        Qp, Q, Qn = Q, Qn, Qp
        Qn.zero_()
    k -= 1
    if conv_eig < neigs and neigs != np.inf:
        Wh = to_host(dense.rowmajor_to_cols(WT, idx[:len(lam)])) if WT is not None and len(idx) else None
        msg = "Number of iterations exceeded. maxit=%d." % maxit
        if conv_eig < 3:
            msg += "Try to change the inner_solver_method for better performance."
        raise NoConvergenceException(lam, Wh, None, msg)
    Wh = to_host(dense.rowmajor_to_cols(WT, idx[:len(lam)])) if WT is not None and len(lam) else np.zeros((n, 0), dtype=complex)
    return lam, Wh, to_host(V[:k + 1]), H[:k, :k - 1].copy(), om[:k].copy(), HH[:k, :k].copy()
