"""Infinite Arnoldi, Chebyshev version, on the device backend -- keyword surface of src/method_iar_chebyshev.jl:66-84.

Same kernels as `iar`, different host recurrences.  With X = the n x k block held in column k
of the basis (Chebyshev coefficients of the current function), per step
    blocks 1..k of the new vector   X * L[0:k,0:k]                              K7 nep_gemm_ts straight into V
    y0 (method_iar_chebyshev.jl:309-366)  every version is  +-M(sigma)^{-1} sum_t A_t (X c_t)  [- X (L T(c))]  with a
       k x m_t coefficient block c_t built on the host from Chebyshev values / the derivation matrix D / the divided
       differences f_t[sigma I + gamma D, sigma]:                                 ONE K1 call + K5 (+ K7 with p = 1)
    orthogonalisation, Ritz extraction, residuals                               K6, K7, K2 as in iar
DEP, PEP and general SPMF versions of compute_y0_cheb are built; a DEP or PEP with sigma != 0 or gamma != 1 goes through
the SPMF version (which carries shift and scale) instead of the reference's explicit shift_and_scale.
"""
import numpy as np
import torch

from . import dense
from .errmeasure import DefaultErrmeasure, estimate_errors
from .exceptions import NoConvergenceException
from .iar import _hosteig
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, DEP, PEP, to_dev, to_host

EPS = np.finfo(float).eps


def _cheb_L(m, a, b):
    """integration map in the Chebyshev basis of [a, b] (method_iar_chebyshev.jl:129-131)"""
    L = np.diag(np.concatenate([[2.0], 1.0 / np.arange(2, m + 1)])) + np.diag(-1.0 / np.arange(1, m - 1), -2)
    return L * (b - a) / 4


def _cheb_T_at(x, idx):
    """T_i(x), real x inside or outside [-1, 1] (method_iar_chebyshev.jl:245-253)"""
    idx = np.asarray(idx, dtype=float)
    if abs(x) <= 1:
        return np.cos(idx * np.arccos(x))
    if x >= 1:
        return np.cosh(idx * np.arccosh(x))
    return ((-1.0) ** idx) * np.cosh(idx * np.arccosh(-x))


def _dd0_mat_fun(f, S, sigma):
    """f[S, sigma I] through f([[S, I], [0, sigma I]])  (method_iar_chebyshev.jl:474-497); f: funcs.ScalarFun"""
    n = S.shape[0]
    A = np.zeros((2 * n, 2 * n), dtype=complex)
    A[:n, :n] = S; A[:n, n:] = np.eye(n); A[n:, n:] = sigma * np.eye(n)
    return np.asarray(f.matfun(A))[:n, n:]


def iar_chebyshev(nep, orthmethod=dense.DGKS, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6, errmeasure=None,
                  sigma=0.0, gamma=1.0, v=None, logger=0, check_error_every=1, compute_y0_method="auto", a=None, b=None,
                  errhist=None, return_device=False):
    from .nep import require_pure_spmf
    require_pure_spmf(nep, "iar_chebyshev")
    n = nep.size(1); m = int(maxit)
    sigma = complex(sigma); gamma = complex(gamma)
    isdep = isinstance(nep, DEP); ispep = isinstance(nep, PEP)
    if a is None:
        a = -float(np.max(nep.tauv)) if isdep else -1.0
    if b is None:
        b = 0.0 if isdep else 1.0
    if compute_y0_method == "auto":
        compute_y0_method = "DEP" if isdep else ("PEP" if ispep else "SPMF")
    if compute_y0_method not in ("DEP", "PEP", "SPMF"):
        raise NotImplementedError("compute_y0_method %r: the DEP, PEP and SPMF versions are built" % (compute_y0_method,))
    if (sigma != 0 or gamma != 1) and compute_y0_method in ("DEP", "PEP"):
        compute_y0_method = "SPMF"
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if v is None:
        v = np.random.randn(n)
    cc = (a + b) / (a - b); kk = 2 / (b - a)
    fv = nep.get_fv(); mt = len(fv)
    L = _cheb_L(m, a, b)
    Tc = np.cos(np.arange(m + 1) * np.arccos(cc))
    if compute_y0_method == "DEP":
        Ttau = np.array([_cheb_T_at(-kk * tau + cc, np.arange(m + 2)) for tau in nep.tauv])
    else:
        Li = np.linalg.inv(L[:m, :m])
        D = np.vstack([np.zeros((1, m)), Li[:m - 1, :]])
        if compute_y0_method == "SPMF":
            DDf = [gamma * _dd0_mat_fun(f, sigma * np.eye(m) + gamma * D, sigma) for f in fv]

    def coefficients(N):
        """k x m_t block C with z = sum_t A_t (X C[:, t]); and the sign of the solve"""
        C = np.zeros((N, mt), dtype=np.complex128, order="F")
        if compute_y0_method == "DEP":             # :309-321  y0 = M0inv (X Tc - sum_j A_{j+1} Y Ttau_j), Y = [0, X L]
            C[:, 0] = Tc[:N]
            for j in range(len(nep.tauv)):
                C[:, j + 1] = -(L[:N, :N] @ Ttau[j, 1:N + 1])
            return C, 1.0
        if compute_y0_method == "PEP":             # :331-343
            vv_ = Tc[:N].astype(complex)
            for j in range(mt - 1):
                C[:, j + 1] = vv_
                vv_ = D[:N, :N] @ vv_
            return C, -1.0
        for i in range(mt):                        # :355-366
            C[:, i] = DDf[i][:N, :N] @ Tc[:N]
        return C, -1.0

    ldv = n * (m + 1)
    V = torch.zeros((m + 1, ldv), dtype=CDT, device="cuda")
    H = np.zeros((m + 1, m), dtype=np.complex128)
    M0inv = create_linsolver(linsolvercreator, nep, sigma)
    v0 = np.asarray(v, dtype=np.complex128)
    V[0, :n] = to_dev(v0 / np.linalg.norm(v0))[0]
    z = torch.empty(n, dtype=CDT, device="cuda")
    active = (np.arange(1, m + 2) * n).astype(np.int64)
    err = np.ones((m, m))
    lam = np.zeros(0, dtype=np.complex128); QT = None; idx = np.zeros(0, dtype=int)
    k = 1; conv_eig = 0
    while k <= m and conv_eig < neigs:
        X = V[k - 1][:n * k].view(k, n)                                   # n x k block, column-major, ld n
        vv = V[k]
        dense.gemm_ts(X, L[:k, :k], k=k, rows=n, ldz=n, out=vv[n:(k + 1) * n].view(k, n))       # blocks 1..k = X L
        C, sgn = coefficients(k)
        nep.dev.mlincomb(C, X, z)
        M0inv.solve_dev(z, out=vv[:n].reshape(1, n), scale=sgn)
        if compute_y0_method != "DEP":                                    # y0 -= Y T(c) = X (L T(c)[1:])
            w = dense.gemm_ts(X, (L[:k, :k] @ Tc[1:k + 1]).reshape(k, 1), k=k, rows=n, ldz=n)
            dense.axpy(-1.0, w, vv, n)
        h, beta, _ = dense.orthogonalize_and_normalize(V, vv, k, rows=n * (k + 1), ldv=ldv, active_rows=active,
                                                       method=orthmethod)
        H[:k, k - 1] = h; H[k, k - 1] = beta
        if ((k % check_error_every == 0) or (k == m)) and k > 2:
            Dv, Z = _hosteig.eig(H[:k, :k].copy())
            QT = dense.gemm_ts(V, Z, rowmajor=True, k=k, rows=n, ldz=ldv)
            lam = sigma + gamma / Dv
            e = estimate_errors(errmeasure, lam, QT)
            conv_eig = int(np.sum(e < tol))
            idx = np.argsort(e, kind="stable")
            err[k - 1, :k] = e[idx]
            if errhist is not None:
                errhist.append(err[k - 1, :k].copy())
            if k == m or conv_eig >= neigs:
                nrof = int(min(len(lam), neigs))
                lam = lam[idx[:nrof]]
                idx = idx[:nrof]
        k += 1
    k -= 1
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
        return lam, Qd, V[:k]
    return lam, to_host(Qd), V[:k]
