"""Nonlinear Arnoldi (Voss) on the device backend -- keyword surface of src/method_nlar.jl:30-58.

Per iteration (method_nlar.jl:94-160): the projected NEP gains one row and column (`expand_projectmatrices`: folded SpMVs
K1 + `nep_gemv_h`), the inner solver returns Ritz pairs of the small problem, the sorter ranks them -- the residual
sorter evaluates ALL lifted Ritz vectors with one K7 GEMM + one K2 residual pass instead of a loop of single
residuals -- then u = V_k y (K7), r = M(nu) u (K1), the new direction M(sigma)^{-1} r (K5) is orthogonalised against
the basis (K6).  Restarts rebuild the projected matrices from scratch (K2 + K9); the reference keeps the stale leading
block of its projected matrices after a restart (`expand_projectmatrices!` only touches the last row and column), which
is not reproduced.
"""
import numpy as np
import torch

from . import dense
from .errmeasure import DefaultErrmeasure, estimate_error, estimate_errors
from .exceptions import NoConvergenceException
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, to_dev, to_host
from .projection import DefaultInnerSolver, create_proj_NEP, inner_solve

EPS = np.finfo(float).eps


def discard_ritz_values(dd, D, R):
    """method_nlar.jl:166-174: Ritz values within R of a converged eigenvalue are set to Inf"""
    dd = np.array(dd, dtype=np.complex128)
    for j in range(len(D)):
        dd[np.abs(dd - D[j]) < R] = np.inf
    return dd


def default_eigval_sorter(nep, dd, vv, sigma, D, R, Vk, errmeasure=None):
    """method_nlar.jl:176-183"""
    dd2 = discard_ritz_values(dd, D, R)
    ii = np.argsort(np.abs(dd2 - sigma), kind="stable")
    return dd2[ii], vv[:, ii]


def residual_eigval_sorter(nep, dd, vv, sigma, D, R, Vk, errmeasure=None):
    """method_nlar.jl:185-196; Vk: device (cbs, n) basis block"""
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    dd = np.asarray(dd, dtype=np.complex128)
    dd2 = discard_ritz_values(dd, D, R)
    cbs, n = Vk.shape
    QT = dense.gemm_ts(Vk, vv, rowmajor=True, k=cbs, rows=n, ldz=n)
    eig_res = estimate_errors(errmeasure, dd, QT)
    with np.errstate(all="ignore"):
        key = eig_res * np.abs(dd2 - sigma)
    key = np.where(np.isnan(key), np.inf, key)
    ii = np.argsort(key, kind="stable")
    return dd[ii], vv[:, ii]


def nlar(nep, orthmethod=dense.MGS, neigs=10, errmeasure=None, tol=EPS * 100, maxit=100, lam=0.0, v=None, logger=0,
         linsolvercreator=None, R=0.01, eigval_sorter=residual_eigval_sorter, qrfact_orth=False, max_subspace=100,
         num_restart_ritz_vecs=8, inner_solver_method=None, inner_logger=0):
    n = nep.size(1)
    if maxit > n:
        maxit = n
    if num_restart_ritz_vecs > neigs:
        num_restart_ritz_vecs = neigs
    if max_subspace < num_restart_ritz_vecs:
        max_subspace = num_restart_ritz_vecs + 20
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if inner_solver_method is None:
        inner_solver_method = DefaultInnerSolver()
    if v is None:
        v = np.random.randn(n)
    sigma = complex(lam)
    nu = sigma
    V = torch.zeros((max_subspace + 1, n), dtype=CDT, device="cuda")
    X = np.zeros((n, neigs), dtype=np.complex128)
    v0 = np.asarray(v, dtype=np.complex128)
    V[0] = to_dev(v0 / np.linalg.norm(v0))[0]
    cbs = 1
    D = np.zeros(neigs, dtype=np.complex128)
    err_hist = EPS * np.ones((maxit, neigs))
    m = 0; k = 1
    proj_nep = create_proj_NEP(nep, max(maxit, max_subspace) + 1)
    linsolver = create_linsolver(linsolvercreator, nep, sigma)
    err = np.inf
    u = None
    fresh = False                     # projected matrices have to be rebuilt (after a restart)
    while m < neigs and k < maxit:
        Vk = V[:cbs]
        if fresh:
            proj_nep.set_projectmatrices(Vk, Vk)
            fresh = False
        else:
            proj_nep.expand_projectmatrices(Vk, Vk)
        dd, vv = inner_solve(inner_solver_method, proj_nep, neigs=neigs, sigma=sigma)
        dd = np.asarray(dd, dtype=np.complex128).reshape(-1); vv = np.asarray(vv, dtype=np.complex128).reshape(cbs, -1)
        if len(dd) == 0:
            raise RuntimeError("We did not find any (non-converged) eigenvalues to target")
        nuv, yv = eigval_sorter(nep, dd, vv, sigma, D[:m], R, Vk)
        nu = nuv[0]
        if np.isinf(nu):
            raise RuntimeError("We did not find any (non-converged) eigenvalues to target")

        def lift(y):
            ud = dense.gemm_ts(Vk, y.reshape(cbs, 1), k=cbs, rows=n, ldz=n)[0]
            dense.scal(ud, 1.0 / dense.nrm2(ud), n)
            return ud
        u = lift(yv[:, 0])
        res = nep.compute_Mlincomb(nu, u.reshape(1, n))
        err = estimate_error(errmeasure, nu, u)
        err_hist[k - 1, m] = err
        if err < tol:
            D[m] = nu
            X[:, m] = to_host(u.reshape(1, n))[:, 0]
            m += 1
            if m >= neigs:
                break                 # done; the reference still expands once more (with a NaN direction if no Ritz value is left)
            nuv, yv = eigval_sorter(nep, dd, vv, sigma, D[:m], R, Vk)
            if np.isinf(nuv[0]):
                raise RuntimeError("We did not find any (non-converged) eigenvalues to target")
            u1 = lift(yv[:, 0])
            res = nep.compute_Mlincomb(nuv[0], u1.reshape(1, n))
        if cbs >= max_subspace:
            # restart (method_nlar.jl:133-140): converged vectors + leading Ritz vectors, orthonormalised
            nr = min(num_restart_ritz_vecs, yv.shape[1])
            Zh = np.column_stack([X[:, :m], to_host(dense.gemm_ts(Vk, yv[:, :nr], k=cbs, rows=n, ldz=n))])
            Qh, _ = np.linalg.qr(Zh)
            cbs = Qh.shape[1]
            V[:cbs] = to_dev(Qh)
            fresh = True
        else:
            dv = linsolver.solve_dev(res.reshape(1, n) if res.dim() == 1 else res)
            dvv = dv.reshape(-1)
            if qrfact_orth:
                Zh = np.column_stack([to_host(Vk), to_host(dvv.reshape(1, n))[:, 0]])
                Qh, _ = np.linalg.qr(Zh)
                cbs += 1
                V[:cbs] = to_dev(Qh)
                fresh = True
            else:
                dense.copy(dvv, V[cbs], n)
                dense.orthogonalize_and_normalize(V, V[cbs], cbs, rows=n, ldv=n, method=orthmethod)
                cbs += 1
        k += 1
    if k >= maxit and m < neigs:
        msg = "Number of iterations exceeded. maxit=%d and only %d eigenvalues converged out of %d." % (maxit, m, neigs)
        raise NoConvergenceException(nu, None if u is None else to_host(u.reshape(1, n))[:, 0], err, msg)
    return D, X, err_hist
