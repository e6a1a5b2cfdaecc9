"""Jacobi-Davidson (Betcke/Voss variant) on the device backend -- keyword surface of src/method_jd.jl:52-66.

Per iteration (method_jd.jl:122-168): the projected NEP W^H M(lam) V gains one row and column
(`expand_projectmatrices`, K1 + nep_gemv_h), the inner solver returns its eigenpairs, u = V s (K7), the error measure
(K2), then the expansion v = M(lam)^{-1} M'(lam) u with a NEW host factorisation of M(lam) (K1 + K5), orthogonalised
against V (K6); with the Petrov-Galerkin projection the test space gains w = M(lam) u (K1 + K6).
jd_effenberger (deflation based, :216-438) is not built.
"""
import numpy as np
import torch

from . import dense
from .errmeasure import DefaultErrmeasure, estimate_error
from .exceptions import NoConvergenceException
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, to_dev, to_host
from .projection import DefaultInnerSolver, create_proj_NEP, inner_solve

EPS = np.finfo(float).eps


def jd_eig_sorter(lamv, V, N, target):
    """method_jd.jl:177-183: the N-th closest Ritz value to the target (N = converged + 1)"""
    lamv = np.asarray(lamv, dtype=np.complex128).reshape(-1)
    NN = min(N, len(lamv))
    c = np.argsort(np.abs(lamv - target), kind="stable")
    return lamv[c[NN - 1]], np.asarray(V)[:, c[NN - 1]].astype(np.complex128)


def jd_betcke(nep, maxit=100, neigs=1, projtype="PetrovGalerkin", inner_solver_method=None, orthmethod=dense.DGKS,
              errmeasure=None, linsolvercreator=None, tol=EPS * 100, lam=0.0, v=None, target=0.0, logger=0, inner_logger=0):
    n = nep.size(1)
    if maxit > n:
        raise ValueError("maxit = %d is larger than size of NEP = %d." % (maxit, n))
    if projtype not in ("Galerkin", "PetrovGalerkin"):
        raise ValueError("Only accepted values of 'projtype' are :Galerkin and :PetrovGalerkin.")
    if inner_solver_method is None:
        inner_solver_method = DefaultInnerSolver()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if v is None:
        v = np.random.randn(n)
    lam = complex(lam); target = complex(target)
    lam_vec = np.zeros(neigs, dtype=np.complex128)
    u_vec = np.zeros((n, neigs), dtype=np.complex128)
    v0 = np.asarray(v, dtype=np.complex128)
    u = to_dev(v0 / np.linalg.norm(v0))[0].clone()
    conveig = 0
    err = estimate_error(errmeasure, lam, u)
    if err < tol:
        lam_vec[conveig] = lam; u_vec[:, conveig] = to_host(u.reshape(1, n))[:, 0]; conveig += 1
    if conveig == neigs:
        return lam_vec, u_vec
    proj_nep = create_proj_NEP(nep, maxit + 1)
    Vm = torch.zeros((maxit + 1, n), dtype=CDT, device="cuda")
    dense.copy(u, Vm[0], n)
    pg = projtype == "PetrovGalerkin"
    if pg:
        Wm = torch.zeros((maxit + 1, n), dtype=CDT, device="cuda")
        w0 = nep.compute_Mlincomb(lam, u.reshape(1, n))
        dense.copy(w0, Wm[0], n); dense.scal(Wm[0], 1.0 / dense.nrm2(Wm[0]), n)
    else:
        Wm = Vm
    one = np.ones(1)
    for k in range(1, maxit + 1):
        V = Vm[:k]; W = Wm[:k]
        proj_nep.expand_projectmatrices(W, V)
        lamv, sv = inner_solve(inner_solver_method, proj_nep, lamv=lam * np.ones(conveig + 1, dtype=complex), sigma=target,
                               neigs=conveig + 1)
        if len(np.atleast_1d(lamv)) == 0:
            raise NoConvergenceException(lam_vec[:conveig], u_vec[:, :conveig], err, "the inner solver returned no eigenpair")
        lam, s = jd_eig_sorter(lamv, np.asarray(sv).reshape(k, -1), conveig + 1, target)
        s = s / np.linalg.norm(s)
        u = dense.gemm_ts(V, s.reshape(k, 1), k=k, rows=n, ldz=n)[0]
        err = estimate_error(errmeasure, lam, u)
        if err < tol and (conveig == 0 or np.all(np.abs(lam - lam_vec[:conveig]) / np.abs(lam_vec[:conveig]) > np.sqrt(np.sqrt(EPS)))):
            lam_vec[conveig] = lam; u_vec[:, conveig] = to_host(u.reshape(1, n))[:, 0]; conveig += 1
        if conveig == neigs:
            return lam_vec, u_vec
        pk = nep.compute_Mlincomb(lam, u.reshape(1, n), one, 1)                   # M'(lam) u
        linsolver = create_linsolver(linsolvercreator, nep, lam)
        vnew = Vm[k]
        linsolver.solve_dev(pk, out=vnew.reshape(1, n))
        dense.orthogonalize_and_normalize(Vm, vnew, k, rows=n, ldv=n, method=orthmethod)
        if pg:
            wnew = Wm[k]
            dense.copy(nep.compute_Mlincomb(lam, u.reshape(1, n)), wnew, n)
            dense.orthogonalize_and_normalize(Wm, wnew, k, rows=n, ldv=n, method=orthmethod)
    msg = "Number of iterations exceeded. maxit=%d and only %d eigenvalues converged out of %d." % (maxit, conveig, neigs)
    raise NoConvergenceException(np.concatenate([lam_vec[:conveig], [lam]]),
                                 np.column_stack([u_vec[:, :conveig], to_host(u.reshape(1, n))[:, 0]]), err, msg)
