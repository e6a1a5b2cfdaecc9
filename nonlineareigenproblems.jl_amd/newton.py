"""resinv, quasinewton, compute_rf (ScalarNewtonInnerSolver), armijo_rule on the device backend.

Mirrors src/method_newton.jl:142-226 (resinv), :380-445 (quasinewton), :598-609 (armijo_rule) and
src/compute_rf_wrapper.jl:25-54 (compute_rf).  The eigenvector iterate lives on the device; per
iteration only scalars (lambda, error, two dot products) cross PCIe.
"""

import numpy as np
import torch

from . import dense
from ._lib import lib, check, hptr, c_vp
from .errmeasure import DefaultErrmeasure, estimate_error
from .exceptions import NoConvergenceException
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, to_dev, to_host, stream_ptr

EPS = np.finfo(float).eps


def _dots2(y, Z2, n):
    """[y^H Z2[:,0], y^H Z2[:,1]] with one launch (X = y for both columns: ldx = 0)"""
    out = np.empty(2, dtype=np.complex128)
    check(lib.nep_coldots(n, 2, c_vp(y.data_ptr()), 0, c_vp(Z2.data_ptr()), n, hptr(out), stream_ptr()))
    return out


class ScalarNewtonInnerSolver:
    """src/compute_rf_wrapper.jl:18-24"""

    def __init__(self, tol=EPS * 100, maxit=80, bad_solution_allowed=True):
        self.tol, self.maxit, self.bad_solution_allowed = tol, maxit, bad_solution_allowed


def _mder_times(nep, lam, vb, out, der):
    """out = M^(der)(lam) v on the device: the stacked-CSR kernel K1 directly for pure SPMF operators, the NEP's own
    compute_Mlincomb for types with extra terms (the dense corner of the waveguide problem)"""
    from .nep import AbstractSPMF
    one = np.ones(1)
    if type(nep).compute_Mlincomb is AbstractSPMF.compute_Mlincomb:
        nep.dev.mlincomb(nep.coeff_block(lam, one, der), vb, out)
    else:
        dense.copy(nep.compute_Mlincomb(lam, vb, a=one, startder=der), out, vb.shape[-1])
    return out


def compute_rf(nep, x, inner_solver=None, y=None, target=0.0, lam=None):
    """Rayleigh functional by scalar Newton: y^H M(lam) x = 0  (compute_rf_wrapper.jl:25-54).
    x, y: device vectors (n,) or NumPy vectors.  Returns a length-1 complex array."""
    if inner_solver is None:
        inner_solver = ScalarNewtonInnerSolver()
    n = nep.size(1)
    xd = x if torch.is_tensor(x) else to_dev(x)[0]
    yd = xd if y is None else (y if torch.is_tensor(y) else to_dev(y)[0])
    lam_iter = complex(target if lam is None else lam)
    dlam = np.inf; count = 0
    Z2 = torch.empty((2, n), dtype=CDT, device="cuda")
    xb = xd.reshape(1, n)
    one = np.ones(1)
    while abs(dlam) > inner_solver.tol and count < inner_solver.maxit:
        count += 1
        _mder_times(nep, lam_iter, xb, Z2[0], 0)
        _mder_times(nep, lam_iter, xb, Z2[1], 1)
        d = _dots2(yd, Z2, n)
        dlam = -d[0] / d[1]
        lam_iter += dlam
    if count == inner_solver.maxit and not inner_solver.bad_solution_allowed:
        raise NoConvergenceException(lam_iter, None, None, "compute_rf: scalar Newton did not converge")
    return np.array([lam_iter])


def _err_at(nep, errmeasure, lam, v, dv, work):
    """estimate_error(errmeasure, lam, v + dv) without modifying v"""
    dense.copy(v, work); dense.axpy(1.0, dv, work)
    return estimate_error(errmeasure, lam, work)


def armijo_rule(nep, errmeasure, err0, lam, v, dlam, dv, armijo_factor, armijo_max):
    """method_newton.jl:598-609; dv is scaled in place on the device"""
    j = 0
    if armijo_factor < 1:
        work = torch.empty_like(v)
        while _err_at(nep, errmeasure, lam + dlam, v, dv, work) > err0 and j < armijo_max:
            j += 1
            dense.scal(dv, armijo_factor)
            dlam = dlam * armijo_factor
    return dlam, dv, j, armijo_factor ** j


def resinv(nep, errmeasure=None, tol=EPS * 100, maxit=100, lam=0.0, v=None, c=None, logger=0,
           inner_solver=None, linsolvercreator=None, armijo_factor=1, armijo_max=5, hist=None):
    """Residual inverse iteration (method_newton.jl:142-226)."""
    n = nep.size(1)
    lam = complex(lam)
    if v is None:
        v = np.random.randn(n)
    vd = to_dev(np.asarray(v, dtype=np.complex128))[0]
    cd = vd.clone() if c is None else to_dev(np.asarray(c, dtype=np.complex128))[0]
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    linsolver = create_linsolver(linsolvercreator, nep, lam)
    use_v_as_rf_vector = dense.nrm2(cd) == 0
    sigma = lam
    err = np.inf
    one = np.ones(1)
    z = torch.empty(n, dtype=CDT, device="cuda")
    dv = torch.empty(n, dtype=CDT, device="cuda")
    for k in range(1, maxit + 1):
        dense.scal(vd, 1.0 / dense.nrm2(vd))
        err = estimate_error(errmeasure, lam, vd)
        if use_v_as_rf_vector:
            dense.copy(vd, cd)
        if hist is not None:
            hist.append((k, err, lam))
        if err < tol:
            return lam, to_host(vd.reshape(1, n))[:, 0]
        lam_vec = compute_rf(nep, vd, inner_solver, y=cd, lam=lam, target=sigma)
        lam1 = lam_vec[np.argmin(abs(lam_vec - lam))]
        dlam = lam1 - lam
        _mder_times(nep, lam1, vd.reshape(1, n), z, 0)
        linsolver.solve_dev(z, out=dv.reshape(1, n), scale=-1.0)
        dlam, dv, j, scaling = armijo_rule(nep, errmeasure, err, lam, vd, dlam, dv, float(armijo_factor), armijo_max)
        lam += dlam
        dense.axpy(1.0, dv, vd)
    raise NoConvergenceException(lam, to_host(vd.reshape(1, n))[:, 0], err,
                                 "Number of iterations exceeded. maxit=%d." % maxit)


def quasinewton(nep, errmeasure=None, tol=EPS * 100, maxit=100, lam=0.0, v=None, ws=None, logger=0,
                linsolvercreator=None, armijo_factor=1, armijo_max=5, hist=None):
    """Quasi-Newton with a fixed factorisation (method_newton.jl:380-445)."""
    n = nep.size(1)
    lam = complex(lam)
    if v is None:
        v = np.random.randn(n)
    vd = to_dev(np.asarray(v, dtype=np.complex128))[0]
    wsd = vd.clone() if ws is None else to_dev(np.asarray(ws, dtype=np.complex128))[0]
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    linsolver = create_linsolver(linsolvercreator, nep, lam)
    err = np.inf
    one = np.ones(1)
    UW = torch.empty((2, n), dtype=CDT, device="cuda")     # u = M(lam) v, w = M'(lam) v
    dv = torch.empty(n, dtype=CDT, device="cuda")
    for k in range(1, maxit + 1):
        err = estimate_error(errmeasure, lam, vd)
        if hist is not None:
            hist.append((k, err, lam))
        if err < tol:
            return lam, to_host(vd.reshape(1, n))[:, 0]
        vb = vd.reshape(1, n)
        _mder_times(nep, lam, vb, UW[0], 0)
        _mder_times(nep, lam, vb, UW[1], 1)
        d = _dots2(wsd, UW, n)
        dlam = -d[0] / d[1]
        # z = dlam*w + u  (in place in u)
        dense.axpy(dlam, UW[1], UW[0], n)
        linsolver.solve_dev(UW[0], out=dv.reshape(1, n), scale=-1.0)
        dlam, dv, j, scaling = armijo_rule(nep, errmeasure, err, lam, vd, dlam, dv, float(armijo_factor), armijo_max)
        lam += dlam
        dense.axpy(1.0, dv, vd)
    raise NoConvergenceException(lam, to_host(vd.reshape(1, n))[:, 0], err,
                                 "Number of iterations exceeded. maxit=%d." % maxit)


def augnewton(nep, errmeasure=None, tol=EPS * 100, maxit=30, lam=0.0, v=None, c=None, logger=0, linsolvercreator=None,
              armijo_factor=1, armijo_max=5):
    """Augmented Newton (method_newton.jl:262-345): a NEW factorisation of M(lam_k) every step (BackslashLinSolver
    semantics through create_linsolver), vectors of length n only.  c = 0 selects v/||v||^2 as normalisation vector."""
    n = nep.size(1)
    lam = complex(lam)
    if v is None:
        v = np.random.randn(n)
    vh = np.asarray(v, dtype=np.complex128).copy()
    ch = vh.copy() if c is None else np.asarray(c, dtype=np.complex128).copy()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    use_v = False
    if np.linalg.norm(ch) == 0:
        use_v = True
        ch = vh / np.linalg.norm(vh) ** 2
    vh = vh / np.vdot(ch, vh)
    vd = to_dev(vh)[0]
    cd = to_dev(ch)[0]
    one = np.ones(1)
    z = torch.empty(n, dtype=CDT, device="cuda")
    tv = torch.empty(n, dtype=CDT, device="cuda")
    err = np.inf
    for k in range(1, maxit + 1):
        err = estimate_error(errmeasure, lam, vd)
        if err < tol:
            return lam, to_host(vd.reshape(1, n))[:, 0]
        _mder_times(nep, lam, vd.reshape(1, n), z, 1)          # z = M'(lam) v
        linsolver = create_linsolver(linsolvercreator, nep, lam)
        linsolver.solve_dev(z, out=tv.reshape(1, n))
        if use_v:
            nv = dense.nrm2(vd)
            dense.copy(vd, cd, n); dense.scal(cd, 1.0 / nv ** 2, n)
        alpha = 1.0 / _dots2(cd, torch.stack([tv, tv]), n)[0]
        dlam = -alpha
        dv = tv                                                                        # dv = alpha*tempvec - v
        dense.scal(dv, alpha, n); dense.axpy(-1.0, vd, dv, n)
        dlam, dv, j, scaling = armijo_rule(nep, errmeasure, err, lam, vd, dlam, dv, float(armijo_factor), armijo_max)
        lam += dlam
        dense.axpy(1.0, dv, vd)
    raise NoConvergenceException(lam, to_host(vd.reshape(1, n))[:, 0], err,
                                 "Number of iterations exceeded. maxit=%d." % maxit)
