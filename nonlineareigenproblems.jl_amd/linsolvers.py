"""Linear solvers for M(lam) x = b: host factorisation (one-off per shift), device triangular solves.

Mirrors src/LinSolvers.jl:100-159 and src/LinSolverCreators.jl:11-196:
`create_linsolver(creator, nep, lam)`, `lin_solve(solver, b; tol)`, `FactorizeLinSolver`,
`BackslashLinSolver`, `FactorizeLinSolverCreator(umfpack_refinements, max_factorizations, nep,
precomp_values)`, `BackslashLinSolverCreator`, `DefaultLinSolverCreator`; plus `LinSolverCache`
(src/rk_helper/linsolvercache.jl:7-26).

The reference factorises with UMFPACK on the host (`factorize(A)`, LinSolvers.jl:116).  Here the
host step is SuperLU (SciPy's bundled copy -- the only sparse LU in this image) and the factors
are uploaded once; every lin_solve is the HIP kernel k_lu_solve (csrc/trsv.hip).
"""
import ctypes as C
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

from . import _lib
from ._lib import lib, check, hptr, c_vp, c_i64
from .nep import CDT, to_dev, to_host, is_dev, stream_ptr


class LinSolver:
    pass


class DeviceLU:
    """nep_lu handle built from a host sparse LU of A (CSC/CSR/dense)."""

    def __init__(self, A, permc_spec="COLAMD", diag_pivot_thresh=None):
        _lib.require_gpu()
        t0 = time.perf_counter()
        n = A.shape[0]
        self.n = n
        Ac = sp.csc_matrix(A, dtype=np.complex128)
        opts = {}
        kw = dict(permc_spec=permc_spec)
        if diag_pivot_thresh is not None:
            kw["diag_pivot_thresh"] = diag_pivot_thresh
        try:
            lu = spla.splu(Ac, **kw)
        except RuntimeError as e:  # "Factor is exactly singular"
            raise np.linalg.LinAlgError("SingularException: " + str(e))
        self.t_factor = time.perf_counter() - t0
        L = sp.csr_matrix(lu.L); U = sp.csr_matrix(lu.U)
        L.sort_indices(); U.sort_indices()
        Lp = np.ascontiguousarray(L.indptr, dtype=np.int32); Li = np.ascontiguousarray(L.indices, dtype=np.int32)
        Lx = np.ascontiguousarray(L.data, dtype=np.complex128)
        Up = np.ascontiguousarray(U.indptr, dtype=np.int32); Ui = np.ascontiguousarray(U.indices, dtype=np.int32)
        Ux = np.ascontiguousarray(U.data, dtype=np.complex128)
        pr = np.ascontiguousarray(lu.perm_r, dtype=np.int32); pc = np.ascontiguousarray(lu.perm_c, dtype=np.int32)
        h = c_vp()
        check(lib.nep_lu_create(n, hptr(Lp), hptr(Li), hptr(Lx), hptr(Up), hptr(Ui), hptr(Ux), hptr(pr), hptr(pc),
                                C.byref(h)))
        self.h = h
        info = (c_i64 * 6)()
        check(lib.nep_lu_info(self.h, info))
        self.nnzL, self.nnzU, self.levL, self.levU, self.solve_bytes = (int(info[1]), int(info[2]), int(info[3]),
                                                                        int(info[4]), int(info[5]))
        self.t_setup = time.perf_counter() - t0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.nep_lu_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self, B, out=None, scale=1.0):
        """B: device tensor (nrhs, n) (column-major n x nrhs block). Returns same-shaped tensor."""
        Bd = B if B.dim() == 2 else B.reshape(1, -1)
        nrhs, n = Bd.shape
        assert n == self.n
        X = torch.empty_like(Bd) if out is None else out
        check(lib.nep_lu_solve(self.h, nrhs, c_vp(Bd.data_ptr()), n, c_vp(X.data_ptr()), n, float(scale), stream_ptr()))
        return X.reshape(B.shape)


class FactorizeLinSolver(LinSolver):
    """src/LinSolvers.jl:109-137: factor M(lam) once, solve many right-hand sides."""

    def __init__(self, nep, lam, umfpack_refinements=0, permc_spec="COLAMD", _lu=None):
        self.lam = lam
        self.umfpack_refinements = umfpack_refinements
        self.lu = _lu if _lu is not None else DeviceLU(nep.compute_Mder(lam), permc_spec=permc_spec)


class BackslashLinSolver(LinSolver):
    """src/LinSolvers.jl:147-159: `A \\ x`, i.e. a fresh factorisation at every lin_solve."""

    def __init__(self, nep, lam, permc_spec="COLAMD"):
        self.A = nep.compute_Mder(lam)
        self.permc_spec = permc_spec


def lin_solve(solver, b, tol=0, scale=1.0):
    """src/LinSolvers.jl:125-137.  b: NumPy vector/matrix (-> NumPy result) or device tensor
    (nrhs, n) / (n,) (-> device result).  `scale` multiplies the result on the device."""
    host = not is_dev(b)
    bd = to_dev(b) if host else b
    if isinstance(solver, BackslashLinSolver):
        lu = DeviceLU(solver.A, permc_spec=solver.permc_spec)
    else:
        lu = solver.lu
    x = lu.solve(bd, scale=scale)
    if host:
        xh = to_host(x if x.dim() == 2 else x.reshape(1, -1))
        return xh[:, 0] if np.ndim(b) == 1 else xh
    return x


class LinSolverCreator:
    pass


class FactorizeLinSolverCreator(LinSolverCreator):
    """src/LinSolverCreators.jl:62-122 (factorisation recycling keyed by lam)."""

    def __init__(self, umfpack_refinements=0, max_factorizations=0, nep=None, precomp_values=(),
                 permc_spec="COLAMD"):
        if np.isscalar(precomp_values):
            precomp_values = [precomp_values]
        if len(precomp_values) > 0 and nep is None:
            raise ValueError("When you want to precompute factorizations you need to supply the keyword "
                             "argument `nep`")
        self.umfpack_refinements = umfpack_refinements
        self.max_factorizations = max_factorizations
        self.permc_spec = permc_spec
        self.recycled_factorizations = {}
        for s in precomp_values:
            self.recycled_factorizations[complex(s)] = DeviceLU(nep.compute_Mder(s), permc_spec=permc_spec)


class BackslashLinSolverCreator(LinSolverCreator):
    def __init__(self, permc_spec="COLAMD"):
        self.permc_spec = permc_spec


DefaultLinSolverCreator = FactorizeLinSolverCreator


def create_linsolver(creator, nep, lam):
    """src/LinSolverCreators.jl:107-122,143."""
    if isinstance(creator, BackslashLinSolverCreator):
        return BackslashLinSolver(nep, lam, creator.permc_spec)
    key = complex(lam)
    if key in creator.recycled_factorizations:
        return FactorizeLinSolver(nep, lam, creator.umfpack_refinements, _lu=creator.recycled_factorizations[key])
    solver = FactorizeLinSolver(nep, lam, creator.umfpack_refinements, permc_spec=creator.permc_spec)
    if len(creator.recycled_factorizations) < creator.max_factorizations:
        creator.recycled_factorizations[key] = solver.lu
    return solver


class LinSolverCache:
    """src/rk_helper/linsolvercache.jl:7-26."""

    def __init__(self, nep, linsolvercreator):
        self.nep = nep
        self.linsolvercreator = linsolvercreator
        self.solvers = {}

    def solve(self, sigma, y, add_to_cache):
        key = complex(sigma)
        if key in self.solvers:
            return lin_solve(self.solvers[key], y)
        solver = create_linsolver(self.linsolvercreator, self.nep, sigma)
        if add_to_cache:
            self.solvers[key] = solver
        return lin_solve(solver, y)
