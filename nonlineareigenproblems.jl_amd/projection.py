"""Projected NEPs and inner solvers on the device backend.

Mirrors `create_proj_NEP`, `Proj_SPMF_NEP`, `set_projectmatrices!`, `expand_projectmatrices!`
(src/NEPTypes.jl:600-800) and `inner_solve` with `IARInnerSolver` / `NewtonInnerSolver` / `DefaultInnerSolver`
(src/inner_solver.jl:9-350): N(lam) = W^H M(lam) V = sum_i f_i(lam) B_i with B_i = W^H A_i V.

Device work of `set_projectmatrices`: V and W are brought into row-major form once (K7 with B = I), then per term i
one SpMM Y_i = A_i V over the stacked CSR (`nep_resid_block` with the coefficient block e_i 1^T, the K2 kernel) and one
reduction GEMM B_i = W^H Y_i on the FP64 matrix cores (K9 `nep_gemm_h_rm`: row-major operands are the MFMA operand
layout with the row index as contraction index).  `expand_projectmatrices` adds one row and one column with folded
SpMVs (K1, k = 1) and `nep_gemv_h` (the K6 dots kernel).  The k x k matrices B_i come back to the host, where the
projected problem is an ordinary (dense) SPMF_NEP of this backend.
"""
import numpy as np
import torch

from . import dense
from .exceptions import NoConvergenceException
from .nep import AbstractSPMF, CDT, DEP, PEP, SPMF_NEP, is_dev, to_dev

EPS = np.finfo(float).eps


def _as_dev_cols(X):
    """host n x k matrix or device (k, n) tensor -> device (k, n) tensor (column-major n x k block)"""
    if is_dev(X):
        return X if X.dim() == 2 else X.reshape(1, -1)
    X = np.asarray(X, dtype=np.complex128)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    if X.shape[1] == 0:
        return torch.empty((0, X.shape[0]), dtype=CDT, device="cuda")
    return to_dev(X)


class Proj_SPMF_NEP:
    """src/NEPTypes.jl:647-800.  Every NEP method is delegated to `nep_proj`, the small SPMF_NEP with the matrices
    B_i = W^H A_i V and the functions of the original problem."""

    def __init__(self, orgnep, maxsize=None):
        if not isinstance(orgnep, AbstractSPMF):
            raise TypeError("create_proj_NEP needs an AbstractSPMF")
        if type(orgnep).compute_Mlincomb is not AbstractSPMF.compute_Mlincomb:
            raise NotImplementedError("the operator has terms outside its SPMF matrices (e.g. the waveguide corner block)")
        self.orgnep = orgnep
        self.orgnep_Av = orgnep.get_Av()
        self.orgnep_fv = orgnep.get_fv()
        n = orgnep.size(1)
        if maxsize is None:
            maxsize = min(n, 201)
        self.maxsize = int(maxsize)
        self.projnep_B_mem = [np.zeros((self.maxsize, self.maxsize), dtype=np.complex128) for _ in self.orgnep_fv]
        self.k = 0
        self.nep_proj = None
        self._z = torch.empty(n, dtype=CDT, device="cuda")

    # ---- one column of A_i V on the device
    def _term_times(self, i, v):
        mt = len(self.orgnep_fv)
        C = np.zeros((1, mt), dtype=np.complex128, order="F")
        C[0, i] = 1.0
        self.orgnep.dev.mlincomb(C, v.reshape(1, -1), self._z, k=1, ldv=v.numel())
        return self._z

    def _rebuild(self, k):
        self.k = k
        self.nep_proj = SPMF_NEP([B[:k, :k].copy() for B in self.projnep_B_mem], self.orgnep_fv) if k > 0 else None

    def set_projectmatrices(self, W, V):
        """B_i[0:k,0:k] = W^H A_i V  (src/NEPTypes.jl:724-741)"""
        Wd, Vd = _as_dev_cols(W), _as_dev_cols(V)
        k = Vd.shape[0]
        if k > self.maxsize:
            raise ValueError("projection larger than the preallocated size (maxsize=%d)" % self.maxsize)
        n = self.orgnep.size(1)
        if k == 0:
            self._rebuild(0)
            return
        from ._lib import lib, check, hptr, c_vp
        from .nep import stream_ptr
        mt = len(self.orgnep_fv)
        eye = np.eye(k, dtype=np.complex128)
        VT = dense.gemm_ts(Vd, eye, rowmajor=True, k=k, rows=n, ldz=Vd.shape[1])          # (n, k) row-major
        WT = VT if Wd.data_ptr() == Vd.data_ptr() else dense.gemm_ts(Wd, eye, rowmajor=True, k=k, rows=n, ldz=Wd.shape[1])
        YT = torch.empty((n, k), dtype=CDT, device="cuda")
        for i in range(mt):
            F = np.zeros((mt, k), dtype=np.complex128, order="F")
            F[i, :] = 1.0
            check(lib.nep_resid_block(self.orgnep.dev.h, k, hptr(F), c_vp(VT.data_ptr()), k, c_vp(YT.data_ptr()), k,
                                      stream_ptr()))                                       # Y_i = A_i V
            self.projnep_B_mem[i][:k, :k] = dense.gemm_h_rm(WT, YT, n, k, k)                # B_i = W^H Y_i
        self._rebuild(k)

    def expand_projectmatrices(self, Wnew, Vnew):
        """adds the last column of Wnew / Vnew to the bases (src/NEPTypes.jl:770-790)"""
        Wd, Vd = _as_dev_cols(Wnew), _as_dev_cols(Vnew)
        k = Vd.shape[0] - 1
        if k + 1 > self.maxsize:
            raise ValueError("projection larger than the preallocated size (maxsize=%d)" % self.maxsize)
        if k + 1 > 4:
            # the new row needs A_i v_j for every old column again (k+1 SpMVs and as many synchronising dot calls per
            # term); from a handful of columns on, one batched pass (K2 SpMM + K9) over the whole basis is cheaper and
            # gives the same matrices
            return self.set_projectmatrices(Wd, Vd)
        n = self.orgnep.size(1)
        for i in range(len(self.orgnep_fv)):
            B = self.projnep_B_mem[i]
            y = self._term_times(i, Vd[k])                      # A_i v_new : new column
            B[:k + 1, k] = dense.gemv_h(Wd, y, k + 1, rows=n, ldv=Wd.shape[1])
            for j in range(k):                                  # w_new^H A_i v_j : new row
                y = self._term_times(i, Vd[j])
                B[k, j] = dense.gemv_h(Wd[k:k + 1], y, 1, rows=n, ldv=Wd.shape[1])[0]
        self._rebuild(k + 1)

    # ---- delegation (src/NEPTypes.jl:792-800)
    def size(self, d=None):
        return (self.k, self.k) if d is None else self.k

    def get_Av(self):
        return self.nep_proj.get_Av()

    def get_fv(self):
        return self.nep_proj.get_fv()

    def compute_Mlincomb(self, *a, **kw):
        return self.nep_proj.compute_Mlincomb(*a, **kw)

    def compute_Mder(self, *a, **kw):
        return self.nep_proj.compute_Mder(*a, **kw)

    def compute_MM(self, *a, **kw):
        return self.nep_proj.compute_MM(*a, **kw)


def create_proj_NEP(orgnep, maxsize=None):
    """src/NEPTypes.jl:636-640"""
    return Proj_SPMF_NEP(orgnep, maxsize)


# ----------------------------------------------------------------------------------------------
class InnerSolver:
    pass


class DefaultInnerSolver(InnerSolver):
    """src/inner_solver.jl:50-58,243-256"""


class NewtonInnerSolver(InnerSolver):
    """src/inner_solver.jl:83-94"""

    def __init__(self, tol=1e-13, maxit=80, starting_vector="Vk", newton_function=None):
        self.tol, self.maxit, self.starting_vector, self.newton_function = tol, maxit, starting_vector, newton_function


class IARInnerSolver(InnerSolver):
    """src/inner_solver.jl:133-144 (iar_function: iar or tiar of this backend)"""

    def __init__(self, tol=1e-13, maxit=80, starting_vector="ones", normalize_DEPs=False, iar_function=None):
        self.tol, self.maxit, self.starting_vector = tol, maxit, starting_vector
        self.normalize_DEPs, self.iar_function = bool(normalize_DEPs), iar_function


def IARChebInnerSolver(tol=1e-13, maxit=80, starting_vector="ones", normalize_DEPs=True):
    """src/inner_solver.jl:158-163"""
    from .iar_chebyshev import iar_chebyshev
    return IARInnerSolver(tol=tol, maxit=maxit, starting_vector=starting_vector, normalize_DEPs=normalize_DEPs,
                          iar_function=iar_chebyshev)


class PolyeigInnerSolver(InnerSolver):
    """src/inner_solver.jl:104,298-304: companion linearisation of the projected PEP (host LAPACK, k*d x k*d)"""


def polyeig(Bv):
    """eigenpairs of sum_i lam^i B_i through the first companion form (src/method_companion.jl); returns (lam, V) with
    the leading-block eigenvectors normalised"""
    import scipy.linalg as sla
    d = len(Bv) - 1
    k = Bv[0].shape[0]
    if d == 1:
        lam, X = sla.eig(-np.asarray(Bv[0]), np.asarray(Bv[1]))
        return lam, X
    A = np.zeros((k * d, k * d), dtype=complex); E = np.eye(k * d, dtype=complex)
    A[:k * (d - 1), k:] = np.eye(k * (d - 1))
    for i in range(d):
        A[k * (d - 1):, k * i:k * (i + 1)] = -np.asarray(Bv[i])
    E[k * (d - 1):, k * (d - 1):] = np.asarray(Bv[d])
    lam, X = sla.eig(A, E)
    X = X[:k, :]
    nrm = np.linalg.norm(X, axis=0)
    X = X / np.where(nrm > 0, nrm, 1.0)[None, :]
    return lam, X


def inner_solve(solver, pnep, lamv=None, V=None, neigs=10, sigma=0.0, tol=None, **kwargs):
    """src/inner_solver.jl:243-350.  Returns (lambdas, eigenvector matrix of the PROJECTED problem, k x #lambdas)."""
    from .iar import iar as _iar
    from .newton import augnewton as _augnewton
    k = pnep.size(1)
    if isinstance(solver, DefaultInnerSolver):
        org = pnep.orgnep
        if isinstance(org, PEP):                                 # inner_solver.jl:244-245
            solver = PolyeigInnerSolver()
        elif isinstance(org, DEP):                               # :246-248
            solver = IARChebInnerSolver()
        else:
            solver = IARInnerSolver() if isinstance(org, SPMF_NEP) else NewtonInnerSolver()
    if isinstance(solver, PolyeigInnerSolver):
        if not isinstance(pnep.orgnep, PEP):
            raise TypeError("Wrong type. PolyeigInnerSolver only handles the PEP type.")
        return polyeig(pnep.get_Av())
    if isinstance(solver, IARInnerSolver):
        nep = pnep.nep_proj
        if isinstance(pnep.orgnep, DEP) and solver.normalize_DEPs:          # :312-324
            AA = pnep.get_Av()
            nep = DEP([np.linalg.solve(AA[0], AA[1 + i]) for i in range(len(AA) - 1)], pnep.orgnep.tauv)
        v0 = np.ones(k) if solver.starting_vector == "ones" else np.random.randn(k)
        fn = solver.iar_function or _iar
        try:
            out = fn(nep, sigma=sigma, neigs=neigs, tol=solver.tol, maxit=solver.maxit, v=v0)
            return out[0], out[1]
        except NoConvergenceException as e:                                  # :339-346: keep what was found
            lam = np.asarray(e.lam, dtype=complex).reshape(-1)
            Q = np.zeros((k, 0), dtype=complex) if e.v is None else np.asarray(e.v).reshape(k, -1)
            return lam, Q
    if isinstance(solver, NewtonInnerSolver):
        from .errmeasure import ResidualErrmeasure
        lamv = np.array([0.0 + 0j]) if lamv is None else np.array(lamv, dtype=complex)
        Vm = np.random.rand(k, len(lamv)).astype(complex) if V is None else np.array(V, dtype=complex)
        fn = solver.newton_function or _augnewton
        errm = ResidualErrmeasure(pnep.nep_proj)
        for j in range(len(lamv)):
            if solver.starting_vector == "ones":
                v0 = np.ones(k) + 0j
            elif solver.starting_vector == "randn":
                v0 = np.random.randn(k) + 0j
            else:
                v0 = Vm[:, j].copy()
            try:
                l1, vp = fn(pnep.nep_proj, lam=lamv[j], v=v0, maxit=solver.maxit, tol=solver.tol, errmeasure=errm)
            except NoConvergenceException as e:
                l1, vp = e.lam, e.v
            Vm[:, j] = vp; lamv[j] = l1
        return lamv, Vm
    raise TypeError("unknown inner solver %r" % (solver,))
