"""Waveguide-specific linear solvers on the device: the Schur-complement solvers and the Sylvester-SMW preconditioner of
src/gallery_extra/waveguide/Waveguide.jl:394-567 and waveguide_preconditioner.jl (Ringh, Mele, Karlsson, Jarlebring,
"Sylvester-based preconditioning for the waveguide eigenvalue problem").

  WEPLinSolverCreator(solver_type="factorized" | "backslash" | "gmres", kwargs=...)     Waveguide.jl:489-519
  lin_solve through the Schur complement of the boundary block (Ringh Prop. 2.1)        Waveguide.jl:552-567
  SchurMatVec, construct_WEP_schur_complement                                           Waveguide.jl:394-425, 523-550
  wep_generate_preconditioner(nep, N, sigma)                                            waveguide_preconditioner.jl:36-47

Device realisation.  The interior operator A(lam) X + X B + K .* X is the interior block of the three stacked sparse terms,
so SchurMatVec is ONE K1 call on the zero-padded vector (its last 2 nz rows are C2T v for free) + P(lam)^{-1} on 2 nz
entries + one rectangular CSR product with C1 (nep_csr_mv).  P(lam)^{-1} = R diag(1/s(lam)) R^H / nz with the dense
scaled-DFT matrix R (two nz x nz GEMVs).  The reference diagonalises the Sylvester operator with FFTs along z (length nz)
and sine transforms along x (FFT length 2 (nx + 1)); here both are dense transforms -- nz x nz DFT and nx x nx sine
matrices applied as GEMMs (nep_zgemm for the DFT, nep_dgemm on the re/im-interleaved block for the real sine matrix), 4 GEMMs
per Sylvester solve: at nz = 999 = 27 * 37 and 2 (nx + 1) = 2008 = 8 * 251 a 4-8 GFLOP GEMM on the FP64 matrix cores (0.1 ms)
costs less than a mixed-radix FFT would save.  The
region sums / expansions of the SMW correction are products with 0/1 indicator matrices (small GEMMs), the mm x mm SMW
matrix is inverted once on the host and applied as a GEMV, so one preconditioner application never synchronises.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib, dense
from ._lib import lib, check, c_vp
from .linsolvers import DeviceLU, FactorizeLinSolver, GMRESLinSolver, LinSolver, LinSolverCreator
from .nep import CDT, DeviceCSR, to_dev, to_host, is_dev, stream_ptr
from .wep import WEP, _corner_derivs

N_, T_, C_ = 0, 1, 2          # op codes of nep_zgemm


def _p(t):
    return c_vp(t.data_ptr() if is_dev(t) else int(t))


def zgemm(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, Cm, ldc):
    check(lib.nep_zgemm(ta, tb, m, n, k, _lib.cd(alpha), _p(A), lda, _p(B), ldb, _lib.cd(beta), _p(Cm), ldc, stream_ptr()))


def _boundary_csr(nep):
    """C1 (N x 2nz) and C2T (2nz x N) of the stacked matrix A1 = [Q0 C1; C2T 0] as device operators (cached on the NEP)"""
    if getattr(nep, "_C1dev", None) is None:
        N = nep.N
        A1 = sp.csr_matrix(nep.A[0])
        nep._C1dev = DeviceCSR(A1[:N, N:])
        nep._C2Tdev = DeviceCSR(A1[N:, :N])
    return nep._C1dev, nep._C2Tdev


class SchurOps:
    """device pieces shared by the Schur-complement solvers for one shift: P(lam)^{-1}, SchurMatVec, the elimination /
    back-substitution of the boundary unknowns"""

    def __init__(self, nep, lam):
        if not isinstance(nep, WEP):
            raise TypeError("WEPLinSolver can only be used in combination with WEPs: type(nep)=%s" % type(nep).__name__)
        self.nep, self.lam = nep, complex(lam)
        self.n, self.N, self.nz, self.nx = nep.n, nep.N, nep.nz, nep.nx
        self.C1, self.C2T = _boundary_csr(nep)
        self.Rm, self.RmH = nep._corner_dev()
        s = _corner_derivs(nep.wd, self.lam, 1)[:, 0]
        self.sinv = to_dev(1.0 / (s * self.nz))[0]                     # 1 / (nz s_j(lam)), 2 nz entries
        self.coef = to_dev(np.array([[1.0, self.lam, self.lam ** 2]], dtype=np.complex128))    # 1 x 3, device resident
        # matrix-free form of the interior operator (five-point stencil, csrc/wep.hip nep_wep_schur_matvec): weights of
        # generate_fd_interior_mat / generate_fd_boundary_mat (Waveguide.jl:17-19), diagonal K + lam^2 - 2/hz^2 - 2/hx^2
        wd = nep.wd
        self.stencil = None
        if nep._pinv_plan() is not None and os.environ.get("NEP_WEP_STENCIL", "1") != "0":
            lam_ = self.lam
            D0 = np.asarray(wd.K, dtype=np.complex128) + (lam_ ** 2 - 2.0 / wd.hz ** 2 - 2.0 / wd.hx ** 2)
            self.stencil = dict(D0=to_dev(D0), cp=_lib.cd(1.0 / wd.hz ** 2 + lam_ / wd.hz), cm=_lib.cd(1.0 / wd.hz ** 2 - lam_ / wd.hz),
                                cx=1.0 / wd.hx ** 2, d1=2.0 / wd.hx, d2=-1.0 / (2.0 * wd.hx), c1s=1.0 / wd.hx ** 2)
        self.pad = torch.zeros(self.n, dtype=CDT, device="cuda")
        self.z = torch.empty(self.n, dtype=CDT, device="cuda")
        self.t = torch.empty(2 * self.nz, dtype=CDT, device="cuda")
        self.u = torch.empty(2 * self.nz, dtype=CDT, device="cuda")
        self.w = torch.empty(2 * self.nz, dtype=CDT, device="cuda")

    def pinv(self, x, out):
        """out = blkdiag(R, R) diag(1/s(lam)) blkdiag(R, R)^H x / nz   (Waveguide.jl:159-162); x, out: device addresses of
        2 nz entries (minus block, plus block); two A^H x products per block (nep_gemv_hd), the diagonal scaling fused into
        the first one; the work vector self.w is used"""
        nz = self.nz
        xa = x.data_ptr() if is_dev(x) else x
        oa = out.data_ptr() if is_dev(out) else out
        st = stream_ptr()
        plan = self.nep._pinv_plan()
        if plan is not None:
            check(lib.nep_wep_pinv_apply(plan, c_vp(self.sinv.data_ptr()), c_vp(xa), c_vp(oa), st))
            return
        for half in (0, 1):
            o = 16 * half * nz
            check(lib.nep_gemv_hd(c_vp(self.Rm.data_ptr()), nz, nz, nz, c_vp(xa + o), c_vp(self.sinv.data_ptr() + o),
                                  c_vp(self.w.data_ptr() + o), st))                         # w = (1/(nz s)) .* (R^H x)
            check(lib.nep_gemv_hd(c_vp(self.RmH.data_ptr()), nz, nz, nz, c_vp(self.w.data_ptr() + o), None, c_vp(oa + o), st))   # (R^H)^H w

    def matvec(self, v, out):
        """out = vec(A(lam) X + X B + K .* X) - C1 P(lam)^{-1} C2T v   (SchurMatVec, Waveguide.jl:398-406)"""
        N, n = self.N, self.n
        if self.stencil is not None and v.data_ptr() != out.data_ptr():
            st = self.stencil
            check(lib.nep_wep_schur_matvec(self.nep._pinv_plan(), c_vp(self.sinv.data_ptr()), self.nx, _p(v), _p(st["D0"]), st["cp"], st["cm"],
                                           st["cx"], st["d1"], st["d2"], st["c1s"], c_vp(self.t.data_ptr()), _p(out), stream_ptr()))
            return out
        check(lib.nep_dev_copy(c_vp(self.pad.data_ptr()), _p(v), 16 * N, stream_ptr()))
        self.nep.dev.mlincomb_dev(self.coef, 1, 1, self.pad.data_ptr(), n, self.z)      # no host->device copy: graph-safe
        self.pinv(self.z.data_ptr() + 16 * N, self.t)
        self.C1.mv(-1.0, self.t, 1.0, self.z, out)
        return out

    def eliminate(self, x, rhs):
        """rhs = x_int - C1 P^{-1} x_ext   (Waveguide.jl:559-561)"""
        xa = x.data_ptr() if is_dev(x) else x
        self.pinv(xa + 16 * self.N, self.t)
        self.C1.mv(-1.0, self.t, 1.0, xa, rhs)
        return rhs

    def recover(self, q, x, out):
        """out = [q; P^{-1}(x_ext - C2T q)]   (Waveguide.jl:565)"""
        xa = x.data_ptr() if is_dev(x) else x
        oa = out.data_ptr() if is_dev(out) else out
        self.C2T.mv(-1.0, q, 1.0, xa + 16 * self.N, self.u)
        self.pinv(self.u, oa + 16 * self.N)
        if q.data_ptr() != oa:
            check(lib.nep_dev_copy(c_vp(oa), c_vp(q.data_ptr()), 16 * self.N, stream_ptr()))
        return out


def construct_WEP_schur_complement(nep, lam):
    """Waveguide.jl:523-550 (Ringh Prop. 3.1) on the host: interior block of M(lam) minus the two dense boundary couplings
    kron(E, P_-^{-1}) + kron(EE, P_+^{-1}) (four dense nz x nz blocks)"""
    wd = nep.wd; nz, nx, N = nep.nz, nep.nx, nep.N
    lam = complex(lam)
    S = sp.csr_matrix(nep.A[0], dtype=np.complex128)[:N, :N] + lam * sp.csr_matrix(nep.A[1])[:N, :N] + lam ** 2 * sp.csr_matrix(nep.A[2])[:N, :N]
    Rm = wd.Rmat()
    s = _corner_derivs(wd, lam, 1)[:, 0]
    Pm = (Rm / s[None, :nz]) @ Rm.conj().T / nz
    Pp = (Rm / s[None, nz:]) @ Rm.conj().T / nz
    d1 = (2 / wd.hx) / wd.hx ** 2; d2 = (-1 / (2 * wd.hx)) / wd.hx ** 2
    ii, jj = np.meshgrid(np.arange(nz), np.arange(nz), indexing="ij")
    rows = np.concatenate([ii.ravel(), ii.ravel(), (N - nz + ii).ravel(), (N - nz + ii).ravel()])
    cols = np.concatenate([jj.ravel(), (nz + jj).ravel(), (N - nz + jj).ravel(), (N - 2 * nz + jj).ravel()])
    vals = np.concatenate([d1 * Pm.ravel(), d2 * Pm.ravel(), d1 * Pp.ravel(), d2 * Pp.ravel()])
    return sp.csc_matrix(S - sp.csr_matrix((vals, (rows, cols)), shape=(N, N)))


class _SchurSolve:
    """`lu`-like object (n, solve, solve_add) whose solve eliminates the boundary unknowns, calls the inner solver on the
    Schur complement and recovers them -- what FactorizeLinSolver's refinement loop drives"""

    def __init__(self, ops, inner):
        self.ops, self.inner, self.n = ops, inner, ops.n
        import inspect
        self._sweep_kw = "sweep" in inspect.signature(inner).parameters      # the iterative inner solver is told which solves are sweeps
        self.rhs = torch.empty(ops.N, dtype=CDT, device="cuda")
        self.q = torch.empty(ops.N, dtype=CDT, device="cuda")
        self.tmp = torch.empty(ops.n, dtype=CDT, device="cuda")

    def _one(self, b, out, tol=None, sweep=False):
        self.ops.eliminate(b, self.rhs)
        if sweep and self._sweep_kw:
            self.inner(self.rhs, self.q, tol, sweep=True)
        else:
            self.inner(self.rhs, self.q, tol)
        self.ops.recover(self.q, b, out)

    def solve(self, B, out=None, scale=1.0, tol=None):
        Bd = B if B.dim() == 2 else B.reshape(1, -1)
        X = torch.empty_like(Bd) if out is None else (out if out.dim() == 2 else out.reshape(1, -1))
        for j in range(Bd.shape[0]):
            self._one(Bd[j], self.tmp, tol)
            dense.copy(self.tmp, X[j], self.n)
        if scale != 1.0:
            dense.scal(X, scale, X.numel())
        return X.reshape(B.shape)

    def solve_add(self, B, add, out, scale=1.0):
        Bd = B if B.dim() == 2 else B.reshape(1, -1)
        self._one(Bd[0], self.tmp, sweep=True)         # a refinement sweep: the right-hand side is a residual
        dense.axpy(1.0, add, self.tmp, self.n)
        dense.copy(self.tmp, out, self.n)
        if scale != 1.0:
            dense.scal(out, scale, self.n)
        return out


class WEPFactorizedLinSolver(FactorizeLinSolver):
    """Waveguide.jl:466-480: the Schur complement is assembled and factorised once (host SuperLU -> device schedule, K5);
    every lin_solve is followed by the refinement of FactorizeLinSolver on the full operator"""

    def __init__(self, nep, lam, kwargs=(), umfpack_refinements=10, expected_solves=200):
        self.ops = SchurOps(nep, lam)
        self.schur_lu = DeviceLU(construct_WEP_schur_complement(nep, lam), expected_solves=expected_solves)
        inner = lambda rhs, q, tol: self.schur_lu.solve(rhs.reshape(1, -1), out=q.reshape(1, -1))
        super().__init__(nep, lam, umfpack_refinements, _lu=_SchurSolve(self.ops, inner))


class WEPBackslashLinSolver(LinSolver):
    """Waveguide.jl:449-463: `schur_comp \\ rhs`, a fresh factorisation at every lin_solve"""

    def __init__(self, nep, lam, kwargs=()):
        self.nep, self.lam = nep, complex(lam)
        self.ops = SchurOps(nep, lam)
        self.schur_comp = construct_WEP_schur_complement(nep, lam)

    def solve_dev(self, b, out=None, scale=1.0):
        lu = DeviceLU(self.schur_comp, expected_solves=1)
        inner = lambda rhs, q, tol: lu.solve(rhs.reshape(1, -1), out=q.reshape(1, -1))
        return _SchurSolve(self.ops, inner).solve(b, out=out, scale=scale)


class _SchurOperator:
    """what GMRESLinSolver needs from a `nep`: size and the operator action"""

    def __init__(self, ops):
        self.ops = ops
        self.out = torch.empty(ops.N, dtype=CDT, device="cuda")

    def size(self, d=None):
        return self.ops.N

    def compute_Mlincomb(self, lam, V):
        return self.ops.matvec(V, self.out)


class WEPGMRESLinSolver(LinSolver):
    """Waveguide.jl:428-446: matrix-free restarted GMRES on the Schur complement (device basis, K6 orthogonalisation);
    kwargs as ((name, value), ...) or a dict: Pl (e.g. wep_generate_preconditioner), reltol / tol, restart, maxiter.

    `refinements` (not in the reference; 0 = its behaviour): iterative refinement of the full system around the GMRES solve
    with the stopping rule of FactorizeLinSolver (componentwise backward error).  Left-preconditioned GMRES controls the
    preconditioned residual; at n = 10^6 the true residual levels off near 5e-12 however small reltol is chosen, while
    a few cheap sweeps (reltol ~ 1e-6, each on the residual of the last) reach the accuracy of a direct solve.
    kwargs key `sweep_reltol`: the inner tolerance of those sweeps (a correction of relative size 1e-9 needs no nine
    digits of its own); default = reltol."""
    accepts_tol = True

    def __init__(self, nep, lam, kwargs=(), refinements=0):
        self.nep, self.lam = nep, complex(lam)
        self.ops = SchurOps(nep, lam)
        kw = dict(kwargs)
        kw.pop("log", None)
        self.sweep_reltol = kw.pop("sweep_reltol", None)
        self.gmres = GMRESLinSolver(_SchurOperator(self.ops), self.lam, kw)
        self.iterations = []

        self._graph = None
        # one operator step = 2 foreign calls (nep_wep_schur_matvec, nep_wep_smw_apply: 11 launches) when both fused forms apply:
        # issued directly, without the copy in / copy out of a graph on fixed buffers (measured on C5: 1.29 s against 1.35 s);
        # the graph remains for the piecewise routes (~30 launches from Python per step)
        Pl = self.gmres._Pl_call
        direct = self.ops.stencil is not None and isinstance(Pl, WEPPreconditioner) and Pl.fused_available()
        if Pl is not None and os.environ.get("NEP_WEP_GRAPH", "0" if direct else "1") != "0":
            self._capture_step()
        elif direct:
            # straight from basis vector j into basis vector j + 1: no operator-owned output block and no copy behind it
            ops_ = self.ops

            def direct_step(v, w):
                ops_.matvec(v, w)
                Pl(w)
            self.gmres.fused_step = direct_step

        def inner(rhs, q, tol, sweep=False):
            if sweep and self.sweep_reltol is not None:
                tol = float(self.sweep_reltol)
            elif self.gmres.reltol is not None:
                tol = None
            self.gmres.solve_dev(rhs, out=q, tol=tol)
            self.iterations.append(self.gmres.iterations)
        self.schur = _SchurSolve(self.ops, inner)
        self.refined = FactorizeLinSolver(nep, lam, refinements, _lu=self.schur) if refinements > 0 else None

    def _capture_step(self):
        """one GMRES operator step w = Pl^{-1} S v (1 K1 call, ~12 GEMMs, ~15 small kernels) recorded once as a hipGraph on
        fixed buffers; per iteration: copy in, replay, copy out.  Falls back to the eager sequence if capture fails."""
        N = self.ops.N
        vin = torch.empty(N, dtype=CDT, device="cuda"); wout = torch.empty(N, dtype=CDT, device="cuda")
        prec = self.gmres._Pl_call

        def step():
            self.ops.matvec(vin, wout)
            prec(wout)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                vin.zero_(); step()                      # warm-up outside the capture (lazy kernel loads)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            torch.cuda.synchronize()
        except Exception as e:                           # keep the eager path
            self._graph_error = repr(e)
            return
        self._graph, self._vin, self._wout = g, vin, wout

        def fused(v, w):
            dense.copy(v, vin, N)
            g.replay()
            dense.copy(wout, w, N)
        self.gmres.fused_step = fused

    def solve_dev(self, b, out=None, scale=1.0, tol=None):
        if self.refined is not None:
            return self.refined.solve_dev(b, out=out, scale=scale)
        # lin_solve(solver, x; tol = eps(Float64)), Waveguide.jl:555
        return self.schur.solve(b, out=out, scale=scale, tol=tol if tol else np.finfo(float).eps)


class WEPLinSolverCreator(LinSolverCreator):
    """Waveguide.jl:489-519"""

    def __init__(self, solver_type="factorized", kwargs=(), refinements=0):
        self.solver_type, self.kwargs, self.refinements = str(solver_type).lstrip(":"), kwargs, refinements

    def create_linsolver(self, nep, lam):
        if not isinstance(nep, WEP):
            raise TypeError("WEPLinSolver can only be used in combination with WEPs: type(nep)=%s" % type(nep).__name__)
        if self.solver_type == "backslash":
            return WEPBackslashLinSolver(nep, lam, self.kwargs)
        if self.solver_type == "gmres":
            return WEPGMRESLinSolver(nep, lam, self.kwargs, refinements=self.refinements)
        if self.solver_type == "factorized":
            return WEPFactorizedLinSolver(nep, lam, self.kwargs)
        raise ValueError("Unknown type of solver_type in linsolvercreator:%s" % self.solver_type)


# ------------------------------------------------------------------------------------------------ preconditioner
class WEPPreconditioner:
    """waveguide_preconditioner.jl:10-421 on the device.  Callable on a device vector r of nx nz entries (in place):
    r <- Linv r - Linv(sum_k alpha_k E_k),  alpha = M^{-1} f(Linv r),  with Linv the Sylvester solve for A(sigma) X + X B,
    E_k the region-wise pieces of K .* X and of the boundary couplings, f the region means."""

    def __init__(self, nep, N, sigma):
        if not isinstance(nep, WEP):
            raise TypeError("the waveguide preconditioner needs a WEP")
        nz, nx = nep.nz, nep.nx
        if nz + 4 != nx:
            raise ValueError("This implementation requires nx = nz + 4. Provided NEP has nz = %d and nx = %d" % (nz, nx))
        if N < 1 or nz % N != 0:
            raise ValueError("This implementation is uniform in the blocking and therefore requires nz/N to be an integer. "
                             "Provided data is nz = %d with N = %d" % (nz, N))
        _lib.require_gpu()
        self.nep, self.N, self.sigma = nep, int(N), complex(sigma)
        wd = nep.wd
        L = nz // N
        self.L, self.mm = L, N * N + 4 * N
        self.ops = SchurOps(nep, self.sigma)
        Kmat = np.asarray(wd.K, dtype=np.complex128)
        k_bar = np.mean(Kmat)
        # ---- Sylvester operator: eigenvalues / transforms (waveguide_preconditioner.jl:127-140)
        v = np.zeros(nz, dtype=complex); v[0] = -2; v[1] = 1; v[nz - 1] = 1; v /= wd.hz ** 2
        w = np.zeros(nz, dtype=complex); w[1] = 1; w[nz - 1] = -1; w *= self.sigma / wd.hz
        D = np.fft.fft(v + w) + (self.sigma ** 2 + k_bar)
        # Sylvester solve: prime-factor DFT along z + one tridiagonal solve per z-mode along x (csrc/wep.hip); the dense
        # transform matrices of round 1 (dense GEMMs, since round 3 the library's own k_gemm_general) serve the shapes those
        # kernels do not take (nx > 2048) and remain the A/B reference behind NEP_WEP_GEMM=1
        self.sylv = None
        self.dd1 = (2 / wd.hx) / wd.hx ** 2; self.dd2 = (-1 / (2 * wd.hx)) / wd.hx ** 2
        if not os.environ.get("NEP_WEP_GEMM"):
            import ctypes as C
            h = c_vp()
            Dc = np.ascontiguousarray(D, dtype=np.complex128)
            st = lib.nep_wep_sylv_create(nz, nx, _lib.hptr(Dc), 1.0 / wd.hx ** 2, C.byref(h))
            if st == 0:
                self.sylv = h
            elif st != _lib.NEP_ERR_UNSUPPORTED:
                check(st)
        if self.sylv is None:
            S = -(4.0 / wd.hx ** 2) * np.sin(np.pi * np.arange(1, nx + 1) / (2 * (nx + 1))) ** 2
            self.G = to_dev(1.0 / (D[:, None] + S[None, :]))                                   # nz x nx
            self.Fs = to_dev(np.fft.fft(np.eye(nz), axis=0) / np.sqrt(nz))                     # unitary DFT, symmetric
            jx = np.arange(1, nx + 1)
            self.Wr = torch.from_numpy(np.ascontiguousarray(np.sqrt(2.0 / (nx + 1)) * np.sin(np.pi * np.outer(jx, jx) / (nx + 1)))).to("cuda")
        self.Ksc = to_dev(Kmat - k_bar)                                                    # K_scaled (Waveguide.jl:229-230)
        # ---- regions: indicator matrices (z: nz x N; x: nx x (N+4)), kappa = i + N j
        Bz = np.kron(np.eye(N), np.ones((L, 1)))
        Bx = np.zeros((nx, N + 4))
        Bx[0, 0] = Bx[1, 1] = Bx[nx - 2, N + 2] = Bx[nx - 1, N + 3] = 1.0
        for j in range(N):
            Bx[2 + j * L:2 + (j + 1) * L, 2 + j] = 1.0
        wx = np.ones(N + 4); wx[2:N + 2] = 1.0 / L
        self.Bz = to_dev(Bz); self.Bx = to_dev(Bx)
        self.Az = to_dev(Bz.T / L)                                                         # N x nz: region means along z
        self.Ax = to_dev(Bx * wx[None, :])                                                 # nx x (N+4): means along x
        dd1 = (2 / wd.hx) / wd.hx ** 2; dd2 = (-1 / (2 * wd.hx)) / wd.hx ** 2
        cb = np.zeros((N + 4, 2)); cb[0, 0] = dd1; cb[1, 0] = dd2; cb[N + 2, 1] = dd2; cb[N + 3, 1] = dd1
        self.cb = to_dev(cb)
        # ---- work space
        e = lambda *shape: torch.empty(shape, dtype=CDT, device="cuda")
        self.T1, self.T2, self.Y, self.Cw = e(nx, nz), e(nx, nz), e(nx, nz), e(nx, nz)
        self.tz = e(N + 4, nz)                 # nz x (N+4)
        self.fb = e(N + 4, N)                  # N x (N+4): functionals / alpha
        self.al = e(N + 4, N)
        self.eb = e(2, nz)                     # nz x 2 = [e_minus, e_plus]
        self.pb = e(2 * nz)
        self.MinvH = None
        self._G = None
        self._fused = False if os.environ.get("NEP_WEP_SMW_FUSED", "1") == "0" else None     # None: not tried yet
        self._generate()

    def __del__(self):
        try:
            if getattr(self, "sylv", None):
                lib.nep_wep_sylv_destroy(self.sylv)
                self.sylv = None
        except Exception:
            pass

    # -- pieces
    def linv(self, X):
        """in place Sylvester solve on a device nz x nx matrix (column-major): X <- F (G .* (F^H X W)) W; the two products
        with the real W run as real GEMMs"""
        nz, nx = self.nep.nz, self.nep.nx
        if self.sylv is not None:
            check(lib.nep_wep_sylv_solve(self.sylv, _p(X), stream_ptr()))
            return X
        self._xw(X, self.T1)
        zgemm(C_, N_, nz, nx, nz, 1.0, self.Fs, nz, self.T1, nz, 0.0, self.T2, nz)
        check(lib.nep_hadamard(nz * nx, 1, c_vp(self.T2.data_ptr()), nz * nx, c_vp(self.G.data_ptr()), nz * nx, stream_ptr()))
        self._xw(self.T2, self.T1)
        zgemm(N_, N_, nz, nx, nz, 1.0, self.Fs, nz, self.T1, nz, 0.0, X, nz)
        return X

    def _xw(self, X, out):
        """out = X W for the real symmetric sine-transform matrix W: the complex nz x nx block as a real 2 nz x nx block"""
        nz, nx = self.nep.nz, self.nep.nx
        check(lib.nep_dgemm(0, 0, 2 * nz, nx, nx, 1.0, _p(X), 2 * nz, c_vp(self.Wr.data_ptr()), nx, 0.0, _p(out), 2 * nz,
                            stream_ptr()))

    def functionals(self, X, out):
        """out (N x (N+4)) = region means of X (waveguide_preconditioner.jl:297-304)"""
        nz, nx, N = self.nep.nz, self.nep.nx, self.N
        if self.sylv is not None:
            check(lib.nep_wep_region_means(nz, nx, N, _p(X), _p(out), stream_ptr()))
            return out
        zgemm(N_, N_, nz, N + 4, nx, 1.0, X, nz, self.Ax, nx, 0.0, self.tz, nz)
        zgemm(N_, N_, N, N + 4, nz, 1.0, self.Az, N, self.tz, nz, 0.0, out, N)
        return out

    def expand(self, alpha, Y):
        """Y (nz x nx) = sum_k alpha_k E_k  (waveguide_preconditioner.jl:263-288, :382-412): K_scaled restricted to the regions,
        plus -P^{-1}(sigma) of the boundary pieces in the first and last column"""
        nz, nx, N = self.nep.nz, self.nep.nx, self.N
        if self.sylv is not None:
            check(lib.nep_wep_region_expand(nz, nx, N, _p(alpha), _p(self.Ksc), self.dd1, self.dd2, _p(Y), _p(self.eb), stream_ptr()))
        else:
            zgemm(N_, N_, nz, N + 4, N, 1.0, self.Bz, nz, alpha, N, 0.0, self.tz, nz)            # Bz A
            zgemm(N_, T_, nz, nx, N + 4, 1.0, self.tz, nz, self.Bx, nx, 0.0, Y, nz)               # (Bz A) Bx^T
            check(lib.nep_hadamard(nz * nx, 1, c_vp(Y.data_ptr()), nz * nx, c_vp(self.Ksc.data_ptr()), nz * nx, stream_ptr()))
            zgemm(N_, N_, nz, 2, N + 4, 1.0, self.tz, nz, self.cb, N + 4, 0.0, self.eb, nz)       # [e_-, e_+]
        self.ops.pinv(self.eb, self.pb)
        dense.axpy(-1.0, self.pb, Y, nz)                                                     # column 1       -= P_-^{-1} e_-
        check(lib.nep_axpy(nz, _lib.cd(-1.0), c_vp(self.pb.data_ptr() + 16 * nz), c_vp(Y.data_ptr() + 16 * nz * (nx - 1)),
                           stream_ptr()))                                                    # column nx      -= P_+^{-1} e_+
        return Y

    def _generate(self):
        """waveguide_preconditioner.jl:221-313: M[:, k] = f(Linv E_k) + identity, inverted once (host, mm x mm)"""
        N, mm = self.N, self.mm
        Mdev = torch.empty((mm, mm), dtype=CDT, device="cuda")            # column kappa at Mdev[kappa]
        plan = self.nep._pinv_plan()
        if self.sylv is not None and plan is not None:
            # all mm columns inside the library (one foreign call instead of ~8 per column)
            nz, nx = self.nep.nz, self.nep.nx
            st = _lib.NEP_ERR_UNSUPPORTED
            if self._fused is not False:
                # mode-space form (no back transforms, interior columns batched); falls through when the grid does not take it
                if self._G is None:
                    self._G = self._mode_means_matrix()
                st = lib.nep_wep_smw_matrix_modes(self.sylv, plan, N, _p(self.Ksc), self.dd1, self.dd2, _p(self.ops.sinv), _p(self._G),
                                                  _p(Mdev), stream_ptr())
                if st not in (0, _lib.NEP_ERR_UNSUPPORTED):
                    check(st)
            if st != 0:
                work = torch.empty(nz * nx + 4 * nz + mm, dtype=CDT, device="cuda")
                check(lib.nep_wep_smw_matrix(self.sylv, plan, N, _p(self.Ksc), self.dd1, self.dd2, _p(self.ops.sinv), _p(work), _p(Mdev),
                                             stream_ptr()))
            self._set_inverse(Mdev, mm)
            return
        unit = torch.zeros((N + 4, N), dtype=CDT, device="cuda").reshape(-1)
        one = torch.ones(1, dtype=CDT, device="cuda")
        for kappa in range(mm):
            unit.zero_()
            unit[kappa:kappa + 1].copy_(one)
            self.expand(unit, self.Y)
            self.linv(self.Y)
            self.functionals(self.Y, Mdev[kappa])
        self._set_inverse(Mdev, mm)

    def _set_inverse(self, Mdev, mm):
        """MinvH = inv(I + M)^H (alpha = (MinvH)^H f through nep_gemv_hd).  On the device by the library's Gauss-Jordan inverse
        (nep_zinv_h_dev: 2 launches per column; 1517 x 1517 in ~35 ms against 0.14 s of numpy.linalg.inv on 8 BLAS threads + two
        36 MB transfers); NEP_WEP_SMW_INV=host keeps the host route, which is also the fallback for a zero pivot."""
        self._Mdev = Mdev
        self._M_host = None
        if os.environ.get("NEP_WEP_SMW_INV", "dev") != "host":
            out = torch.empty((mm, mm), dtype=CDT, device="cuda")
            work = torch.empty(2 * mm + 2, dtype=CDT, device="cuda")
            info = C.c_int32(0)
            check(lib.nep_zinv_h_dev(mm, _p(Mdev), mm, 1.0, _p(out), mm, _p(work), C.byref(info), stream_ptr()))
            if info.value == 0:
                self.MinvH = out
                return
        self._M_host = to_host(Mdev) + np.eye(mm)
        self.MinvH = to_dev(np.linalg.inv(self._M_host).conj().T)

    @property
    def _M(self):
        if self._M_host is None:
            self._M_host = to_host(self._Mdev) + np.eye(self._Mdev.shape[0])
        return self._M_host

    @property
    def cond(self):
        """2-norm condition number of the SMW matrix (diagnostic: an SVD of the mm x mm matrix, 0.3 s at mm = 1517, so only
        computed when asked for)"""
        if getattr(self, "_cond", None) is None:
            self._cond = float(np.linalg.cond(self._M))
        return self._cond

    def _mode_means_matrix(self):
        """G (N x nz): mean over the z of region rz of column i of the inverse transform F[z, i] = exp(-2 pi i z i / nz) / sqrt(nz)
        (what nep_wep_sylv_solve's last kernel applies) -- the region means of F U are G (U summed over the x-regions)"""
        nz, N, L = self.nep.nz, self.N, self.L
        i = np.arange(nz)
        # mean over z = rz L + l of w^(z i), w = exp(-2 pi i / nz):  w^(rz L i) * (sum_l w^(l i)) / L  (exponents reduced mod nz)
        inner = np.exp(-2j * np.pi * (np.outer(np.arange(L), i) % nz) / nz).sum(axis=0) / L                 # nz
        outer = np.exp(-2j * np.pi * (np.outer(np.arange(N) * L, i) % nz) / nz)                              # N x nz
        G = outer * inner[None, :] / np.sqrt(nz)
        return to_dev(np.ascontiguousarray(G).T)                                       # to_dev stores (rows, cols) column-major

    def fused_available(self):
        """True when nep_wep_smw_apply takes this grid (tried once on a zero vector)"""
        if self._fused is None:
            self(torch.zeros(self.nep.nz * self.nep.nx, dtype=CDT, device="cuda"))
        return self._fused is True

    def __call__(self, r):
        """solve_smw (waveguide_preconditioner.jl:323-421), in place on the device vector r"""
        nz, nx, N, mm = self.nep.nz, self.nep.nx, self.N, self.mm
        if self.sylv is not None and self._fused is not False:
            plan = self.nep._pinv_plan()
            if plan is not None:
                if self._G is None:
                    self._G = self._mode_means_matrix()
                st = lib.nep_wep_smw_apply(self.sylv, plan, N, _p(self.Ksc), self.dd1, self.dd2, _p(self.ops.sinv), _p(self.MinvH),
                                           _p(self._G), _p(r), stream_ptr())
                if st == 0:
                    self._fused = True
                    return r
                if st != _lib.NEP_ERR_UNSUPPORTED:
                    check(st)
            self._fused = False                                                # piecewise route from now on
        self.linv(r)                                                           # C = Linv r
        self.functionals(r, self.fb)
        check(lib.nep_gemv_hd(c_vp(self.MinvH.data_ptr()), mm, mm, mm, c_vp(self.fb.data_ptr()), None, c_vp(self.al.data_ptr()),
                              stream_ptr()))
        self.expand(self.al, self.Y)
        self.linv(self.Y)
        dense.axpy(-1.0, self.Y, r, nz * nx)
        return r


def wep_generate_preconditioner(nep, N, sigma):
    """waveguide_preconditioner.jl:36-47"""
    return WEPPreconditioner(nep, N, sigma)
