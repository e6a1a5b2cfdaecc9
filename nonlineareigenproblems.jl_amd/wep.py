"""Waveguide eigenvalue problem (WEP) inputs and NEP type for the device backend.

Generators restate src/gallery_extra/waveguide/waveguide_FD.jl:10-64 (FD matrices), :91-182 (wavenumbers
TAUSCH / JARLEBRING) and src/gallery_extra/waveguide/Waveguide.jl:9-106 (SPMF structure, R/Rinv, S-functions).
`nep_gallery("WEP", nx=, nz=, benchmark_problem=, delta=)` mirrors src/gallery_extra/GalleryWaveguide.jl:60-94.

Device representation: the literal SPMF has 3 + 2 nz terms whose last 2 nz matrices are dense
rank-one nz x nz blocks (2 nz^3 = 2e9 non-zeros at nz = 999).  Here the NEP is
    M(lam) = A1 + lam A2 + lam^2 A3  +  [0 0; 0 P(lam)],   P(lam) = blkdiag(Rm, Rm) diag(s(lam)) blkdiag(Rm, Rm)^H / nz
i.e. three big REAL sparse matrices in one stacked CSR (the HBM-bound part) plus a factored corner term applied as
two small dense products with Rm (nz x nz, the scaled DFT matrix of Waveguide.jl:53-65).
"""
import os

import numpy as np
import scipy.sparse as sp

from . import funcs


def generate_fd_interior_mat(nx, nz, hx, hz):
    ex = np.ones(nx); ez = np.ones(nz)
    Dxx = sp.diags([ex[:-1], -2 * ex, ex[:-1]], [-1, 0, 1], format="lil")
    Dzz = sp.diags([ez[:-1], -2 * ez, ez[:-1]], [-1, 0, 1], format="lil")
    Dzz[0, nz - 1] = 1; Dzz[nz - 1, 0] = 1                       # periodicity in z
    Dz = sp.diags([-ez[:-1], ez[:-1]], [-1, 1], format="lil")
    Dz[0, nz - 1] = -1; Dz[nz - 1, 0] = 1
    return sp.csc_matrix(Dxx) / hx ** 2, sp.csc_matrix(Dzz) / hz ** 2, sp.csc_matrix(Dz) / (2 * hz)


def generate_fd_boundary_mat(nx, nz, hx, hz):
    Iz = sp.identity(nz, format="csc")
    e1 = sp.csc_matrix(([1.0], ([0], [0])), shape=(nx, 1))
    en = sp.csc_matrix(([1.0], ([nx - 1], [0])), shape=(nx, 1))
    C1 = sp.hstack([sp.kron(e1, Iz), sp.kron(en, Iz)]) / hx ** 2
    d1 = 2 / hx; d2 = -1 / (2 * hx)
    vm = sp.csc_matrix(([d1, d2], ([0, 0], [0, 1])), shape=(1, nx))
    vp = sp.csc_matrix(([d1, d2], ([0, 0], [nx - 1, nx - 2])), shape=(1, nx))
    C2T = sp.vstack([sp.kron(vm, Iz), sp.kron(vp, Iz)])
    return sp.csc_matrix(C1), sp.csc_matrix(C2T)


def generate_wavenumber_fd(nx, nz, wg, delta):
    wg = wg.upper()
    if wg == "TAUSCH":
        xm, xp = 0.0 - delta, 2 / np.pi + 0.4 + delta
        k1 = np.sqrt(2.3) * np.pi; k2 = np.sqrt(3) * np.pi; k3 = np.pi

        def k(x, z):
            x, z = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(z, dtype=float))
            return (k1 * (x <= 0) + k2 * (x > 0) * (x <= 2 / np.pi) +
                    k2 * (x > 2 / np.pi) * (x <= 2 / np.pi + 0.4) * (z > 0.5) +
                    k3 * (x > 2 / np.pi) * (z <= 0.5) * (x <= 2 / np.pi + 0.4) + k3 * (x > 2 / np.pi + 0.4))
    elif wg == "JARLEBRING":
        xm, xp = -1.0 - delta, 1.0 + delta
        k1 = np.sqrt(2.3) * np.pi; k2 = 2 * np.sqrt(3) * np.pi; k3 = 4 * np.sqrt(3) * np.pi; k4 = np.pi

        def k(x, z):
            x, z = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(z, dtype=float))
            return (k1 * (x <= -1) + k4 * (x > 1) + k4 * (x > 0.5) * (x <= 1) * (z <= 0.4) +
                    k3 * (x > 0) * (x <= 0.5) + k3 * (x > 0.5) * (x <= 1) * (z > 0.4) +
                    k3 * (x > -1) * (x <= 0) * (z > 0.5) * (z - x / 2 <= 1) +
                    k2 * (x > -1) * (x <= 0) * (z > 0.5) * (z - x / 2 > 1) +
                    k3 * (x > -1) * (x <= 0) * (z <= 0.5) * (z + x / 2 > 0) +
                    k2 * (x > -1) * (x <= 0) * (z <= 0.5) * (z + x / 2 <= 0))
    else:
        raise ValueError("No wavenumber loaded: The given Waveguide '%s' is not supported in 'FD' discretization." % wg)
    X = np.linspace(xm, xp, nx + 2); hx = X[1] - X[0]
    Z = np.linspace(0.0, 1.0, nz + 1); hz = Z[1] - Z[0]
    K = k(X[None, 1:-1], Z[1:, None]) ** 2                 # nz x nx, vectorised column-major (z fastest)
    return K, hx, hz, float(k(-np.inf, 0.5)), float(k(np.inf, 0.5))


class WaveguideData:
    """all problem data of one waveguide discretisation"""

    def __init__(self, nx, nz, benchmark_problem="TAUSCH", delta=0.1):
        if nz % 2 == 0:
            raise ValueError("Variable nz must be odd! You have used nz = %d." % nz)
        self.nx, self.nz = int(nx), int(nz)
        self.K, self.hx, self.hz, self.Km, self.Kp = generate_wavenumber_fd(nx, nz, benchmark_problem, delta)
        self.n = nx * nz + 2 * nz
        p = (nz - 1) / 2
        self.d0 = -3 / (2 * self.hx)
        self.b = 4 * np.pi * 1j * np.arange(-p, p + 1)
        self.cM = self.Km ** 2 - 4 * np.pi ** 2 * np.arange(-p, p + 1) ** 2
        self.cP = self.Kp ** 2 - 4 * np.pi ** 2 * np.arange(-p, p + 1) ** 2
        self.bb = np.exp(-2j * np.pi * np.arange(nz) * (-p) / nz)

    def big_matrices(self):
        """[A1, A2, A3] of Waveguide.jl:17-19 as real CSR matrices (n x n)"""
        nx, nz = self.nx, self.nz
        Dxx, Dzz, Dz = generate_fd_interior_mat(nx, nz, self.hx, self.hz)
        C1, C2T = generate_fd_boundary_mat(nx, nz, self.hx, self.hz)
        Ix = sp.identity(nx, format="csr"); Iz = sp.identity(nz, format="csr")
        Q0 = sp.kron(Ix, Dzz, format="csr") + sp.kron(Dxx, Iz, format="csr") + sp.diags(self.K.ravel(order="F"))
        Q1 = sp.kron(Ix, 2 * Dz, format="csr")
        Q2 = sp.identity(nx * nz, format="csr")
        N = nx * nz
        Z12 = sp.csr_matrix((N, 2 * nz)); Z21 = sp.csr_matrix((2 * nz, N)); Z22 = sp.csr_matrix((2 * nz, 2 * nz))
        A1 = sp.bmat([[Q0, C1], [C2T, Z22]], format="csr")
        A2 = sp.bmat([[Q1, Z12], [Z21, Z22]], format="csr")
        A3 = sp.bmat([[Q2, Z12], [Z21, Z22]], format="csr")
        return [A1, A2, A3]

    def Rmat(self):
        """Rm[:, j] = R(e_j) = reverse(bb .* fft(e_j))  (Waveguide.jl:53-59) as a dense nz x nz matrix"""
        nz = self.nz
        F = np.fft.fft(np.eye(nz), axis=0)
        return (self.bb[:, None] * F)[::-1, :].copy()

    def corner_funs(self):
        """the 2 nz corner functions s_j(lam) = 1im*sqrt(lam^2 + b_j lam + c_j) + d0 (branch Im sqrt >= 0)"""
        return ([funcs.WEPSqrt(self.b[j], self.cM[j], self.d0) for j in range(self.nz)] +
                [funcs.WEPSqrt(self.b[j], self.cP[j], self.d0) for j in range(self.nz)])


# =================================================================================================
# device NEP

import torch

from . import _lib
from ._lib import lib, check, hptr, c_vp
from .nep import AbstractSPMF, CDT, to_dev, to_host, is_dev, stream_ptr


def _corner_derivs(wd, lam, k, scale=1.0):
    """D[r, j] = scale^j s_r^(j)(lam), r < 2 nz, j < k, vectorised over the 2 nz corner functions
    (same Taylor-coefficient recurrence as funcs.WEPSqrt; Waveguide.jl:580-616 computes the same numbers)."""
    lam = complex(lam)
    b = np.concatenate([wd.b, wd.b]); c = np.concatenate([wd.cM, wd.cP]).astype(np.complex128)
    q0 = lam * lam + b * lam + c
    s0 = np.sqrt(q0)
    sg = np.sign(q0.imag); sg[sg == 0] = 1.0
    t = np.zeros((2 * wd.nz, max(k, 1)), dtype=np.complex128)
    t[:, 0] = sg * s0
    q1 = (2 * lam + b) * scale
    q2 = scale * scale
    for m in range(1, k):
        qm = q1 if m == 1 else (q2 if m == 2 else 0.0)
        acc = np.zeros(2 * wd.nz, dtype=np.complex128)
        for i in range(1, m):
            acc += t[:, i] * t[:, m - i]
        t[:, m] = (qm - acc) / (2 * t[:, 0])
    fact = np.cumprod(np.concatenate([[1.0], np.arange(1, max(k, 1))]))
    D = 1j * t[:, :k] * fact[None, :k]
    D[:, 0] += wd.d0
    return D


def _corner_values(wd, lams):
    """S[r, s] = s_r(lam_s) for all 2 nz corner functions and all shifts at once: column s equals _corner_derivs(wd, lam_s, 1)[:, 0]
    (the residual batches of a convergence check evaluate up to 100 shifts: 44 us per call of the scalar form)"""
    la = np.asarray(lams, dtype=np.complex128).reshape(1, -1)
    b = np.concatenate([wd.b, wd.b]).reshape(-1, 1); c = np.concatenate([wd.cM, wd.cP]).astype(np.complex128).reshape(-1, 1)
    q0 = la * la + b * la + c
    s0 = np.sqrt(q0)
    sg = np.sign(q0.imag); sg[sg == 0] = 1.0
    return 1j * (sg * s0) + wd.d0


class WEP(AbstractSPMF):
    """Waveguide eigenvalue problem on the device: 3 real sparse terms (stacked CSR / SELL) + factored corner.
    Mirrors `nep_gallery(WEP, nx=, nz=, benchmark_problem=, neptype=, delta=)` (GalleryWaveguide.jl:60-94); both
    reference formats ("SPMF" and "WEP") describe this same operator (test/wep_small.jl:13-22)."""

    def __init__(self, nx=3 * 5 * 7, nz=3 * 5 * 7, benchmark_problem="TAUSCH", delta=0.1):
        self.wd = WaveguideData(nx, nz, benchmark_problem, delta)
        self.n = self.wd.n
        self.nx, self.nz = self.wd.nx, self.wd.nz
        self.N = self.nx * self.nz
        self.A = self.wd.big_matrices()
        self.fi = [funcs.one(), funcs.ident(), funcs.Monomial(2)]
        self._Rm = None

    def get_Av(self):
        return self.A

    def get_fv(self):
        return self.fi

    # ---- corner data on the device
    def _pinv_plan(self):
        """nep_wep_pinv handle (prime-factor DFT plan + bb on the device) for P(lam)^{-1}, or None (NEP_WEP_GEMM=1 / nz too large:
        the dense R matrices are used instead)"""
        if getattr(self, "_pinv_h", False) is False:
            self._pinv_h = None
            import os
            if not os.environ.get("NEP_WEP_GEMM"):
                import ctypes as C
                _lib.require_gpu()
                h = c_vp()
                bb = np.ascontiguousarray(self.wd.bb, dtype=np.complex128)
                if lib.nep_wep_pinv_create(self.nz, hptr(bb), C.byref(h)) == 0:
                    self._pinv_h = h
        return self._pinv_h

    def _corner_dev(self):
        if self._Rm is None:
            _lib.require_gpu()
            Rm = self.wd.Rmat()
            self._Rm = to_dev(Rm)                         # column-major nz x nz
            self._RmH = to_dev(Rm.conj().T)
        return self._Rm, self._RmH

    def _corner_lincomb(self, Dtab, V, ldv, k, z):
        """z[N:] += blkdiag(Rm,Rm) * sum_j Dtab[:, j] .* (blkdiag(Rm,Rm)^H V[N:, j]) / nz
        Dtab: device tensor (k, 2nz) = column-major 2nz x k table (1/nz already folded in)."""
        Rm, RmH = self._corner_dev()
        nz, N = self.nz, self.N
        va = V.data_ptr() if is_dev(V) else V
        st = stream_ptr()
        P = torch.empty((k, nz), dtype=CDT, device="cuda")
        t = torch.empty(nz, dtype=CDT, device="cuda")
        y = torch.empty(nz, dtype=CDT, device="cuda")
        for half in (0, 1):
            row0 = N + half * nz
            check(lib.nep_gemm_ts_dev(c_vp(RmH.data_ptr()), nz, nz, nz, c_vp(va + 16 * row0), ldv, 0, k,
                                      c_vp(P.data_ptr()), nz, 0, st))
            check(lib.nep_rowdot(nz, k, c_vp(P.data_ptr()), nz, c_vp(Dtab.data_ptr() + 16 * half * nz), 2 * nz,
                                 c_vp(t.data_ptr()), st))
            check(lib.nep_gemm_ts_dev(c_vp(Rm.data_ptr()), nz, nz, nz, c_vp(t.data_ptr()), nz, 0, 1,
                                      c_vp(y.data_ptr()), nz, 0, st))
            check(lib.nep_axpy(nz, _lib.cd(1.0), c_vp(y.data_ptr()), c_vp(z.data_ptr() + 16 * row0), st))
        return z

    # ---- driver hooks
    def derivative_table(self, sigma, m, rowscale=None):
        tab = super().derivative_table(sigma, m, rowscale)
        Dc = _corner_derivs(self.wd, sigma, m + 1) / self.nz          # 2nz x (m+1)
        tab["Dc"] = Dc
        if rowscale is not None:
            tab["Dcdev"] = to_dev(Dc[:, 1:m + 1] * np.asarray(rowscale)[None, :])   # (m, 2nz)
        return tab

    def lincomb_rowscale(self, tab, k, V, ldv, z):
        super().lincomb_rowscale(tab, k, V, ldv, z)
        return self._corner_lincomb(tab["Dcdev"], V, ldv, k, z)

    def lincomb_general(self, tab, G, V, k, ldv, z):
        super().lincomb_general(tab, G, V, k, ldv, z)
        Deff = tab["Dc"][:, 1:G.shape[1] + 1] @ G.T                      # 2nz x k
        return self._corner_lincomb(to_dev(Deff), V, ldv, k, z)

    def compute_Mlincomb(self, lam, V, a=None, startder=0):
        host = not is_dev(V)
        Vd = to_dev(V) if host else (V if V.dim() == 2 else V.reshape(1, -1))
        k = Vd.shape[0]
        a = np.ones(k) if a is None else np.asarray(a, dtype=np.complex128)
        z = self.dev.mlincomb(self.coeff_block(lam, a, startder), Vd)
        D = _corner_derivs(self.wd, lam, k + startder)[:, startder:] / self.nz
        D = np.where((a != 0)[None, :], D * a[None, :], 0.0)
        self._corner_lincomb(to_dev(D), Vd, Vd.shape[1], k, z)
        return to_host(z.reshape(1, -1))[:, 0] if host else z

    def refine_denominator_extra(self, lam, x):
        """device vector d (n complex, real parts used) with d[N:] = |P(lam)| |x[N:]| for the dense corner block P: the
        part of (|M||x|) that the SPMF terms do not see in the refinement criterion (linsolvers.FactorizeLinSolver)"""
        key = complex(lam)
        if getattr(self, "_Pabs_key", None) != key:
            Pm = np.abs(self.corner_matrix(lam))
            nz = self.nz
            self._Pabs = (to_dev(Pm[:nz, :nz].astype(np.complex128)), to_dev(Pm[nz:, nz:].astype(np.complex128)))
            self._Pabs_key = key
            self._den = torch.zeros(self.n, dtype=CDT, device="cuda")
            self._absx = torch.empty(2 * nz, dtype=CDT, device="cuda")
        nz, N = self.nz, self.N
        st = stream_ptr()
        check(lib.nep_absvec(2 * nz, c_vp(x.data_ptr() + 16 * N), c_vp(self._absx.data_ptr()), st))
        for half in (0, 1):
            check(lib.nep_gemm_ts_dev(c_vp(self._Pabs[half].data_ptr()), nz, nz, nz, c_vp(self._absx.data_ptr() + 16 * half * nz),
                                      nz, 0, 1, c_vp(self._den.data_ptr() + 16 * (N + half * nz)), nz, 0, st))
        return self._den

    def corner_matrix(self, lam, i=0):
        """dense 2nz x 2nz corner block of M^(i)(lam) (host)"""
        Rm = self.wd.Rmat(); nz = self.nz
        s = _corner_derivs(self.wd, lam, i + 1)[:, i]
        Pm = np.zeros((2 * nz, 2 * nz), dtype=np.complex128)
        Pm[:nz, :nz] = (Rm * s[:nz][None, :]) @ Rm.conj().T / nz
        Pm[nz:, nz:] = (Rm * s[nz:][None, :]) @ Rm.conj().T / nz
        return Pm

    def compute_Mder(self, lam, i=0):
        """explicit M^(i)(lam) (sparse with the dense corner) for the host factorisation.  The reference's WEP_FD has
        no compute_Mder (Waveguide.jl:384-386) and solves through a Schur complement; the SPMF format does."""
        # the SPMF part has no entries in the corner block (the three big matrices are zero there), so the dense
        # 2nz x 2nz corner is simply added as a second sparse matrix (block-diagonal: two nz x nz blocks)
        M = sp.csc_matrix(super().compute_Mder(lam, i), dtype=np.complex128)
        nz, N, n = self.nz, self.N, self.n
        Pm = self.corner_matrix(lam, i)
        ii, jj = np.meshgrid(np.arange(nz), np.arange(nz), indexing="ij")
        rows = np.concatenate([(N + ii).ravel(), (N + nz + ii).ravel()])
        cols = np.concatenate([(N + jj).ravel(), (N + nz + jj).ravel()])
        vals = np.concatenate([Pm[:nz, :nz].ravel(), Pm[nz:, nz:].ravel()])
        Cn = sp.csc_matrix((vals, (rows, cols)), shape=(n, n))
        corner_spmf = M[N:, N:]
        if corner_spmf.nnz:
            # general case: entries of the SPMF part inside the corner block are overwritten, as the assignment did
            M = M - sp.bmat([[sp.csc_matrix((N, N)), None], [None, corner_spmf]], format="csc")
        return sp.csc_matrix(M + Cn)

    def resid_norms_async(self, lams, QT):
        """enqueues the residual batch and the corner term; the returned object's get() runs the two synchronising read-backs
        (tiar's deferred convergence checks: the host prepares check k + 1 while the device works on check k)"""
        from .nep import PendingNorms
        if hasattr(QT, "cpu_matrix") or os.environ.get("NEP_WEP_RESID_SPLIT", "1") == "0":
            return PendingNorms(result=self.resid_norms(lams, QT))
        fin = self.resid_norms(lams, QT, _defer=True)
        ev = torch.cuda.Event(); ev.record()

        class _P:
            def __init__(s_):
                s_.result = None
            def ready(s_):
                return s_.result is not None or ev.query()
            def get(s_):
                if s_.result is None:
                    s_.result = fin()
                return s_.result
        return _P()

    def resid_norms(self, lams, QT, _defer=False):
        """(||M(lam_s) q_s||, ||q_s||, F) for the k columns of the row-major block QT.  ONE pass over the three sparse terms
        (nep_resid_split_dev) gives the squared norms of the SPMF residual over the N interior rows, the squared norms of Q
        and the residual rows of the 2 nz boundary unknowns; the dense corner term (Waveguide.jl:351-374) is added to that
        2 nz x k block only.  (Until round 3 the whole n x k residual block was written, the corner added, and two column-norm
        kernels read both blocks again: four times the bytes of Q per check.  NEP_WEP_RESID_SPLIT=0 keeps that form.)"""
        la = np.asarray(lams, dtype=np.complex128)
        k = len(la)
        F = np.empty((3, k), dtype=np.complex128, order="F")
        for i, f in enumerate(self.fi):
            F[i, :] = f.values(la)
        n, nz, N = self.n, self.nz, self.N
        st = stream_ptr()
        if hasattr(QT, "cpu_matrix"):
            # column-major Ritz block (dense.ColMajorBlock, device (k, n)): tiled K2 with contiguous column loads; the
            # residual rows of the 2 nz boundary unknowns come back column-major and take the corner term
            Qc = QT.t
            tail = torch.empty((k, 2 * nz), dtype=CDT, device="cuda")
            out = self.dev.resid_batch_cm(F, Qc, k, row0=N, tail=tail)
            Rm, RmH = self._corner_dev()
            S = _corner_values(self.wd, la) / nz
            Sd = to_dev(S)
            P = torch.empty((k, nz), dtype=CDT, device="cuda")
            Y = torch.empty((k, nz), dtype=CDT, device="cuda")
            for half in (0, 1):
                row0 = N + half * nz
                check(lib.nep_gemm_ts_dev(c_vp(RmH.data_ptr()), nz, nz, nz, c_vp(Qc.data_ptr() + 16 * row0), n, 0, k,
                                          c_vp(P.data_ptr()), nz, 0, st))
                check(lib.nep_hadamard(nz, k, c_vp(P.data_ptr()), nz, c_vp(Sd.data_ptr() + 16 * half * nz), 2 * nz, st))
                check(lib.nep_gemm_ts_dev(c_vp(Rm.data_ptr()), nz, nz, nz, c_vp(P.data_ptr()), nz, 0, k,
                                          c_vp(Y.data_ptr()), nz, 0, st))
                tail[:, half * nz:(half + 1) * nz] += Y
            tn2 = (tail.real ** 2 + tail.imag ** 2).sum(dim=1)
            o = out.cpu().numpy()
            return np.sqrt(o[:k] + tn2.cpu().numpy()), np.sqrt(o[k:]), F
        ldq = QT.shape[1]
        split = os.environ.get("NEP_WEP_RESID_SPLIT", "1") != "0"
        if split:
            out = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
            RT = torch.empty((2 * nz, k), dtype=CDT, device="cuda")                 # residual rows N .. n-1 only
            check(lib.nep_resid_split_dev(self.dev.h, k, hptr(F), c_vp(QT.data_ptr()), ldq, N, c_vp(out.data_ptr()),
                                          c_vp(RT.data_ptr()), k, st))
            tail0 = 0
        else:
            RT = torch.empty((n, k), dtype=CDT, device="cuda")
            check(lib.nep_resid_block(self.dev.h, k, hptr(F), c_vp(QT.data_ptr()), ldq, c_vp(RT.data_ptr()), k, st))
            tail0 = N
        # corner: R[N:, s] += Rfull diag(s(lam_s)) Rfull^H Q[N:, s] / nz
        Rm, RmH = self._corner_dev()
        S = _corner_values(self.wd, la) / nz      # 2nz x k
        Sd = to_dev(S)
        P = torch.empty((k, nz), dtype=CDT, device="cuda")
        Y = torch.empty((nz, k), dtype=CDT, device="cuda")
        for half in (0, 1):
            row0 = N + half * nz
            check(lib.nep_gemm_ts_dev(c_vp(RmH.data_ptr()), nz, nz, nz, c_vp(QT.data_ptr() + 16 * row0 * ldq), ldq, 1, k,
                                      c_vp(P.data_ptr()), nz, 0, st))
            check(lib.nep_hadamard(nz, k, c_vp(P.data_ptr()), nz, c_vp(Sd.data_ptr() + 16 * half * nz), 2 * nz, st))
            check(lib.nep_gemm_ts_dev(c_vp(Rm.data_ptr()), nz, nz, nz, c_vp(P.data_ptr()), nz, 0, k,
                                      c_vp(Y.data_ptr()), k, 1, st))
            check(lib.nep_axpy(nz * k, _lib.cd(1.0), c_vp(Y.data_ptr()), c_vp(RT.data_ptr() + 16 * (tail0 + half * nz) * k), st))
        rn = np.empty(k); qn = np.empty(k)
        if split:
            def finish():
                check(lib.nep_rowmajor_colnorms(2 * nz, k, c_vp(RT.data_ptr()), k, hptr(rn), st))      # (synchronises)
                o = out.cpu().numpy()
                return np.sqrt(o[:k] + rn ** 2), np.sqrt(o[k:]), F
            if _defer:
                return finish
            return finish()
        else:
            check(lib.nep_rowmajor_colnorms(n, k, c_vp(RT.data_ptr()), k, hptr(rn), st))
            check(lib.nep_rowmajor_colnorms(n, k, c_vp(QT.data_ptr()), ldq, hptr(qn), st))
        return rn, qn, F

    def fro_norms(self):
        raise TypeError("StandardSPMFErrmeasure is not defined for the WEP (use ResidualErrmeasure)")
