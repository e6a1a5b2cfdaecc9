"""NEP types on the MI355X backend -- the host-side mirror of the reference's NEP interface.

Same names and argument meaning as the reference (src/NEPCore.jl:89-194, src/NEPTypes.jl):
`size`, `issparse`, `compute_Mlincomb(nep, lam, V[, a[, startder]])`, `compute_Mder(nep, lam, i)`,
`compute_MM(nep, S, V)`, `get_Av`, `get_fv`; types `SPMF_NEP`, `DEP`, `PEP`, `SumNEP`, `DerSPMF`,
`shift_and_scale`.  Every AbstractSPMF owns one device object (`nep_spmf`, a stacked CSR of all its
A_i) and all compute_* calls run the HIP kernels behind the C ABI; the only host arithmetic is the
O(k*m_t) coefficient block and compute_Mder (sparse linear combination for the one-off host LU,
src/NEPTypes.jl:343-367).

Dense blocks on the device are torch complex128 tensors used as raw memory: a column-major
n x k block is a contiguous tensor of shape (k, n) (row i of the tensor = column i of the block).
Inputs may be NumPy arrays (uploaded, result downloaded -> same semantics as the reference: a new
vector is returned and V is never modified) or such tensors (result stays on the device).
"""
import ctypes as C

import os

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib, funcs
from ._lib import lib, check, hptr, c_vp, c_i32, c_i64

CDT = torch.complex128


_RAW = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """raw handle of torch's current stream on this process's device.  torch.cuda.current_stream().cuda_stream costs
    ~10 us per call (Stream object construction) -- 1 000 kernel launches of a gun iar run = 10 ms; the raw accessor that
    torch's own inductor / triton glue uses costs well under 1 us and sees stream contexts and graph capture alike."""
    if _RAW is None:
        return c_vp(torch.cuda.current_stream().cuda_stream)
    return c_vp(_RAW(torch.cuda.current_device()))


def dev_zeros(cols, rows):
    return torch.zeros((cols, rows), dtype=CDT, device="cuda")


def dev_empty(cols, rows):
    return torch.empty((cols, rows), dtype=CDT, device="cuda")


def to_dev(A):
    """host column-major block (n x k ndarray or vector) -> device (k, n) tensor"""
    _lib.require_gpu()
    A = np.asarray(A, dtype=np.complex128)
    if A.ndim == 1:
        A = A.reshape(-1, 1)
    return torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda")


def to_host(T):
    """device (k, n) tensor -> host n x k ndarray"""
    if not T.is_cuda:
        return T.numpy().T.copy()          # a fresh array, as on the device branch (never a view that aliases the tensor)
    # through a pinned block (torch's caching host allocator hands the same pages out again once the caller has dropped the array):
    # a pageable `T.cpu()` of the 7 MB gun block runs at ~10 GB/s, the pinned copy at the link's rate
    h = torch.empty(T.shape, dtype=T.dtype, pin_memory=True)
    with torch.cuda.device(T.device):
        h.copy_(T, non_blocking=True)
        torch.cuda.current_stream(T.device).synchronize()      # the copy was enqueued on T's device's current stream
    return h.numpy().T.copy()


def to_host_cm(T):
    """device (k, n) tensor -> host n x k ndarray in COLUMN-major order (as the reference's Julia arrays are): the transposed view of
    the downloaded block, no second, transposing pass over it on the host (the eigenvector block a Krylov driver returns)"""
    return T.cpu().numpy().T


def is_dev(x):
    return isinstance(x, torch.Tensor)


def dptr(T, col=0, row=0):
    assert T.dtype == CDT and T.is_contiguous()
    return c_vp(T.data_ptr() + 16 * (col * T.shape[-1] + row))


def _to_csr(A):
    if sp.issparse(A):
        M = sp.csr_matrix(A)
        M.sum_duplicates()
        M.sort_indices()
    else:
        M = sp.csr_matrix(np.asarray(A))
        # keep explicit zeros out; dense matrices become full CSR rows
    return M


def _term_arrays(Av):
    """per-term CSR arrays in the argument layout of nep_spmf_create (the caller keeps `keep` alive)"""
    mt = len(Av)
    csr = [_to_csr(A) for A in Av]
    keep = []
    rp = (c_vp * mt)(); ci = (c_vp * mt)(); vv = (c_vp * mt)()
    isc = (c_i32 * mt)()
    for i, M in enumerate(csr):
        cplx = np.iscomplexobj(M.data)
        ip = np.ascontiguousarray(M.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(M.indices, dtype=np.int32)
        dv = np.ascontiguousarray(M.data, dtype=np.complex128 if cplx else np.float64)
        keep += [ip, ix, dv]
        rp[i] = ip.ctypes.data; ci[i] = ix.ctypes.data; vv[i] = dv.ctypes.data
        isc[i] = 1 if cplx else 0
    return rp, ci, vv, isc, keep


_TILE_KEYS = ("blocks", "max_footprint", "stride", "xp", "zp", "entries", "footprint_slots", "stream_bytes")


def tiles_analyze(Av, k=3):
    """host-only dry run of the footprint tiles of the one-launch K1 kernel (nep_spmf_tiles_analyze): tile statistics and
    the relative difference between z = sum_t A_t (V c_t) evaluated through the tiles and directly.  Needs no GPU."""
    rp, ci, vv, isc, keep = _term_arrays(Av)
    info = (c_i64 * 8)(); err = C.c_double(0.0)
    check(lib.nep_spmf_tiles_analyze(Av[0].shape[0], len(Av), rp, ci, vv, isc, int(k), info, C.byref(err)))
    d = dict(zip(_TILE_KEYS, [int(x) for x in info]))
    d["max_rel_err"] = float(err.value)
    return d


class SPMFDevice:
    """Owns the device-side stacked CSR (nep_spmf handle)."""

    def __init__(self, Av):
        _lib.require_gpu()
        self.n = Av[0].shape[0]
        self.mt = len(Av)
        rp, ci, vv, isc, keep = _term_arrays(Av)
        h = c_vp()
        check(lib.nep_spmf_create(self.n, self.mt, rp, ci, vv, isc, C.byref(h)))
        self.h = h
        info = (c_i64 * 6)()
        check(lib.nep_spmf_info(self.h, info))
        self.nnz = int(info[2]); self.valbytes = int(info[3]); self.lanes = int(info[4])
        self.matrix_bytes = int(info[5])

    def tile_info(self):
        """footprint tiles of the one-launch compute_Mlincomb kernel (all zero: none)"""
        info = (c_i64 * 8)()
        check(lib.nep_spmf_tile_info(self.h, info))
        return dict(zip(_TILE_KEYS, [int(x) for x in info]))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.nep_spmf_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def mlincomb(self, Cmat, V, z=None, k=None, ldv=None):
        """z = sum_i A_i (V C[:,i]); V device tensor (k, n) (or any tensor + explicit k, ldv)."""
        Cm = _lib.as_c128(Cmat, "F")
        if k is None:
            k = V.shape[0]; ldv = V.shape[1]
        assert Cm.shape == (k, self.mt)
        if z is None:
            z = torch.empty(self.n, dtype=CDT, device="cuda")
        check(lib.nep_mlincomb(self.h, k, hptr(Cm), c_vp(V.data_ptr() if is_dev(V) else V), ldv,
                               c_vp(z.data_ptr()), stream_ptr()))
        return z

    def mlincomb_dev(self, Cdev, ldc, k, V, ldv, z):
        """same with the coefficient block resident on the device (nep_mlincomb_dev); V, z may be raw
        device addresses (int) or tensors."""
        va = V.data_ptr() if is_dev(V) else V
        check(lib.nep_mlincomb_dev(self.h, k, c_vp(Cdev.data_ptr()), ldc, c_vp(va), ldv, c_vp(z.data_ptr()),
                                   stream_ptr()))
        return z

    def resid_batch(self, F, QT, k, ldq):
        Fm = _lib.as_c128(F, "F")
        assert Fm.shape == (self.mt, k)
        rn = np.empty(k); qn = np.empty(k)
        check(lib.nep_resid_batch(self.h, k, hptr(Fm), c_vp(QT.data_ptr()), ldq, hptr(rn), hptr(qn), stream_ptr()))
        return rn, qn

    def resid_batch_dev(self, F, QT, k, ldq, out_dev):
        """asynchronous K2: squared norms into the device tensor out_dev (2k float64), layout of nep_resid_batch_dev"""
        Fm = _lib.as_c128(F, "F")
        assert Fm.shape == (self.mt, k)
        check(lib.nep_resid_batch_dev(self.h, k, hptr(Fm), c_vp(QT.data_ptr()), ldq, c_vp(out_dev.data_ptr()), stream_ptr()))

    def resid_batch_cm(self, F, Qc, k, row0=-1, tail=None):
        """K2 with a column-major block Qc (device (k, n) tensor): squared norms as a device tensor of 2k doubles (no
        synchronisation); row0 >= 0: rows below it in the norms, the rows from it on written to `tail` (device (k, n - row0))"""
        Fm = _lib.as_c128(F, "F")
        assert Fm.shape == (self.mt, k) and Qc.shape[0] >= k and Qc.shape[1] == self.n and Qc.is_contiguous()
        out = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
        check(lib.nep_resid_batch_cm_dev(self.h, k, hptr(Fm), c_vp(Qc.data_ptr()), self.n, int(row0), c_vp(out.data_ptr()),
                                         c_vp(tail.data_ptr()) if tail is not None else None,
                                         tail.shape[1] if tail is not None else 0, stream_ptr()))
        return out

    def algorithmic_bytes(self, k):
        """matrix bytes + 16 n k (read V) + 16 n (write z)."""
        return self.matrix_bytes + 16 * self.n * k + 16 * self.n


class DeviceCSR:
    """Rectangular complex CSR operator on the device (nep_csr handle): y = alpha A x + beta z."""

    def __init__(self, A):
        _lib.require_gpu()
        M = sp.csr_matrix(A)
        M.sum_duplicates(); M.sort_indices()
        self.shape = M.shape
        ip = np.ascontiguousarray(M.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(M.indices, dtype=np.int32)
        dv = np.ascontiguousarray(M.data, dtype=np.complex128)
        h = c_vp()
        check(lib.nep_csr_create(M.shape[0], M.shape[1], hptr(ip), hptr(ix), hptr(dv), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.nep_csr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def mv(self, alpha, x, beta, z, y):
        """x, z, y: raw device addresses (int) or tensors; z may be None when beta == 0"""
        a = lambda t: c_vp(t.data_ptr() if is_dev(t) else t) if t is not None else c_vp(0)
        check(lib.nep_csr_mv(self.h, _lib.cd(alpha), a(x), _lib.cd(beta), a(z), a(y), stream_ptr()))


class PendingNorms:
    """result of an asynchronous residual batch: ready() polls the event, get() waits and unpacks"""

    def __init__(self, pin=None, ev=None, k=0, F=None, keep=None, result=None, panel=256):
        self.pin, self.ev, self.k, self.F, self.keep, self.result = pin, ev, k, F, keep, result
        self.panel = panel            # column panel width of nep_resid_batch_dev's output layout

    def ready(self):
        return self.result is not None or self.ev.query()

    def get(self):
        if self.result is None:
            self.ev.synchronize()
            sq = self.pin.numpy()
            k = self.k
            rn = np.empty(k); qn = np.empty(k)
            P = self.panel
            for j0 in range(0, k, P):
                kk = min(P, k - j0)
                rn[j0:j0 + kk] = np.sqrt(sq[2 * j0:2 * j0 + kk])
                qn[j0:j0 + kk] = np.sqrt(sq[2 * j0 + kk:2 * j0 + 2 * kk])
            self.result = (rn, qn, self.F)
            self.keep = None
        return self.result


# ----------------------------------------------------------------------------------------------
class NEP:
    def size(self, d=None):
        return (self.n, self.n) if d is None else self.n

    # ---- the reference's generic fallbacks (src/NEPCore.jl:218-270) for NEP types that implement only ONE of the three compute
    # functions: a user type assigns e.g. `compute_Mlincomb = NEP.compute_Mlincomb_from_MM`, exactly as the reference's
    # `compute_Mlincomb(nep::MyNEP, lam, V, a) = compute_Mlincomb_from_MM(nep, lam, V, a)`.  The work is done by the type's own
    # compute_MM / compute_Mder -- on the device when those are (SPMF types; Mder_NEP's rectangular-CSR operator below).
    def compute_Mlincomb_from_MM(self, lam, V, a=None, startder=0):
        """NEPCore.jl:218-228: Mlincomb through compute_MM of the bidiagonal S = diag(lam) + subdiag(a_{j+1} / a_j * j); V and a
        are not modified (the reference's non-`!` form copies them)"""
        host = not is_dev(V)
        Vh = np.array(to_host(V) if not host else V, dtype=np.complex128, copy=True)
        if Vh.ndim == 1:
            Vh = Vh.reshape(-1, 1)
        k = Vh.shape[1]
        a = np.ones(k, dtype=np.complex128) if a is None else np.array(a, dtype=np.complex128, copy=True)
        if startder > 0:                                   # NEPCore.jl:156-160
            a = np.concatenate([np.zeros(startder, dtype=np.complex128), a])
            Vh = np.hstack([np.zeros((Vh.shape[0], startder), dtype=np.complex128), Vh]); k += startder
        z0 = a == 0
        Vh[:, z0] = 0
        a[z0] = 1
        S = np.diag(np.full(k, complex(lam)))
        if k > 1:
            S = S + np.diag((a[1:k] / a[0:k - 1]) * np.arange(1, k), -1)
        Z = self.compute_MM(S, Vh)
        z = a[0] * np.asarray(to_host(Z) if is_dev(Z) else Z)[:, 0]
        return z if host else to_dev(z.reshape(-1, 1))[0]

    def compute_Mlincomb_from_Mder(self, lam, V, a=None):
        """NEPCore.jl:164-172 ("poor man's" route): sum_i a_i M^(i-1)(lam) v_i with the matrices from compute_Mder"""
        host = not is_dev(V)
        Vh = np.asarray(to_host(V) if not host else V, dtype=np.complex128)
        if Vh.ndim == 1:
            Vh = Vh.reshape(-1, 1)
        k = Vh.shape[1]
        a = np.ones(k, dtype=np.complex128) if a is None else np.asarray(a, dtype=np.complex128)
        z = torch.zeros(self.n, dtype=CDT, device="cuda")
        Vd = to_dev(Vh)
        for i in range(k):
            if a[i] != 0:
                DeviceCSR(sp.csr_matrix(self.compute_Mder(lam, i), dtype=np.complex128)).mv(complex(a[i]), Vd[i], 1.0, z, z)
        return to_host(z.reshape(1, -1))[:, 0] if host else z

    def compute_Mder_from_MM(self, lam, i=0):
        """NEPCore.jl:258-265: MM of a (transposed) Jordan block yields the derivatives -- S = J^T (x) I_n of size n (i + 1), as in
        the reference only meant for small problems"""
        import math
        n = self.n
        J = np.diag(np.full(i + 1, complex(lam))) + np.diag(np.ones(i), -1)       # transpose(jordan_matrix(i + 1, lam))
        S = np.kron(J, np.eye(n))
        e = np.zeros((1, i + 1)); e[0, 0] = 1.0                                   # sparse(1.0I, 1, i + 1)[:, end:-1:1] has its one at the LAST place
        e = e[:, ::-1]
        Vb = math.factorial(i) * np.kron(e, np.eye(n))
        W = self.compute_MM(S, Vb.astype(np.complex128))
        W = np.asarray(to_host(W) if is_dev(W) else W)
        return W[:n, :n]


class Mder_NEP(NEP):
    """A NEP known only through a function lam -> M(lam) (and optionally its derivatives i <= maxder): the reference's
    `Mder_NEP` (src/nep_type_helpers.jl:6-12,106-146) and the "Custom NEP type" of test/nleigs/nleigs_nep_types.jl.  The
    matrices come from the user's host function; every product with them runs on the device (a rectangular-CSR operator per
    evaluation point, nep_csr_mv).  `nleigs` accepts it through matrix-valued divided differences."""

    def __init__(self, n, Mder_fun, maxder=0):
        self.n = int(n)
        self.Mder_fun = Mder_fun
        self.maxder = int(maxder)
        self._ops = {}                                  # (lam, i) -> DeviceCSR, small LRU

    def compute_Mder(self, lam, i=0):
        if i > self.maxder:
            raise ValueError("Derivatives higher than %d are not available" % self.maxder)
        return self.Mder_fun(lam) if self.maxder == 0 else self.Mder_fun(lam, i)

    def _op(self, lam, i=0):
        key = (complex(lam), int(i))
        op = self._ops.pop(key, None)
        if op is None:
            op = DeviceCSR(sp.csr_matrix(self.compute_Mder(lam, i), dtype=np.complex128))
            while len(self._ops) >= 8:
                self._ops.pop(next(iter(self._ops)))
        self._ops[key] = op
        return op

    def compute_Mlincomb(self, lam, V, a=None, startder=0):
        """sum_j a_j M^(j-1+startder)(lam) v_j  (NEPCore.jl:164-172, compute_Mlincomb_from_Mder) on the device"""
        host = not is_dev(V)
        Vd = to_dev(V) if host else (V if V.dim() == 2 else V.reshape(1, -1))
        k = Vd.shape[0]
        a = np.ones(k) if a is None else np.asarray(a)
        z = torch.zeros(self.n, dtype=CDT, device="cuda")
        for j in range(k):
            if a[j] != 0:
                self._op(lam, j + startder).mv(complex(a[j]), Vd[j], 1.0, z, z)
        return to_host(z.reshape(1, -1))[:, 0] if host else z

    def resid_norms(self, lams, QT):
        """(||M(lam_s) q_s||, ||q_s||, None) for the columns of the row-major block QT (one matrix assembly + upload per
        Ritz value: the price of a NEP that is only a function handle)"""
        from . import dense
        k = len(lams)
        cols = dense.rowmajor_to_cols(QT, np.arange(k, dtype=np.int32))          # (k, n) column-major n x k
        Y = torch.empty_like(cols)
        for s_ in range(k):
            self._op(lams[s_], 0).mv(1.0, cols[s_], 0.0, None, Y[s_])
        nr = np.empty(k); nq = np.empty(k)
        check(lib.nep_colnorms(self.n, k, c_vp(Y.data_ptr()), self.n, hptr(nr), stream_ptr()))
        check(lib.nep_colnorms(self.n, k, c_vp(cols.data_ptr()), self.n, hptr(nq), stream_ptr()))
        return nr, nq, None


class AbstractSPMF(NEP):
    """M(lam) = sum_i f_i(lam) A_i   (src/NEPTypes.jl:96-113)."""
    _dev = None
    _fro = None

    def get_Av(self):
        raise NotImplementedError

    def get_fv(self):
        raise NotImplementedError

    def issparse(self):
        return sp.issparse(self.get_Av()[0])

    @property
    def dev(self):
        if self._dev is None:
            self._dev = SPMFDevice(self.get_Av())
            # the other one-off derived data of the matrices are built with the upload instead of inside the first solver call:
            # the Frobenius norms (error measures) always; the term values on the union pattern (compute_Mder, the device LU)
            # only for small problems (gun: 88 598 union entries x 4 terms, 2 ms) -- on a waveguide-sized pattern the aligned block
            # costs seconds of host time and 240 MB, and the solvers used there (GMRES on the Schur complement, tiar) never ask
            # for it; it stays lazy (first compute_Mder / aligned_terms_dev call).  Failures are not swallowed.
            try:
                self.fro_norms()
            except TypeError:                 # a NEP type for which the SPMF error measure is not defined (WEP: it says so)
                pass
            if self.issparse() and sum(int(A.nnz) for A in self.get_Av()) <= int(os.environ.get("NEP_ALIGNED_PREFETCH_NNZ", "2000000")):
                self._aligned_terms()
        return self._dev

    # ---- element type of host results (test/compute_types.jl): the reference returns promote_type(eltype(nep), typeof(lam),
    # eltype(V)) -- real when the NEP is real (real matrices, functions real on the reals) and every argument is real, complex
    # otherwise.  The device computes in complex128 throughout; for an all-real call the imaginary parts are exact zeros and the
    # host result is handed back as float64.  (Julia's other precisions -- Float16/32, BigFloat -- have no counterpart here.)
    def is_real(self):
        if getattr(self, "_is_real", None) is None:
            try:
                self._is_real = bool(all(not np.iscomplexobj(A.data if sp.issparse(A) else A) for A in self.get_Av())
                                     and all(f.real_on_reals() for f in self.get_fv()))
            except Exception:
                self._is_real = False
        return self._is_real

    def _promote(self, result, *args):
        """result (host, complex128) -> float64 when the NEP and all of `args` are real"""
        if self.is_real() and all(a is None or not np.iscomplexobj(a) for a in args):
            if sp.issparse(result):
                return result.real.astype(np.float64) if np.iscomplexobj(result.data) else result
            return np.ascontiguousarray(np.real(result)) if np.iscomplexobj(result) else result
        return result

    # ---- coefficient block C[j,i] = a_j f_i^(j)(lam) (Appendix A of SURVEY.md; NEPCore.jl:218-228)
    def coeff_block(self, lam, a, startder=0):
        a = np.asarray(a, dtype=np.complex128)
        k = len(a)
        fv = self.get_fv()
        Cm = np.empty((k, len(fv)), dtype=np.complex128, order="F")
        for i, f in enumerate(fv):
            d = f.derivs(lam, k + startder)[startder:]
            Cm[:, i] = np.where(a != 0, a * d, 0.0)   # a_j == 0 drops the column (NEPTypes.jl:982-983)
        return Cm

    def compute_Mlincomb(self, lam, V, a=None, startder=0):
        """sum_j a_j M^(j-1+startder)(lam) v_j   (src/NEPCore.jl:111-160, src/NEPTypes.jl:972-1013).
        NumPy in -> NumPy out; device tensor in -> device tensor out.  V is not modified."""
        host = not is_dev(V)
        Vd = to_dev(V) if host else (V if V.dim() == 2 else V.reshape(1, -1))
        k = Vd.shape[0]
        if a is None:
            a = np.ones(k)
        if len(a) != k:
            raise ValueError("length of a must equal the number of columns of V")
        z = self.dev.mlincomb(self.coeff_block(lam, a, startder), Vd)
        return self._promote(to_host(z.reshape(1, -1))[:, 0], lam, V, a) if host else z

    compute_Mlincomb_ = compute_Mlincomb  # the `!` variant may overwrite V; ours never needs to

    def compute_Mder(self, lam, i=0):
        """M^(i)(lam) as a host sparse/dense matrix (src/NEPTypes.jl:362-394); host-side, used once per
        shift for the factorisation."""
        Av = self.get_Av(); fv = self.get_fv()
        coef = np.array([f.derivs(lam, i + 1)[i] for f in fv], dtype=np.complex128)
        al = self._aligned_terms()
        if al is not None:
            # all terms on ONE union sparsity pattern (built once, like the reference's align_sparsity_patterns option of
            # SPMF_NEP): the matrix of a shift is a 1 x m_t by m_t x nnz product instead of m_t sparse additions
            # (contour_beyn assembles 64 of them: 1.6 ms -> 0.7 ms each on gun)
            indptr, indices, D = al
            # (einsum, not `D @ coef`: OpenBLAS' threaded zgemv takes 24 ms for this 88 598 x 4 product, einsum 0.6 ms)
            return self._promote(sp.csc_matrix((np.einsum("ij,j->i", D, coef), indices, indptr), shape=Av[0].shape), lam)
        Z = None
        for A, c in zip(Av, coef):
            T = A * c
            Z = T if Z is None else Z + T
        return self._promote(Z, lam)

    def compute_Mder_batch(self, lams):
        """the matrices M(lam_b) of several shifts on the union sparsity pattern: (indptr, indices, values) with values of shape
        (B, nnz), or None when a term is dense.  One B x m_t by m_t x nnz product instead of B assemblies (contour_beyn)."""
        al = self._aligned_terms()
        if al is None:
            return None
        indptr, indices, D = al
        fv = self.get_fv()
        Cf = np.array([[f.derivs(lam, 1)[0] for f in fv] for lam in lams], dtype=np.complex128)      # B x m_t
        return indptr, indices, np.ascontiguousarray(np.einsum("ij,bj->bi", D, Cf))

    def aligned_terms_dev(self):
        """(indptr, indices, D_dev, G) for the device-side assembly of M(lam_b) batches (nep_lu_factor_dev_batch_terms): D_dev is
        the nnz x m_t term-value block on the union pattern as a DEVICE tensor (uploaded once), G = D^H D its m_t x m_t Gram
        matrix (||M(lam)||_F^2 = c^H G c without forming M); None when a term is dense"""
        al = self._aligned_terms()
        if al is None:
            return None
        if getattr(self, "_aligned_dev", None) is None:
            indptr, indices, D = al
            Dc = np.ascontiguousarray(D, dtype=np.complex128)
            self._aligned_dev = (torch.from_numpy(Dc).to("cuda"), Dc.conj().T @ Dc)
        return al[0], al[1], self._aligned_dev[0], self._aligned_dev[1]

    def _aligned_terms(self):
        """(indptr, indices, D) with D[:, t] = values of A_t scattered onto the union CSC pattern of all terms, or None if a
        term is dense"""
        if getattr(self, "_aligned", False) is not False:
            return self._aligned
        Av = self.get_Av()
        self._aligned = None
        if all(sp.issparse(A) for A in Av) and len(Av) > 0:
            n = Av[0].shape[0]
            cs = [sp.csc_matrix(A) for A in Av]
            for M in cs:
                M.sum_duplicates(); M.sort_indices()
            # union pattern from the entry KEYS (column * n + row), stored zeros included: a sparse add of pattern matrices
            # would drop explicitly stored zeros (their slot would then be missing below) and wraps at 256 overlapping terms
            keys = []
            for M in cs:
                colM = np.repeat(np.arange(M.shape[1], dtype=np.int64), np.diff(M.indptr))
                keys.append(colM * n + M.indices)
            keyU = np.unique(np.concatenate(keys))                         # increasing: columns ascending, rows sorted inside
            nnz_sum = sum(len(k_) for k_ in keys)
            if len(keyU) * len(cs) > 8 * max(nnz_sum, 1) and len(keyU) * len(cs) > (1 << 22):
                return None                                                # many disjoint (low-rank) terms: the dense nnz_union x m_t
            D = np.zeros((len(keyU), len(cs)), dtype=np.complex128)       #   block would dwarf sum(nnz); compute_Mder sums term by term
            for t, (M, key) in enumerate(zip(cs, keys)):
                pos = np.searchsorted(keyU, key)
                assert np.array_equal(keyU[pos], key)
                D[pos, t] = M.data
            indices = (keyU % n).astype(np.int32)
            indptr = np.concatenate(([0], np.cumsum(np.bincount(keyU // n, minlength=n)))).astype(np.int32)

            class _U:                                                      # the three attributes the caller reads
                pass
            U = _U(); U.indptr = indptr; U.indices = indices
            self._aligned = (U.indptr.copy(), U.indices.copy(), D)
        return self._aligned

    def compute_MM(self, S, V):
        """sum_i A_i V f_i(S)   (src/NEPTypes.jl:276-319): host f_i(S), device GEMM + SpMM."""
        from .dense import gemm_ts
        S_in = S
        S = np.atleast_2d(np.asarray(S, dtype=np.complex128))
        p = S.shape[0]
        host = not is_dev(V)
        Vd = to_dev(V) if host else V
        fv = self.get_fv()
        isdiag = np.count_nonzero(S - np.diag(np.diag(S))) == 0
        Fs = []
        for f in fv:
            if isdiag:
                Fs.append(np.diag(np.array([f(s) for s in np.diag(S)], dtype=np.complex128)))
            else:
                Fs.append(np.asarray(f.matfun(S), dtype=np.complex128))
        B = np.hstack(Fs)                                    # p x (p*mt)
        mt = len(fv)
        XT = gemm_ts(Vd, B, rowmajor=True)                   # (n, p*mt) row-major
        ZT = torch.empty((self.n, p), dtype=CDT, device="cuda")
        check(lib.nep_spmm_terms(self.dev.h, p, c_vp(XT.data_ptr()), p * mt, c_vp(ZT.data_ptr()), p, stream_ptr()))
        if host:
            return self._promote(ZT.cpu().numpy(), S_in, V)
        return ZT.t().contiguous()

    # ---- driver-facing hooks: every Krylov driver goes through these three, so a NEP type with extra
    # non-SPMF structure (the WEP corner term) only overrides them
    def derivative_table(self, sigma, m, rowscale=None):
        """fD[j,i] = f_i^(j)(sigma), j = 0..m (DerSPMF, NEPTypes.jl:1108-1128).  With `rowscale` (length m) the
        block C[j-1,:] = rowscale[j-1]*fD[j,:] is uploaded once; lincomb_rowscale then uses its first k rows."""
        fD = np.column_stack([f.derivs(sigma, m + 1) for f in self.get_fv()])
        tab = {"fD": fD, "m": m, "sigma": sigma}
        if rowscale is not None:
            tab["Cdev"] = to_dev(np.asarray(rowscale)[:, None] * fD[1:m + 1, :])      # (mt, m): ldc = m
        return tab

    def lincomb_rowscale(self, tab, k, V, ldv, z):
        """z = sum_{j=1..k} rowscale_j M^(j)(sigma) V[:, j-1]   (V: device address or tensor, leading dim ldv)"""
        return self.dev.mlincomb_dev(tab["Cdev"], tab["m"], k, V, ldv, z)

    def lincomb_general(self, tab, G, V, k, ldv, z):
        """z = sum_{j<k} sum_{q} G[j,q] M^(q+1)(sigma) V[:, j]   (G: k x q host matrix)"""
        C = G @ tab["fD"][1:G.shape[1] + 1, :]
        return self.dev.mlincomb(C, V, z, k=k, ldv=ldv)

    def prefers_colmajor_ritz(self):
        """NEP_K2_CM=1: large sparse problems whose matrices have footprint tiles keep the Ritz block of a convergence check
        column-major (dense.ColMajorBlock) and K2 runs as nep_resid_batch_cm_dev.  Opt-in: the kernel is 1.4-7x faster than
        the row-major forms at n = 1e6, but on config C5 the check as a whole did not gain (resid phase 0.180 s against 0.144 s:
        the corner term on the column-major tail goes through strided torch updates) -- the entry point is there for hosts whose
        blocks are column-major anyway (Julia)."""
        if os.environ.get("NEP_K2_CM", "0") != "1" or self.n < 32768 or len(self.get_Av()) > 4:
            return False
        return self.dev.tile_info()["blocks"] > 0

    def resid_norms(self, lams, QT):
        """(||M(lam_s) q_s||, ||q_s||, F) for the k columns of the row-major block QT (or of a dense.ColMajorBlock)"""
        fv = self.get_fv()
        la = np.asarray(lams, dtype=np.complex128)
        F = np.empty((len(fv), len(la)), dtype=np.complex128, order="F")
        for i, f in enumerate(fv):
            F[i, :] = f.values(la)
        if hasattr(QT, "cpu_matrix"):                     # dense.ColMajorBlock
            o = self.dev.resid_batch_cm(F, QT.t, len(la)).cpu().numpy()
            return np.sqrt(o[:len(la)]), np.sqrt(o[len(la):]), F
        rn, qn = self.dev.resid_batch(F, QT, len(la), QT.shape[1])
        return rn, qn, F

    def resid_norms_async(self, lams, QT):
        """enqueues K2 and the device->pinned-host copy of the squared norms; returns a PendingNorms whose get() gives
        (rn, qn, F) like resid_norms.  Only for pure SPMF operators (subclasses with extra terms fall back to sync)."""
        if type(self).resid_norms is not AbstractSPMF.resid_norms or hasattr(QT, "cpu_matrix"):
            return PendingNorms(result=self.resid_norms(lams, QT))
        fv = self.get_fv()
        la = np.asarray(lams, dtype=np.complex128)
        k = len(la)
        F = np.empty((len(fv), k), dtype=np.complex128, order="F")
        for i, f in enumerate(fv):
            F[i, :] = f.values(la)
        out = torch.empty(2 * k, dtype=torch.float64, device="cuda")
        self.dev.resid_batch_dev(F, QT, k, QT.shape[1], out)
        pin = torch.empty(2 * k, dtype=torch.float64, pin_memory=True)
        pin.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return PendingNorms(pin=pin, ev=ev, k=k, F=F, keep=(out, QT), panel=min(256, max(1, 3072 // len(fv))))

    def fro_norms(self):
        if self._fro is None:
            # (sum of squares without BLAS: no thread-pool start-up for a handful of norms)
            def fro(x):
                x = np.asarray(x).ravel()
                return float(np.sqrt(np.sum(x.real * x.real) + (np.sum(x.imag * x.imag) if np.iscomplexobj(x) else 0.0)))
            self._fro = [fro(A.data) if sp.issparse(A) else fro(A) for A in self.get_Av()]
        return self._fro


def require_pure_spmf(nep, who):
    """drivers that feed their own coefficient blocks to the stacked-CSR kernels (nleigs, iar_chebyshev, ilan) are only
    correct when M(lam) is exactly its SPMF sum; NEP types with extra terms (the dense corner of the waveguide problem) are
    refused loudly instead of being solved without them"""
    if not isinstance(nep, AbstractSPMF) or type(nep).compute_Mlincomb is not AbstractSPMF.compute_Mlincomb:
        raise NotImplementedError("%s: the operator has terms outside its SPMF matrices (type %s); use iar / tiar / resinv / "
                                  "quasinewton / augnewton, which go through the NEP's own compute_Mlincomb" % (who, type(nep).__name__))


class SPMF_NEP(AbstractSPMF):
    """src/NEPTypes.jl:162-237."""

    def __init__(self, AA, fii, check_consistency=True):
        if len(AA) != len(fii):
            raise ValueError("Inconsistency: Number of supplied matrices = %d but the number of supplied "
                             "functions are = %d" % (len(AA), len(fii)))
        sps = [sp.issparse(A) for A in AA]
        if not (all(sps) or not any(sps)):
            raise ValueError("Mixing sparse and dense matrices is not allowed in SPMF_NEP. Either use a "
                             "consistent format, or split your problem into a sparse and a dense part and "
                             "use SumNEP.")
        for i, A in enumerate(AA[1:]):
            if A.shape != AA[0].shape:
                raise ValueError("The dimensions of the matrices mismatch: size(AA[1]) != size(AA[%d])" % (i + 2))
        fii = [f if isinstance(f, funcs.ScalarFun) else funcs.FromMatrixFunction(f) for f in fii]
        self.A = [sp.csc_matrix(A) if sp.issparse(A) else np.asarray(A) for A in AA]
        self.fi = list(fii)
        self.n = AA[0].shape[0]

    def get_Av(self):
        return self.A

    def get_fv(self):
        return self.fi


class LowRankMatrixAndFunction:
    """src/rk_helper/rk_nep.jl:41-53: a matrix A = L U^H of low rank with its function f.  As in the reference (:70-100)
    the factors are the LU factors of the dense block that holds the non-zeros of A, compacted to the columns of L / rows of
    U that carry anything; `L`, `U` are sparse n x r.  The row permutation of the LU (which the reference drops, relying
    on no interchange taking place) is folded into L, so A = L U^H holds for every input."""

    def __init__(self, A, f, L=None, U=None):
        import scipy.linalg as sla
        self.f = f
        if L is not None and U is not None:
            self.L, self.U = sp.csc_matrix(L), sp.csc_matrix(U)
            self.A = sp.csc_matrix(A) if A is not None and getattr(A, "nnz", 1) else sp.csc_matrix(self.L @ self.U.conj().T)
            return
        A = sp.csc_matrix(A)
        self.A = A
        n = A.shape[0]
        coo = A.tocoo()
        nz = coo.data != 0
        if not np.any(nz):
            self.L = sp.csc_matrix((n, 0)); self.U = sp.csc_matrix((n, 0))
            return
        r0, r1 = coo.row[nz].min(), coo.row[nz].max() + 1
        c0, c1 = coo.col[nz].min(), coo.col[nz].max() + 1
        Pm, Lb, Ub = sla.lu(A[r0:r1, c0:c1].toarray())
        sel = [i for i in range(min(Lb.shape[1], Ub.shape[0]))
               if np.count_nonzero(Lb[i:, i]) > 1 or np.count_nonzero(Ub[i, i:]) > 0]
        Lf = sp.lil_matrix((n, len(sel)), dtype=Lb.dtype); Lf[r0:r1, :] = (Pm @ Lb)[:, sel]
        Uf = sp.lil_matrix((n, len(sel)), dtype=Ub.dtype); Uf[c0:c1, :] = Ub[sel, :].conj().T
        self.L, self.U = sp.csc_matrix(Lf), sp.csc_matrix(Uf)


class LowRankFactorizedNEP(SPMF_NEP):
    """src/NEPTypes.jl (LowRankFactorizedNEP) + rk_nep.jl:59-67: an SPMF whose matrices carry low-rank factors
    A_i = L_i U_i^H.  The terms run through the same stacked-CSR kernels as any sparse SPMF term; `nleigs` uses the factors
    to shrink the blocks of its Krylov vectors beyond the polynomial degree from n to r = sum r_i rows
    (method_nleigs.jl:206-211,406-414)."""

    def __init__(self, Amf):
        super().__init__([M.A for M in Amf], [M.f for M in Amf])
        self.L = [M.L for M in Amf]
        self.U = [M.U for M in Amf]
        self.rank = int(sum(M.U.shape[1] for M in Amf))


class DEP(AbstractSPMF):
    """-lam I + sum_i A_i exp(-tau_i lam)   (src/NEPTypes.jl:427-513)."""

    def __init__(self, AA, tauv=(0.0, 1.0)):
        tauv = np.asarray(tauv)
        if np.iscomplexobj(tauv):
            raise ValueError("Incorrect construction of DEP. The delays need to be real.")
        self.A = [sp.csc_matrix(A) if sp.issparse(A) else np.asarray(A) for A in AA]
        self.tauv = np.array(tauv, dtype=float)
        self.n = AA[0].shape[0]

    def get_Av(self):
        J = sp.identity(self.n, format="csc") if sp.issparse(self.A[0]) else np.eye(self.n)
        return [J] + list(self.A)

    def get_fv(self):
        fv = [-funcs.ident()]
        for tau in self.tauv:
            fv.append(funcs.one() if tau == 0 else funcs.Exp(-tau))
        return fv


class PEP(AbstractSPMF):
    """sum_i lam^i A_i   (src/types_poly.jl:31-98)."""

    def __init__(self, AA):
        self.A = [sp.csc_matrix(A) if sp.issparse(A) else np.asarray(A) for A in AA]
        self.n = AA[0].shape[0]

    def get_Av(self):
        return self.A

    def get_fv(self):
        return [funcs.Monomial(i) for i in range(len(self.A))]


class SumNEP(AbstractSPMF):
    """SPMFSumNEP (src/NEPTypes.jl:845-898): the stacked CSR simply holds the terms of both halves."""

    def __init__(self, nep1, nep2):
        if nep1.size() != nep2.size():
            raise ValueError("size mismatch in SumNEP")
        self.nep1, self.nep2 = nep1, nep2
        self.n = nep1.n

    def get_Av(self):
        return list(self.nep1.get_Av()) + list(self.nep2.get_Av())

    def get_fv(self):
        return list(self.nep1.get_fv()) + list(self.nep2.get_fv())


class DerSPMF(AbstractSPMF):
    """Derivative table precomputed at sigma (src/NEPTypes.jl:1055-1160).  Shares the device object
    of the wrapped SPMF."""

    def __init__(self, spmf, sigma, m):
        self.spmf = spmf
        self.sigma = complex(sigma)
        self.n = spmf.n
        self.fD = np.column_stack([f.derivs(self.sigma, 2 * m + 2) for f in spmf.get_fv()])

    def get_Av(self):
        return self.spmf.get_Av()

    def get_fv(self):
        return self.spmf.get_fv()

    @property
    def dev(self):
        return self.spmf.dev

    def coeff_block(self, lam, a, startder=0):
        a = np.asarray(a, dtype=np.complex128)
        k = len(a)
        if complex(lam) != self.sigma or k + startder > self.fD.shape[0]:
            return self.spmf.coeff_block(lam, a, startder)
        return np.asfortranarray(np.where((a != 0)[:, None], a[:, None] * self.fD[startder:startder + k, :], 0.0))


def shift_and_scale(orgnep, shift=0, scale=1):
    """src/NEPTransformations.jl:92-105 (SPMF version: same matrices, composed functions)."""
    fv = [f.affine(scale, shift) for f in orgnep.get_fv()]
    new = SPMF_NEP(orgnep.get_Av(), fv)
    new._dev = orgnep._dev  # same matrices -> share the device object
    return new
