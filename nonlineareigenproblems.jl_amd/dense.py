"""Dense device helpers: tall-skinny GEMM (K7), Gram-Schmidt (K6), BLAS-1 wrappers."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, hptr, c_vp, c_i32, c_i64, c_dbl
from .nep import CDT, stream_ptr

DGKS, CGS, MGS = 0, 1, 2


class DGKS_:  # marker objects mirroring IterativeSolvers' DGKS()/ClassicalGramSchmidt()/ModifiedGramSchmidt()
    code = DGKS


class ClassicalGramSchmidt:
    code = CGS


class ModifiedGramSchmidt:
    code = MGS


def _orth_code(method):
    if isinstance(method, int):
        return method
    return getattr(method, "code", DGKS)


def gemm_ts(Z, B, rowmajor=False, out=None, k=None, rows=None, ldz=None):
    """Y = Z[:, :k] * B on the FP64 matrix cores (nep_gemm_ts).
    Z: device tensor (>=k, rows) holding a column-major block (ldz = Z.shape[1] unless given).
    B: host k x p.  Returns device tensor: (p, rows) column-major block, or (rows, p) if rowmajor."""
    Bm = _lib.as_c128(B, "F")
    kk, p = Bm.shape
    if k is None:
        k = kk
    assert kk == k
    if rows is None:
        rows = Z.shape[-1]
    if ldz is None:
        ldz = Z.shape[-1]
    if out is None:
        out = torch.empty((rows, p) if rowmajor else (p, rows), dtype=CDT, device="cuda")
    ldy = out.shape[1]
    check(lib.nep_gemm_ts(c_vp(Z.data_ptr()), ldz, rows, k, hptr(Bm), Bm.shape[0], p, c_vp(out.data_ptr()), ldy,
                          1 if rowmajor else 0, stream_ptr()))
    return out


def gemm_ts_dev(Z, Bd, p, ldb, rowmajor=False, out=None, k=None, rows=None, ldz=None, b_rowmajor=False):
    """Y = Z[:, :k] * B with B resident on the device (nep_gemm_ts_dev): B[c, j] = Bd[j * ldb + c] (or Bd[c * ldb + j] when
    b_rowmajor) -- the eigenvector block nep_hess_eigvecs_dev leaves in HBM goes straight into the Ritz GEMM"""
    if rows is None:
        rows = Z.shape[-1]
    if ldz is None:
        ldz = Z.shape[-1]
    if out is None:
        out = torch.empty((rows, p) if rowmajor else (p, rows), dtype=CDT, device="cuda")
    check(lib.nep_gemm_ts_dev(c_vp(Z.data_ptr()), ldz, rows, k, c_vp(Bd.data_ptr()), ldb, 1 if b_rowmajor else 0, p,
                              c_vp(out.data_ptr()), out.shape[1], 1 if rowmajor else 0, stream_ptr()))
    return out


def orthogonalize_and_normalize(V, w, k, rows=None, ldv=None, active_rows=None, method=DGKS):
    """IterativeSolvers.orthogonalize_and_normalize!(V, w, h, method): returns (h, beta, npasses);
    w (device vector, `rows` entries) is orthogonalised against the first k columns of V and
    normalised in place."""
    if rows is None:
        rows = w.numel()
    if ldv is None:
        ldv = V.shape[-1]
    h = np.zeros(k, dtype=np.complex128)
    beta = c_dbl(0.0)
    npass = c_i32(0)
    act = None
    if active_rows is not None:
        act = np.ascontiguousarray(active_rows, dtype=np.int64)
        assert len(act) >= k
    check(lib.nep_orth(c_vp(V.data_ptr()), ldv, rows, k, hptr(act) if act is not None else None,
                       c_vp(w.data_ptr()), hptr(h), C.byref(beta), _orth_code(method), C.byref(npass), stream_ptr()))
    return h, beta.value, npass.value


def orthogonalize_and_normalize_dev(V, w, k, out, rows=None, ldv=None, active_dev=None, method=DGKS):
    """asynchronous variant (nep_orth_dev): nothing is read back.  `out` (device, >= k+2 complex) receives h[0..k),
    (beta,0), (passes, 2*breakdown + another_pass_wanted); `active_dev` is a DEVICE int64 tensor or None."""
    if rows is None:
        rows = w.numel()
    if ldv is None:
        ldv = V.shape[-1]
    check(lib.nep_orth_dev(c_vp(V.data_ptr()), ldv, rows, k, c_vp(active_dev.data_ptr()) if active_dev is not None else None,
                           c_vp(w.data_ptr()), c_vp(out.data_ptr()), _orth_code(method), stream_ptr()))


def gemv_h(V, w, k, rows=None, ldv=None):
    """h = V[:, :k]^H w (host result); V: device (cols, ldv) tensor = column-major block, w: device vector"""
    if rows is None:
        rows = w.numel()
    if ldv is None:
        ldv = V.shape[-1]
    h = np.empty(k, dtype=np.complex128)
    check(lib.nep_gemv_h(c_vp(V.data_ptr()), ldv, rows, k, c_vp(w.data_ptr()), hptr(h), stream_ptr()))
    return h


def gram_h(W, Y, k, p, rows, ldw=None, ldy=None):
    """G = W[:, :k]^H Y[:, :p] (k x p, host) column by column with gemv_h (W is streamed once per column of Y)"""
    ldy = Y.shape[-1] if ldy is None else ldy
    G = np.empty((k, p), dtype=np.complex128)
    for j in range(p):
        G[:, j] = gemv_h(W, Y[j], k, rows=rows, ldv=ldw)
    return G


def gemm_h_rm(WT, YT, rows, k, p, ldw=None, ldy=None):
    """C = W^H Y (k x p, host) for ROW-major device blocks WT (rows, k), YT (rows, p): K9, FP64 MFMA"""
    ldw = WT.shape[-1] if ldw is None else ldw
    ldy = YT.shape[-1] if ldy is None else ldy
    Cm = np.empty((k, p), dtype=np.complex128, order="F")
    check(lib.nep_gemm_h_rm(c_vp(WT.data_ptr()), ldw, c_vp(YT.data_ptr()), ldy, rows, k, p, hptr(Cm), stream_ptr()))
    return Cm


def nrm2(x, length=None):
    out = c_dbl(0.0)
    check(lib.nep_nrm2(length if length is not None else x.numel(), c_vp(x.data_ptr()), C.byref(out), stream_ptr()))
    return out.value


def scal(x, alpha, length=None):
    check(lib.nep_scal(length if length is not None else x.numel(), _lib.cd(alpha), c_vp(x.data_ptr()), stream_ptr()))


def axpy(alpha, x, y, length=None):
    check(lib.nep_axpy(length if length is not None else x.numel(), _lib.cd(alpha), c_vp(x.data_ptr()),
                       c_vp(y.data_ptr()), stream_ptr()))


def copy(src, dst, length=None):
    check(lib.nep_dev_copy(c_vp(dst.data_ptr()), c_vp(src.data_ptr()), 16 * (length if length is not None else src.numel()), stream_ptr()))


class ColMajorBlock:
    """a Ritz block kept COLUMN-major: `t` is a device (k, rows) tensor = rows x k column-major (what nep_gemm_ts writes with
    y_rowmajor = 0).  Large sparse problems use it for their convergence checks: the tiled residual kernel then reads every
    column contiguously (nep_resid_batch_cm_dev); everything that accepts a row-major (rows, k) block accepts this wrapper."""

    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return (self.t.shape[1], self.t.shape[0])

    def cpu_matrix(self):
        return self.t.cpu().numpy().T


def rowmajor_to_cols(QT, cols=None):
    """(rows, k) row-major device block -> (ncols, rows) column-major tensor of the chosen columns"""
    if isinstance(QT, ColMajorBlock):
        if cols is None:
            return QT.t
        return QT.t[torch.as_tensor(np.ascontiguousarray(cols, dtype=np.int64), device=QT.t.device)].contiguous()
    rows, k = QT.shape
    if cols is None:
        cols = np.arange(k)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    out = torch.empty((len(cols), rows), dtype=CDT, device="cuda")
    if len(cols):
        check(lib.nep_rowmajor_to_colmajor(rows, k, c_vp(QT.data_ptr()), k, hptr(cols), len(cols),
                                           c_vp(out.data_ptr()), rows, stream_ptr()))
    return out


HESS_EIG_KMAX = 128


def hess_eig_worksize(k):
    nb = c_i64(0)
    check(lib.nep_hess_eig_worksize(int(k), C.byref(nb)))
    return int(nb.value)


def hess_eig_dev(Hd, k, ldh=None, work=None, w=None, Z=None, mirror=None):
    """`eigen(H[1:k,1:k])` (src/method_iar.jl:112) on the device: Hd is a device tensor holding the k x k upper Hessenberg
    matrix column-major with leading dimension ldh.  Returns (w, Z): w device (k+2,) = eigenvalues, then the two status
    words (nep_hess_eigvals_dev / nep_hess_eigvecs_dev); Z device (k, k) whose row j is eigenvector j (= column-major k x k).
    Asynchronous: nothing is read back."""
    if ldh is None:
        ldh = Hd.shape[-1]
    if work is None:
        work = torch.empty(hess_eig_worksize(k), dtype=torch.uint8, device="cuda")
    if w is None:
        w = torch.empty(k + 2, dtype=CDT, device="cuda")
    if Z is None:
        Z = torch.empty((k, k), dtype=CDT, device="cuda")
    mp = c_vp(mirror.data_ptr()) if mirror is not None else None
    check(lib.nep_hess_eigvals_dev(int(k), c_vp(Hd.data_ptr()), int(ldh), c_vp(w.data_ptr()), c_vp(work.data_ptr()), mp, stream_ptr()))
    check(lib.nep_hess_eigvecs_dev(int(k), c_vp(w.data_ptr()), c_vp(Z.data_ptr()), int(Z.shape[-1]), c_vp(work.data_ptr()), mp, stream_ptr()))
    return w, Z
