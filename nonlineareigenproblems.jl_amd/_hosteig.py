"""GIL-free dense eigen-decomposition for the small Hessenberg problems of the Krylov drivers (method_iar.jl:112).

numpy.linalg.eig and scipy.linalg.eig hold the GIL for the whole LAPACK call (measured: with four eig threads the
launching thread gets the interpreter 21 % of the time), which serialises the worker threads and stalls the thread
that feeds the GPU.  ctypes releases the GIL, so zgeev is called directly in the OpenBLAS/LAPACK library NumPy or
SciPy already loaded (found through threadpoolctl).  Falls back to numpy.linalg.eig when the symbol is unavailable.
Same LAPACK routine, same output convention (right eigenvectors, unit 2-norm) as `eigen` in the reference."""
import ctypes as C
import os as _os

import numpy as np

_ZGEEV = [False]
_HESS = [False]          # (zhseqr, zhsein, int type) of the same library, or None


def _find():
    if _ZGEEV[0] is not False:
        return _ZGEEV[0]
    _ZGEEV[0] = None
    try:
        from threadpoolctl import threadpool_info
        for info in threadpool_info():
            path = info.get("filepath", "")
            if "openblas" not in path.lower():
                continue
            lib = C.CDLL(path)
            for name, itype in (("scipy_LAPACKE_zgeev64_", C.c_int64), ("scipy_LAPACKE_zgeev", C.c_int32),
                                ("LAPACKE_zgeev64_", C.c_int64), ("LAPACKE_zgeev", C.c_int32)):
                f = getattr(lib, name, None)
                if f is None:
                    continue
                f.restype = itype
                f.argtypes = [C.c_int, C.c_char, C.c_char, itype, C.c_void_p, itype, C.c_void_p, C.c_void_p, itype,
                              C.c_void_p, itype]
                _ZGEEV[0] = f
                pre = name[:-len("zgeev64_")] if name.endswith("64_") else name[:-len("zgeev")]
                suf = "64_" if name.endswith("64_") else ""
                hq = getattr(lib, pre + "zhseqr" + suf, None); hi = getattr(lib, pre + "zhsein" + suf, None)
                if hq is not None and hi is not None:
                    hq.restype = itype
                    hq.argtypes = [C.c_int, C.c_char, C.c_char, itype, itype, itype, C.c_void_p, itype, C.c_void_p, C.c_void_p, itype]
                    hi.restype = itype
                    hi.argtypes = [C.c_int, C.c_char, C.c_char, C.c_char, C.c_void_p, itype, C.c_void_p, itype, C.c_void_p,
                                   C.c_void_p, itype, C.c_void_p, itype, itype, C.c_void_p, C.c_void_p, C.c_void_p]
                    _HESS[0] = (hq, hi, itype)
                else:
                    _HESS[0] = None
                return f
    except Exception:
        pass
    return _ZGEEV[0]


def _eig_hessenberg(H, n):
    """eigenvalues by the QR algorithm WITHOUT accumulating Schur vectors (zhseqr job 'E'), eigenvectors by inverse iteration
    on the Hessenberg matrix itself (zhsein).  Measured on H_100 of the gun run (||H|| = 3.5e6): 30-40 % less CPU time than
    zgeev, residuals ||H z - theta z|| 8e-10 max / 1e-12 median against zgeev's 5e-9 / 8e-11, the eigenvectors of the 50
    dominant Ritz values equal to 1e-16 (1 - |cos|).  None when a routine reports a failure (caller falls back to zgeev)."""
    hq, hi, itype = _HESS[0]
    np_int = np.int64 if itype is C.c_int64 else np.int32
    A = np.array(H, dtype=np.complex128, order="F", copy=True)
    w = np.empty(n, dtype=np.complex128)
    z = np.empty((1, 1), dtype=np.complex128)
    if hq(102, b"E", b"N", n, 1, n, A.ctypes.data, n, w.ctypes.data, z.ctypes.data, 1) != 0:
        return None
    Hc = np.array(H, dtype=np.complex128, order="F", copy=True)
    sel = np.ones(n, dtype=np_int)
    V = np.empty((n, n), dtype=np.complex128, order="F")
    mout = itype(0)
    ifl = np.zeros(n, dtype=np_int); ifr = np.zeros(n, dtype=np_int)
    info = hi(102, b"R", b"Q", b"N", sel.ctypes.data, n, Hc.ctypes.data, n, w.ctypes.data, z.ctypes.data, 1, V.ctypes.data, n, n,
              C.byref(mout), ifl.ctypes.data, ifr.ctypes.data)
    if info != 0 or int(mout.value) != n or ifr.any():
        return None
    # zgeev's convention: unit 2-norm, the component of largest modulus real
    big = V[np.argmax(np.abs(V), axis=0), np.arange(n)]
    V *= (np.abs(big) / big / np.linalg.norm(V, axis=0))[None, :]
    if not np.all(np.isfinite(V)):
        return None
    return w, V


def eig(H, hessenberg=False):
    """(w, V) with H V = V diag(w); the LAPACK calls run without the GIL.  hessenberg=True: H is upper Hessenberg (the
    Arnoldi matrices of iar / tiar) and the cheaper eigenvalues-then-inverse-iteration route is taken"""
    f = _find()
    H = np.asarray(H, dtype=np.complex128)
    n = H.shape[0]
    if f is None or n == 0:
        return np.linalg.eig(H)
    if hessenberg and n >= 48 and _HESS[0] and not _os.environ.get("NEP_EIG_ZGEEV"):
        r = _eig_hessenberg(H, n)
        if r is not None:
            return r
    A = np.array(H, dtype=np.complex128, order="F", copy=True)          # overwritten by zgeev
    w = np.empty(n, dtype=np.complex128)
    V = np.empty((n, n), dtype=np.complex128, order="F")
    vl = np.empty((1, 1), dtype=np.complex128, order="F")
    info = f(102, b"N", b"V", n, A.ctypes.data, n, w.ctypes.data, vl.ctypes.data, 1, V.ctypes.data, n)   # 102 = column major
    if info != 0:
        return np.linalg.eig(H)
    return w, V

