"""GIL-free dense eigen-decomposition for the small Hessenberg problems of the Krylov drivers (method_iar.jl:112).

numpy.linalg.eig and scipy.linalg.eig hold the GIL for the whole LAPACK call (measured: with four eig threads the
launching thread gets the interpreter 21 % of the time), which serialises the worker threads and stalls the thread
that feeds the GPU.  ctypes releases the GIL, so zgeev is called directly in the OpenBLAS/LAPACK library NumPy or
SciPy already loaded (found through threadpoolctl).  Falls back to numpy.linalg.eig when the symbol is unavailable.
Same LAPACK routine, same output convention (right eigenvectors, unit 2-norm) as `eigen` in the reference."""
import ctypes as C

import numpy as np

_ZGEEV = [False]


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
                return f
    except Exception:
        pass
    return _ZGEEV[0]


def eig(H):
    """(w, V) with H V = V diag(w); the LAPACK call runs without the GIL"""
    f = _find()
    H = np.asarray(H, dtype=np.complex128)
    n = H.shape[0]
    if f is None or n == 0:
        return np.linalg.eig(H)
    A = np.array(H, dtype=np.complex128, order="F", copy=True)          # overwritten by zgeev
    w = np.empty(n, dtype=np.complex128)
    V = np.empty((n, n), dtype=np.complex128, order="F")
    vl = np.empty((1, 1), dtype=np.complex128, order="F")
    info = f(102, b"N", b"V", n, A.ctypes.data, n, w.ctypes.data, vl.ctypes.data, 1, V.ctypes.data, n)   # 102 = column major
    if info != 0:
        return np.linalg.eig(H)
    return w, V

