"""ctypes access to oracle/c/libnep_cpu_ref.so (C restatement of the CPU hot path).
Test / cpu_baseline infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "libnep_cpu_ref.so")


def load():
    src = os.path.join(_DIR, "nep_cpu.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B"])
    lib = C.CDLL(_SO)
    lib.ref_gs_pass.restype = C.c_double
    lib.ref_omp_threads.restype = C.c_int32
    return lib


class CscTerms:
    def __init__(self, Av):
        self.n = Av[0].shape[0]
        self.mt = len(Av)
        self.keep = []
        self.cp = (C.c_void_p * self.mt)(); self.rv = (C.c_void_p * self.mt)(); self.nz = (C.c_void_p * self.mt)()
        self.rp = (C.c_void_p * self.mt)(); self.ci = (C.c_void_p * self.mt)(); self.nzr = (C.c_void_p * self.mt)()
        for i, A in enumerate(Av):
            A = sp.csc_matrix(A); A.sort_indices()
            a = np.ascontiguousarray(A.indptr, dtype=np.int32); b = np.ascontiguousarray(A.indices, dtype=np.int32)
            c = np.ascontiguousarray(A.data.real, dtype=np.float64)
            self.keep += [a, b, c]
            self.cp[i] = a.ctypes.data; self.rv[i] = b.ctypes.data; self.nz[i] = c.ctypes.data
            R = sp.csr_matrix(A); R.sort_indices()                       # row layout for the OpenMP variant
            a = np.ascontiguousarray(R.indptr, dtype=np.int32); b = np.ascontiguousarray(R.indices, dtype=np.int32)
            c = np.ascontiguousarray(R.data.real, dtype=np.float64)
            self.keep += [a, b, c]
            self.rp[i] = a.ctypes.data; self.ci[i] = b.ctypes.data; self.nzr[i] = c.ctypes.data


def mlincomb(lib, terms, Cm, V):
    """z = sum_i A_i (V C[:,i]) with the reference's per-term gemv + CSC scatter structure"""
    V = np.asfortranarray(V, dtype=np.complex128); Cm = np.asfortranarray(Cm, dtype=np.complex128)
    n, k = V.shape
    z = np.empty(n, dtype=np.complex128); work = np.empty(n, dtype=np.complex128)
    lib.ref_mlincomb_csc(C.c_int64(n), C.c_int32(terms.mt), terms.cp, terms.rv, terms.nz, C.c_int32(k),
                         Cm.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p), C.c_int64(n),
                         z.ctypes.data_as(C.c_void_p), work.ctypes.data_as(C.c_void_p))
    return z


def mlincomb_omp(lib, terms, Cm, V, W=None):
    """the same product with the all-cores (OpenMP, row-parallel CSR) variant"""
    V = np.asfortranarray(V, dtype=np.complex128); Cm = np.asfortranarray(Cm, dtype=np.complex128)
    n, k = V.shape
    z = np.empty(n, dtype=np.complex128)
    if W is None:
        W = np.empty((n, terms.mt), dtype=np.complex128, order="F")
    lib.ref_mlincomb_csr_omp(C.c_int64(n), C.c_int32(terms.mt), terms.rp, terms.ci, terms.nzr, C.c_int32(k),
                             Cm.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p), C.c_int64(n),
                             z.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p))
    return z
