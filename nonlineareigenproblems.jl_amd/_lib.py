"""ctypes binding of libnepmi355.so (the C ABI declared in include/nepmi355.h).

There is NO CPU fallback: if the library cannot be loaded the import fails, and if no GPU is
visible every compute call fails with NepError (status NEP_ERR_HIP)."""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (loads the HIP runtime that the library binds to -- see build.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnepmi355.so")


class NepError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("libnepmi355 status %d: %s" % (status, msg))
        self.status = status


NEP_OK, NEP_ERR_HIP, NEP_ERR_ARG, NEP_ERR_SINGULAR, NEP_ERR_BREAKDOWN, NEP_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
NEP_ERR_RETRY, NEP_ERR_NOCONV = -6, -7


def _load():
    # build in-tree (hipcc cross-compiles without a GPU) when the library is missing OR older than its sources: a stale
    # .so with changed argument lists would corrupt memory through ctypes.  Never falls back to a CPU path.
    from . import build
    if build.needs_build():
        import shutil
        if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            build.build_lib(verbose=False)        # serialised across processes by a file lock, linked to a temp name + rename
        elif not os.path.exists(LIB_PATH):
            raise ImportError("libnepmi355.so is missing and hipcc is not available to build it")
        elif not os.environ.get("NEP_ALLOW_STALE_LIB"):
            raise ImportError("libnepmi355.so was built from other sources than the ones in %s (digest %s, sources %s) and hipcc "
                              "is not available to rebuild it; NEP_ALLOW_STALE_LIB=1 loads it anyway"
                              % (build.CSRC, build.built_digest(), build.source_digest()))
    l = C.CDLL(LIB_PATH)
    # the binary itself says which sources it was built from: refuse a library that does not match the binding
    try:
        l.nep_src_digest.restype = C.c_char_p
        have = l.nep_src_digest().decode()
    except AttributeError:
        have = None
    if have != build.source_digest() and not os.environ.get("NEP_ALLOW_STALE_LIB"):
        raise ImportError("libnepmi355.so reports source digest %r, the binding expects %r" % (have, build.source_digest()))
    return l


lib = _load()

c_i32, c_i64, c_dbl, c_vp, c_sz = C.c_int32, C.c_int64, C.c_double, C.c_void_p, C.c_size_t
P = C.POINTER


class cdouble(C.Structure):
    _fields_ = [("re", c_dbl), ("im", c_dbl)]


class IarOpts(C.Structure):                    # nep_iar_opts (include/nepmi355.h)
    _fields_ = [("maxit", c_i32), ("check_error_every", c_i32), ("orth_method", c_i32), ("umfpack_refinements", c_i32),
                ("errmeasure", c_i32), ("refine_hint", c_i32), ("tol", c_dbl), ("neigs", c_dbl), ("sigma", cdouble), ("gamma", cdouble)]


class IarResult(C.Structure):                  # nep_iar_result
    _fields_ = [("k", c_i32), ("nconv", c_i32), ("nret", c_i32), ("refine_plan", c_i32), ("refine_hint_off", c_i32),
                ("retry_reason", c_i32)]


FV_EVAL = C.CFUNCTYPE(c_i32, c_vp, c_i32, c_vp, c_vp)     # nep_fv_eval


# name -> argtypes (restype is always int32 unless listed)
SIGNATURES = {
    "nep_version": [],
    "nep_src_digest": [],
    "nep_last_error": [],
    "nep_device_count": [P(c_i32)],
    "nep_set_device": [c_i32],
    "nep_device_name": [C.c_char_p, c_i32],
    "nep_dev_alloc": [P(c_vp), c_sz],
    "nep_dev_free": [c_vp],
    "nep_dev_memset": [c_vp, c_i32, c_sz, c_vp],
    "nep_upload": [c_vp, c_vp, c_sz, c_vp],
    "nep_download": [c_vp, c_vp, c_sz, c_vp],
    "nep_dev_copy": [c_vp, c_vp, c_sz, c_vp],
    "nep_stream_sync": [c_vp],
    "nep_stream_pair_serializes": [c_vp, c_vp, P(c_i32)],
    "nep_spmf_create": [c_i64, c_i32, P(c_vp), P(c_vp), P(c_vp), P(c_i32), P(c_vp)],
    "nep_spmf_destroy": [c_vp],
    "nep_spmf_info": [c_vp, P(c_i64)],
    "nep_spmf_tile_info": [c_vp, P(c_i64)],
    "nep_spmf_tiles_analyze": [c_i64, c_i32, P(c_vp), P(c_vp), P(c_vp), P(c_i32), c_i32, P(c_i64), P(c_dbl)],
    "nep_k1_set_mode": [c_i32],
    "nep_k2_set_sp_mode": [c_i32],
    "nep_csc_to_csr": [c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp],
    "nep_mlincomb": [c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp],
    "nep_mlincomb_dev": [c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "nep_resid_batch": [c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp],
    "nep_resid_batch_dev": [c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp],
    "nep_resid_block": [c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp],
    "nep_resid_split_dev": [c_vp, c_i32, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp],
    "nep_resid_batch_cm_dev": [c_vp, c_i32, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp],
    "nep_spmm_terms": [c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp],
    "nep_zinv_h_dev": [c_i32, c_vp, c_i64, c_dbl, c_vp, c_i64, c_vp, P(c_i32), c_vp],
    "nep_orth_qr_dev": [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp],
    "nep_zgemm_sk": [c_i32, c_i32, c_i32, c_i32, c_i32, cdouble, c_vp, c_i64, c_vp, c_i64, cdouble, c_vp, c_i64, c_i32, c_vp, c_vp],
    "nep_zgemm": [c_i32, c_i32, c_i32, c_i32, c_i32, cdouble, c_vp, c_i64, c_vp, c_i64, cdouble, c_vp, c_i64, c_vp],
    "nep_dgemm": [c_i32, c_i32, c_i32, c_i32, c_i32, c_dbl, c_vp, c_i64, c_vp, c_i64, c_dbl, c_vp, c_i64, c_vp],
    "nep_csr_create": [c_i64, c_i64, c_vp, c_vp, c_vp, P(c_vp)],
    "nep_csr_destroy": [c_vp],
    "nep_csr_mv": [c_vp, cdouble, c_vp, cdouble, c_vp, c_vp, c_vp],
    "nep_orth": [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, P(c_dbl), c_i32, P(c_i32), c_vp],
    "nep_orth_dev": [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp],
    "nep_gemv_h": [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp],
    "nep_gemm_h_rm": [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp],
    "nep_gemm_ts": [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp],
    "nep_gemm_ts_dev": [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp],
    "nep_lu_create": [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, P(c_vp)],
    "nep_lu_create_csc": [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, P(c_vp)],
    "nep_lu_refactor": [c_vp, c_vp, c_vp],
    "nep_lu_set_row_scale": [c_vp, c_vp],
    "nep_lu_is_block_schedule": [c_vp, P(c_i32)],
    "nep_lu_analyze": [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, P(c_i64)],
    "nep_lu_destroy": [c_vp],
    "nep_lu_set_expected_solves": [c_i32],
    "nep_lu_set_plan_threads": [c_i32],
    "nep_cw_backward_error": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_absvec": [c_i64, c_vp, c_vp, c_vp],
    "nep_lu_info": [c_vp, P(c_i64)],
    "nep_lu_schedule": [c_vp, P(c_i64)],
    "nep_lu_solve": [c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_dbl, c_vp],
    "nep_lu_solve_add": [c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_dbl, c_vp],
    "nep_wep_sylv_create": [c_i32, c_i32, c_vp, c_dbl, P(c_vp)],
    "nep_wep_sylv_destroy": [c_vp],
    "nep_wep_sylv_info": [c_vp, P(c_i32)],
    "nep_wep_sylv_solve": [c_vp, c_vp, c_vp],
    "nep_wep_pinv_create": [c_i32, c_vp, P(c_vp)],
    "nep_wep_pinv_destroy": [c_vp],
    "nep_wep_pinv_apply": [c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_wep_schur_matvec": [c_vp, c_vp, c_i32, c_vp, c_vp, cdouble, cdouble, c_dbl, c_dbl, c_dbl, c_dbl, c_vp, c_vp, c_vp],
    "nep_wep_smw_matrix": [c_vp, c_vp, c_i32, c_vp, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp],
    "nep_wep_smw_matrix_modes": [c_vp, c_vp, c_i32, c_vp, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp],
    "nep_wep_smw_apply": [c_vp, c_vp, c_i32, c_vp, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_wep_region_means": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp],
    "nep_wep_region_expand": [c_i32, c_i32, c_i32, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_vp, c_vp],
    "nep_iar_create": [c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, P(c_vp)],
    "nep_iar_destroy": [c_vp],
    "nep_lu_refac_create": [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_lu_refac_destroy": [c_vp],
    "nep_lu_refac_info": [c_vp, c_vp],
    "nep_lu_refac_wide_info": [c_vp, c_vp],
    "nep_lu_refac_hash": [c_vp, c_vp],
    "nep_lu_factor_dev": [c_vp, c_vp, c_i32, C.c_double, c_vp, c_vp, c_vp, c_vp],
    "nep_lu_refac_analyze": [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_lu_factor_dev_batch": [c_vp, c_i32, c_vp, c_i32, C.c_double, c_vp, c_vp, c_vp, c_vp],
    "nep_lu_factor_dev_batch_terms": [c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, C.c_double, c_vp, c_vp, c_vp],
    "nep_iar_step": [c_vp, c_i32, c_i32, c_vp],
    "nep_iar_steps": [c_vp, c_i32, c_i32, c_i32, c_vp],
    "nep_iar_wait": [c_vp, c_i32],
    "nep_iar_stream_wait": [c_vp, c_i32, c_vp],
    "nep_devprim_exclusive_sum": [c_vp, c_vp, c_i64, c_vp],
    "nep_devprim_sort_pairs": [c_vp, c_vp, c_i64, c_i32, c_vp],
    "nep_refine_review": [c_i32, c_i32, c_i32, c_vp, c_i32, c_vp],
    "nep_iar_run": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_hess_eig_worksize": [c_i32, P(c_i64)],
    "nep_hess_eigvals_dev": [c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp],
    "nep_hess_eigvecs_dev": [c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp],
    "nep_hess_eigvals_batch_dev": [c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
    "nep_hess_eigvecs_batch_dev": [c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
    "nep_comm_unique_id": [c_vp],
    "nep_comm_create": [c_i32, c_i32, c_vp, P(c_vp)],
    "nep_comm_destroy": [c_vp],
    "nep_comm_info": [c_vp, P(c_i32)],
    "nep_allgather_sum": [c_vp, c_vp, c_i64, c_vp, c_vp],
    "nep_sum_ranks": [c_vp, c_i64, c_i32, c_vp, c_vp],
    "nep_iar_shift_scale": [c_i64, c_i32, c_vp, c_vp, c_vp],
    "nep_rk_bw": [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "nep_discretize_polygon": [c_i32, c_vp, c_i32, c_vp],
    "nep_block_recur": [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "nep_axpy": [c_i64, cdouble, c_vp, c_vp, c_vp],
    "nep_scal": [c_i64, cdouble, c_vp, c_vp],
    "nep_nrm2": [c_i64, c_vp, P(c_dbl), c_vp],
    "nep_colnorms": [c_i64, c_i32, c_vp, c_i64, c_vp, c_vp],
    "nep_coldots": [c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "nep_coldotsu": [c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "nep_gemv_hd": [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "nep_rowdot": [c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "nep_hadamard": [c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp],
    "nep_rowmajor_colnorms": [c_i64, c_i32, c_vp, c_i64, c_vp, c_vp],
    "nep_rowmajor_to_colmajor": [c_i64, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp],
}

for _name, _args in SIGNATURES.items():
    _f = getattr(lib, _name)  # AttributeError here = header/library mismatch
    _f.argtypes = _args
    _f.restype = C.c_char_p if _name in ("nep_last_error", "nep_src_digest") else c_i32


def check(status):
    if status != 0:
        raise NepError(status, lib.nep_last_error().decode(errors="replace"))


def hptr(a):
    """host pointer of a C-contiguous/F-contiguous numpy array (caller keeps it alive)"""
    return a.ctypes.data_as(c_vp)


def cd(z):
    z = complex(z)
    return cdouble(z.real, z.imag)


def device_count():
    n = c_i32(0)
    check(lib.nep_device_count(C.byref(n)))
    return n.value


_pinned = [False]


def require_gpu():
    if device_count() < 1:
        raise NepError(NEP_ERR_HIP, "no HIP device visible: the MI355X backend has no CPU fallback")
    if not _pinned[0]:
        _pinned[0] = True
        from . import _affinity          # host side of this rank -> CPUs of the GPU's NUMA node (NEP_NO_PIN=1 disables)
        try:
            _affinity.pin_to_gpu_numa(torch.cuda.current_device())
        except Exception:
            pass
        import sys
        wdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_workers")
        if wdir not in sys.path:
            sys.path.insert(0, wdir)
        import nep_amd_hostlu as _nep_hostlu               # big BLAS thread pools make the host side stall at random (see there)
        _nep_hostlu.cap_blas_threads()


def as_c128(a, order="F"):
    return np.require(np.asarray(a, dtype=np.complex128), requirements=["A", "O", "F" if order == "F" else "C"])
