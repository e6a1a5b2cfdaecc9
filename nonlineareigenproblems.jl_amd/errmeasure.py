"""Error measures (src/errmeasure.jl:91-190) evaluated with the batched residual kernel K2.

`estimate_error(errm, lam, v)` is the reference's one-pair interface; `estimate_errors(errm, lams,
QT)` evaluates all Ritz pairs of an iteration with ONE pass over the stacked CSR (the reference
issues k separate compute_Mlincomb calls, src/method_iar.jl:134-135)."""
import numpy as np
import torch

from .nep import AbstractSPMF, is_dev

EPS = np.finfo(float).eps


class Errmeasure:
    pass


class _PendingErrs:
    """errors of an asynchronous residual batch (see AbstractSPMF.resid_norms_async)"""

    def __init__(self, pending, fn, value=None):
        self.pending, self.fn, self.value = pending, fn, value

    def ready(self):
        return self.value is not None or self.pending.ready()

    def get(self):
        if self.value is None:
            self.value = self.fn(*self.pending.get())
        return self.value


def _batch_norms(nep, lams, QT):
    """QT: device (rows, k) row-major block of the k vectors. Returns ((||M(lam_s) q_s||, ||q_s||), F)."""
    rn, qn, F = nep.resid_norms(lams, QT)
    return (rn, qn), F


def _as_QT(v):
    """a single vector / host matrix (n x k) -> device row-major (n, k)"""
    if is_dev(v):
        return v.reshape(-1, 1) if v.dim() == 1 else v
    v = np.asarray(v, dtype=np.complex128)
    if v.ndim == 1:
        v = v.reshape(-1, 1)
    return torch.from_numpy(np.ascontiguousarray(v)).to("cuda")


class ResidualErrmeasure(Errmeasure):
    """||M(lam) v|| / ||v||   (errmeasure.jl:114,128-130)"""

    def __init__(self, nep):
        self.nep = nep

    def batch(self, lams, QT):
        (rn, qn), _ = _batch_norms(self.nep, lams, QT)
        return rn / qn

    def batch_async(self, lams, QT):
        p = self.nep.resid_norms_async(lams, QT)
        return _PendingErrs(p, lambda rn, qn, F: rn / qn)


class StandardSPMFErrmeasure(Errmeasure):
    """backward error ||M(lam)v|| / (||v|| sum_i ||A_i||_F |f_i(lam)|)   (errmeasure.jl:174-190)"""

    def __init__(self, nep):
        if not isinstance(nep, AbstractSPMF):
            raise TypeError("StandardSPMFErrmeasure needs an AbstractSPMF")
        self.nep = nep
        self.coeffs = np.array(nep.fro_norms())

    def batch(self, lams, QT):
        (rn, qn), F = _batch_norms(self.nep, lams, QT)
        denom = self.coeffs @ np.abs(F)
        return rn / (qn * denom)

    def batch_async(self, lams, QT):
        p = self.nep.resid_norms_async(lams, QT)
        return _PendingErrs(p, lambda rn, qn, F: rn / (qn * (self.coeffs @ np.abs(F))))


class DefaultErrmeasure(Errmeasure):
    """errmeasure.jl:91-101"""

    def __init__(self, nep):
        from .wep import WEP
        spmf = isinstance(nep, AbstractSPMF) and not isinstance(nep, WEP)     # reference: WEP <: NEP, not AbstractSPMF
        self.errm = StandardSPMFErrmeasure(nep) if spmf else ResidualErrmeasure(nep)

    def batch(self, lams, QT):
        return self.errm.batch(lams, QT)

    def batch_async(self, lams, QT):
        return self.errm.batch_async(lams, QT)


def estimate_error(errm, lam, v):
    """estimate_error(E, lam, v)  (errmeasure.jl:128-135,186-190); callables are accepted like the
    reference's `ErrmeasureType = Union{Errmeasure, Function}` (:79)."""
    if callable(errm) and not isinstance(errm, Errmeasure):
        return errm(lam, v)
    return float(errm.batch([lam], _as_QT(v))[0])


def estimate_errors(errm, lams, QT):
    if callable(errm) and not isinstance(errm, Errmeasure):
        Q = QT.cpu_matrix() if hasattr(QT, "cpu_matrix") else QT.cpu().numpy()
        return np.array([errm(l, Q[:, s]) for s, l in enumerate(lams)])
    return errm.batch(list(lams), QT)


def estimate_errors_async(errm, lams, QT):
    """like estimate_errors, but only enqueues the device work; returns an object with ready() / get()"""
    if isinstance(errm, Errmeasure) and hasattr(errm, "batch_async") and hasattr(getattr(errm, "nep", getattr(getattr(errm, "errm", None), "nep", None)), "resid_norms_async"):
        return errm.batch_async(list(lams), QT)
    return _PendingErrs(None, None, value=estimate_errors(errm, lams, QT))
