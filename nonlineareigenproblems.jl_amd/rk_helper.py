"""Rational-Krylov host utilities for nleigs (tiny, setup only): src/rk_helper/rk_utils.jl:14-128 (lejabagby,
scgendivdiffs, ratnewtoncoeffsm), src/rk_helper/discretizepolygon.jl, src/rk_helper/inpolygon.jl and the
structure discovery of src/rk_helper/rk_nep.jl:102-153 (p, q; the stacked matrix BBCC is the device's
stacked CSR).  Pure host arithmetic on O(maxdgr) points -- not a kernel."""
import numpy as np

import scipy.sparse as sp

from .nep import PEP, SumNEP, LowRankFactorizedNEP


def _det3p(q1x, q1y, q2x, q2y, px, py):
    return (q1x - px) * (q2y - py) - (q2x - px) * (q1y - py)


def inpolygon(px, py, polyx, polyy):
    """point-in-polygon with the boundary counted as inside (rk_helper/inpolygon.jl, Hormann-Agathos): the per-edge tests of
    the reference, evaluated for all edges at once -- an eigenvalue check tests ~100 Ritz values against the 1573-gon of
    the gun target set, 1.6e5 edge tests per check"""
    if not (np.isfinite(px) and np.isfinite(py)):
        return False
    q1x = np.asarray(polyx, dtype=float); q1y = np.asarray(polyy, dtype=float)
    q2x = np.roll(q1x, -1); q2y = np.roll(q1y, -1)
    if np.any((q1x == px) & (q1y == py)):
        return True
    on_h = (q2y == py) & ((q2x == px) | ((q1y == py) & ((q2x > px) == (q1x < px))))
    if np.any(on_h):
        return True
    cross = (q1y < py) != (q2y < py)
    sure = cross & (q1x >= px) & (q2x > px)
    maybe = cross & ~sure & ((q1x >= px) | (q2x > px))
    det = _det3p(q1x[maybe], q1y[maybe], q2x[maybe], q2y[maybe], px, py)
    if np.any(det == 0.0):        # isapprox(0, det) with default rtol and atol = 0 holds only for det == 0
        return True
    flips = int(np.count_nonzero(sure)) + int(np.count_nonzero((det > 0) == (q2y[maybe] > q1y[maybe])))
    return bool(flips & 1)


def in_Sigma(z, Sigma, tol):
    """src/method_nleigs.jl:521-531"""
    Sigma = np.asarray(Sigma)
    if len(Sigma) == 2 and np.all(Sigma.imag == 0):
        rx = np.array([Sigma[0].real, Sigma[0].real, Sigma[1].real, Sigma[1].real]); iy = np.array([-tol, tol, tol, -tol])
    else:
        rx, iy = Sigma.real, Sigma.imag
    return np.array([inpolygon(p.real, p.imag, rx, iy) for p in np.atleast_1d(z)], dtype=bool)


def _discretize_native(zv, L, npts):
    """the boundary walk through the library (host code); None when switched off or not applicable"""
    import os
    if os.environ.get("NEP_RK_NATIVE", "1") == "0" or npts < 1 or len(zv) < 3 or not np.isfinite(L) or not L > 0:
        return None
    from ._lib import lib, hptr
    zc = np.ascontiguousarray(zv, dtype=np.complex128)
    out = np.empty(npts, dtype=np.complex128)
    out[0] = L                                   # (L = np.sum(...) travels in: NumPy's pairwise sum is not a running sum)
    if lib.nep_discretize_polygon(len(zc), hptr(zc), int(npts), hptr(out)) != 0:
        return None
    return out


def discretizepolygon(z, include_interior_points=False, npts=10000, nptsint=5):
    z = np.asarray(z, dtype=complex)
    if len(z) == 0:
        z = np.array([0j])
    if len(z) == 1:
        zz = z[0] + np.exp(2j * np.pi * np.arange(1, npts + 1) / npts)
    elif len(z) == 2:
        zz = (z[1] - z[0]) / 2 * (np.cos(np.pi * np.arange(npts - 1, -1, -1) / (npts - 1)) + 1) + z[0]
    else:
        z = np.concatenate([z, z[:1]])
        L = np.sum(abs(np.diff(z)))
        # the point-by-point walk of discretizepolygon.jl (10 000 interpreted iterations: 8 ms of a nleigs call) in the library's
        # host code, operation by operation (nep_discretize_polygon; NEP_RK_NATIVE=0 keeps the interpreted walk -- same bits,
        # tests/test_host_logic.py::test_discretizepolygon_native_walk_equals_the_interpreted_one)
        zz = _discretize_native(z[:-1], L, npts)
        if zz is None:
            ind = 0; alph = 0.0
            pts = [z[0]]
            remL = L / npts
            while len(pts) < npts:
                d = abs(z[ind + 1] - z[ind])
                if (1 - alph) * d < remL:
                    ind += 1
                    remL -= (1 - alph) * d
                    alph = 0.0
                else:
                    alph += remL / d
                    remL = L / npts
                    pts.append(z[ind] + alph * (z[ind + 1] - z[ind]))
            zz = np.asarray(pts, dtype=complex)
    zz = np.concatenate([zz, z])
    Z = np.zeros(0, dtype=complex)
    if include_interior_points:
        if len(z) == 2:
            xnr = 2 * nptsint
            xnr += (xnr % 2 == 0)
            return zz, np.linspace(z[0], z[1], xnr)[1::2]
        points = zz if len(z) == 1 else z
        rx, iy = points.real, points.imag
        rmin, rmax, imin, imax = rx.min(), rx.max(), iy.min(), iy.max()
        it = 0
        spacing = (rmax - rmin) / 2.0001 / np.sqrt(nptsint)
        eps = np.finfo(float).eps
        while len(Z) < nptsint:
            it += 1
            if it > 10:
                raise RuntimeError("Failed to find interior polygon points. Polygon too narrow? (Note that intervals "
                                   "should be given by their two endpoints only.)")
            xnr = int((rmax - rmin) / (2 * spacing)); ynr = int((imax - imin) / (2 * spacing))
            spacing /= np.sqrt(np.sqrt(2))
            if xnr <= 1 or ynr <= 1:
                continue
            xpts = np.linspace(rmin, rmax, xnr)[1::2]
            ypts = np.linspace(imin - eps, imax + eps, ynr)[1::2]
            Z = np.array([x + 1j * y for x in xpts for y in ypts if inpolygon(x, y, rx, iy)], dtype=complex)
    return zz, Z


def lejabagby(A, B, C, m, keepA=False, forceInf=0):
    A = np.asarray(A, dtype=complex); B = np.asarray(B, dtype=float); C = np.asarray(C, dtype=complex)
    a = [A[0]]; b = [np.inf if forceInf > 0 else B[0]]; beta = [1.0]
    sA = np.ones(len(A), dtype=complex); sB = np.ones(len(B), dtype=complex); sC = np.ones(len(C), dtype=complex)
    # (the three updates  s <- s * betainv * (X - a_j) / (1 - X * binv)  as the same ufunc calls in the same order, into two work
    # arrays per sequence instead of five fresh 10 000-element temporaries per step: identical values, a quarter less time)
    def step_(s_, X, aj, binv, betainv, w1, w2):
        np.multiply(s_, betainv, out=w1)
        np.subtract(X, aj, out=w2)
        np.multiply(w1, w2, out=w1)
        np.multiply(X, binv, out=w2)
        np.subtract(1, w2, out=w2)
        np.divide(w1, w2, out=s_)
    Bc = B.astype(complex)
    wk = [(np.empty(len(X), dtype=complex), np.empty(len(X), dtype=complex)) for X in (A, B, C)]
    with np.errstate(all="ignore"):
        for j in range(m - 1):
            binv = 1 / b[j]; betainv = 1 / beta[j]
            step_(sA, A, a[j], binv, betainv, *wk[0])
            step_(sB, Bc, a[j], binv, betainv, *wk[1])
            step_(sC, C, a[j], binv, betainv, *wk[2])
            a.append(A[j + 1] if keepA else A[int(np.argmax(np.where(np.isnan(sA), -np.inf, abs(sA))))])
            b.append(np.inf if forceInf > j + 1 else B[int(np.argmin(np.where(np.isnan(sB), np.inf, abs(sB))))])
            beta.append(float(np.max(abs(sC))))
            if beta[j + 1] < np.finfo(float).eps:
                beta[j + 1] = 1.0
    return np.array(a, dtype=complex), np.array(b, dtype=float), np.array(beta, dtype=float)


def ratnewtoncoeffsm(fun, sigma, xi, beta):
    """scalar generalized divided differences through the matrix function of H K^{-1} (rk_utils.jl:99-119);
    `fun` is a funcs.ScalarFun (its matfun is used)"""
    m = len(sigma) - 1
    K = np.eye(m + 1, dtype=complex)
    H = np.diag(np.asarray(sigma[:m + 1], dtype=complex))
    with np.errstate(all="ignore"):
        K[np.arange(1, m + 1), np.arange(m)] = np.asarray(beta[1:m + 1]) / np.asarray(xi[:m])
    H[np.arange(1, m + 1), np.arange(m)] = beta[1:m + 1]
    Pd = 1.0 / np.max(abs(K), axis=0)
    K = K * Pd[None, :]; H = H * Pd[None, :]
    M = np.linalg.solve(K.T, H.T).T
    return np.asarray(fun.matfun(M), dtype=complex)[:, 0] * beta[0]


def evalrat(sigma, xi, beta, z):
    """nodal rational function at the point z (rk_utils.jl:121-128)"""
    r = 1.0 / beta[0] + 0j
    with np.errstate(all="ignore"):
        for j in range(len(sigma)):
            r = r * (z - sigma[j]) / (1 - z / xi[j]) / beta[j + 1]
    return r


def ratnewtoncoeffs_scalar(fun, sigma, xi, beta):
    """rational divided differences of a scalar function by differencing (rk_utils.jl:73-93); distinct sigma required.
    `fun` is a funcs.ScalarFun (its value derivs(z, 1)[0] is used)"""
    m = len(sigma)
    fv = lambda z: complex(fun.derivs(complex(z), 1)[0])
    D = np.zeros(m, dtype=complex)
    D[0] = fv(sigma[0]) * beta[0]
    for j in range(1, m):
        Qj = sum(D[k] * evalrat(sigma[:k], xi[:k], beta[:k + 1], sigma[j]) for k in range(j))
        D[j] = (fv(sigma[j]) - Qj) / evalrat(sigma[:j], xi[:j], beta[:j + 1], sigma[j])
    return D


def ratnewtoncoeffs(fun, sigma, xi, beta):
    """src/rk_helper/rk_utils.jl:73-93 for a matrix-valued function lam -> M(lam) (SciPy sparse): rational divided
    differences D_0..D_{m-1} by differencing (host: m evaluations of the user's function + sparse linear combinations);
    the sigma must be distinct"""
    m = len(sigma)
    D = [fun(sigma[0]) * beta[0]]
    for j in range(1, m):
        Qj = None
        for k in range(j):
            T = D[k] * evalrat(sigma[:k], xi[:k], beta[:k + 1], sigma[j])
            Qj = T if Qj is None else Qj + T
        D.append((fun(sigma[j]) - Qj) * (1.0 / evalrat(sigma[:j], xi[:j], beta[:j + 1], sigma[j])))
    return D


def scgendivdiffs(sigma, xi, beta, pff, isfunm=True):
    """rk_utils.jl:56-66"""
    if isfunm:
        return np.vstack([ratnewtoncoeffsm(f, sigma, xi, beta) for f in pff])
    return np.vstack([ratnewtoncoeffs_scalar(f, sigma, xi, beta) for f in pff])


def rk_structure(nep):
    """(p, q) of rk_nep.jl:102-125: degree of the polynomial part and number of nonlinear terms"""
    Av = nep.get_Av()
    if isinstance(nep, PEP):
        return len(Av) - 1, 0
    if isinstance(nep, SumNEP) and isinstance(nep.nep1, PEP):
        return len(nep.nep1.get_Av()) - 1, len(nep.nep2.get_Av())
    return -1, len(Av)


class LowRankStructure:
    """rk_nep.jl:128-152: the factors of a PEP + LowRankFactorizedNEP problem as two device operators --
    UUH = [U_1 ... U_q]^H (r x n) and Lall = [L_1 ... L_q] (n x r; the rows LL / iLr of the reference) -- and iL, the
    term index of every factor column."""

    def __init__(self, nep):
        from .nep import DeviceCSR
        lr = nep.nep2
        self.r = lr.rank
        UU = sp.hstack(lr.U).tocsc()
        self.UUH = DeviceCSR(sp.csr_matrix(UU.conj().T))
        self.Lall = DeviceCSR(sp.hstack(lr.L).tocsr())
        self.iL = np.concatenate([np.full(L.shape[1], i, dtype=int) for i, L in enumerate(lr.L)])


def low_rank_structure(nep):
    """None unless nep = PEP + LowRankFactorizedNEP with at least one low-rank term (rk_nep.jl:124-126)"""
    if isinstance(nep, SumNEP) and isinstance(nep.nep1, PEP) and isinstance(nep.nep2, LowRankFactorizedNEP) \
            and len(nep.nep2.get_Av()) > 0 and nep.nep2.rank > 0:
        return LowRankStructure(nep)
    return None
