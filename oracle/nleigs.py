"""Oracle NLEIGS (test infrastructure only): NumPy/SciPy restatement of

  src/method_nleigs.jl:60-377     main loop (dynamic variant, static=false, return_details=false)
  src/method_nleigs.jl:380-396    constructD
  src/method_nleigs.jl:399-518    backslash (continuation-vector solve; full and low-rank branches)
  src/method_nleigs.jl:521-531    in_Sigma
  src/rk_helper/rk_utils.jl:14-128    lejabagby, scgendivdiffs, ratnewtoncoeffsm, evalrat
  src/rk_helper/rk_nep.jl:102-153     get_rk_nep (p, q, BBCC, low-rank factors L, UU, iL)
  src/rk_helper/discretizepolygon.jl, inpolygon.jl
  src/rk_helper/linsolvercache.jl:7-26

Restrictions (documented in DESIGN.md): SPMF-type NEPs only.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from . import neps, solvers


# ------------------------------------------------------------------------------------------------
def det3p(q1x, q1y, q2x, q2y, px, py):
    return (q1x - px) * (q2y - py) - (q2x - px) * (q1y - py)


def inpolygon(px, py, polyx, polyy):
    """Hormann-Agathos point-in-polygon (rk_helper/inpolygon.jl)."""
    if not (np.isfinite(px) and np.isfinite(py)):
        return False
    c = False
    n = len(polyx)
    for idx in range(n):
        q1x, q1y = polyx[idx], polyy[idx]
        q2x, q2y = polyx[(idx + 1) % n], polyy[(idx + 1) % n]
        if q1x == px and q1y == py:
            return True
        if q2y == py:
            if q2x == px:
                return True
            elif q1y == py and (q2x > px) == (q1x < px):
                return True
        if (q1y < py) != (q2y < py):
            if q1x >= px:
                if q2x > px:
                    c = not c
                else:
                    det = det3p(q1x, q1y, q2x, q2y, px, py)
                    if np.isclose(0, det, rtol=np.sqrt(np.finfo(float).eps), atol=0):
                        return True
                    elif (det > 0) == (q2y > q1y):
                        c = not c
            elif q2x > px:
                det = det3p(q1x, q1y, q2x, q2y, px, py)
                if np.isclose(0, det, rtol=np.sqrt(np.finfo(float).eps), atol=0):
                    return True
                elif (det > 0) == (q2y > q1y):
                    c = not c
    return c


def in_Sigma(z, Sigma, tol):
    Sigma = np.asarray(Sigma)
    if len(Sigma) == 2 and np.all(Sigma.imag == 0):
        rx = np.array([Sigma[0].real, Sigma[0].real, Sigma[1].real, Sigma[1].real]); iy = np.array([-tol, tol, tol, -tol])
    else:
        rx = Sigma.real; iy = Sigma.imag
    return np.array([inpolygon(p.real, p.imag, rx, iy) for p in np.atleast_1d(z)], dtype=bool)


def discretizepolygon(z, include_interior_points=False, npts=10000, nptsint=5):
    z = np.asarray(z, dtype=complex)
    if len(z) == 0:
        z = np.array([0j])
    if len(z) == 1:
        zz = list(z[0] + np.exp(2j * np.pi * np.arange(1, npts + 1) / npts))
    elif len(z) == 2:
        zz = list((z[1] - z[0]) / 2 * (np.cos(np.pi * np.arange(npts - 1, -1, -1) / (npts - 1)) + 1) + z[0])
    else:
        z = np.concatenate([z, z[:1]])
        L = np.sum(abs(np.diff(z)))
        ind = 0; alph = 0.0
        zz = [z[0]]
        remL = L / npts
        while len(zz) < npts:
            d = abs(z[ind + 1] - z[ind])
            if (1 - alph) * d < remL:
                ind += 1
                remL -= (1 - alph) * d
                alph = 0.0
            else:
                alph += remL / d
                remL = L / npts
                zz.append(z[ind] + alph * (z[ind + 1] - z[ind]))
    zz = np.concatenate([np.asarray(zz, dtype=complex), z])
    Z = np.zeros(0, dtype=complex)
    if include_interior_points:
        if len(z) == 2:
            xnr = 2 * nptsint
            if xnr % 2 == 0:
                xnr += 1
            xpts = np.linspace(z[0], z[1], xnr)
            return zz, xpts[1::2]
        points = zz if len(z) == 1 else z
        rx, iy = points.real, points.imag
        rmin, rmax = rx.min(), rx.max(); imin, imax = iy.min(), iy.max()
        it = 0
        spacing = (rmax - rmin) / 2.0001 / np.sqrt(nptsint)
        while len(Z) < nptsint:
            it += 1
            if it > 10:
                raise RuntimeError("Failed to find interior polygon points. Polygon too narrow?")
            xnr = int((rmax - rmin) / (2 * spacing)); ynr = int((imax - imin) / (2 * spacing))
            spacing /= np.sqrt(np.sqrt(2))
            if xnr <= 1 or ynr <= 1:
                continue
            xpts = np.linspace(rmin, rmax, xnr)[1::2]
            eps = np.finfo(float).eps
            ypts = np.linspace(imin - eps, imax + eps, ynr)[1::2]
            Z = np.array([x + 1j * y for x in xpts for y in ypts], dtype=complex)
            Z = np.array([p for p in Z if inpolygon(p.real, p.imag, rx, iy)], dtype=complex)
    return zz, Z


def lejabagby(A, B, C, m, keepA=False, forceInf=0):
    """rk_utils.jl:14-46"""
    A = np.asarray(A, dtype=complex); B = np.asarray(B, dtype=float); C = np.asarray(C, dtype=complex)
    a = [A[0]]
    b = [np.inf if forceInf > 0 else B[0]]
    beta = [1.0]
    sA = np.ones(len(A), dtype=complex); sB = np.ones(len(B), dtype=complex); sC = np.ones(len(C), dtype=complex)
    with np.errstate(all="ignore"):
        for j in range(m - 1):
            binv = 1 / b[j]; betainv = 1 / beta[j]
            sA = sA * betainv * (A - a[j]) / (1 - A * binv)
            sB = sB * betainv * (B - a[j]) / (1 - B * binv)
            sC = sC * betainv * (C - a[j]) / (1 - C * binv)
            if keepA:
                a.append(A[j + 1])
            else:
                t = np.where(np.isnan(sA), -np.inf, abs(sA))
                a.append(A[int(np.argmax(t))])
            if forceInf > j + 1:
                b.append(np.inf)
            else:
                t = np.where(np.isnan(sB), np.inf, abs(sB))
                b.append(B[int(np.argmin(t))])
            beta.append(float(np.max(abs(sC))))
            if beta[j + 1] < np.finfo(float).eps:
                beta[j + 1] = 1.0
    return np.array(a, dtype=complex), np.array(b, dtype=float), np.array(beta, dtype=float)


def ratnewtoncoeffsm(fm, sigma, xi, beta):
    """rk_utils.jl:99-119: rational divided differences via a matrix function of H K^{-1}"""
    m = len(sigma) - 1
    sigma = np.asarray(sigma, dtype=complex); xi = np.asarray(xi, dtype=float); beta = np.asarray(beta, dtype=float)
    K = np.diag(np.ones(m + 1)).astype(complex)
    H = np.diag(sigma[:m + 1]).astype(complex)
    with np.errstate(all="ignore"):
        sub = beta[1:m + 1] / xi[:m]
    K[np.arange(1, m + 1), np.arange(m)] = sub
    H[np.arange(1, m + 1), np.arange(m)] = beta[1:m + 1]
    Pd = 1.0 / np.max(abs(K), axis=0)
    K = K * Pd[None, :]; H = H * Pd[None, :]
    M = np.linalg.solve(K.T, H.T).T          # H / K
    return np.asarray(fm(M), dtype=complex)[:, 0] * beta[0]


def evalrat(sigma, xi, beta, z):
    """rk_utils.jl:121-128: nodal rational function at the point z"""
    r = 1.0 / beta[0] + 0j
    with np.errstate(all="ignore"):
        for j in range(len(sigma)):
            r = r * (z - sigma[j]) / (1 - z / xi[j]) / beta[j + 1]
    return r


def ratnewtoncoeffs_scalar(f, sigma, xi, beta):
    """rk_utils.jl:73-93 for a scalar function (evaluated on 1 x 1 matrices): divided differences by differencing;
    the sigma must be distinct"""
    m = len(sigma)
    fv = lambda z: complex(np.asarray(f(np.array([[z]], dtype=complex))).ravel()[0])
    D = np.zeros(m, dtype=complex)
    D[0] = fv(sigma[0]) * beta[0]
    for j in range(1, m):
        Qj = sum(D[k] * evalrat(sigma[:k], xi[:k], beta[:k + 1], sigma[j]) for k in range(j))
        D[j] = (fv(sigma[j]) - Qj) / evalrat(sigma[:j], xi[:j], beta[:j + 1], sigma[j])
    return D


def ratnewtoncoeffs(fun, sigma, xi, beta):
    """rk_utils.jl:73-93 for a MATRIX-valued function (lam -> M(lam), sparse or dense): rational divided differences
    D_0..D_{m-1} by differencing; the sigma must be distinct"""
    m = len(sigma)
    D = [fun(sigma[0]) * beta[0]]
    for j in range(1, m):
        Qj = None
        for k in range(j):
            T = D[k] * evalrat(sigma[:k], xi[:k], beta[:k + 1], sigma[j])
            Qj = T if Qj is None else Qj + T
        D.append((fun(sigma[j]) - Qj) * (1.0 / evalrat(sigma[:j], xi[:j], beta[:j + 1], sigma[j])))
    return D


def _fro(A):
    return float(np.sqrt(abs(A.multiply(A.conj()).sum()))) if sp.issparse(A) else float(np.linalg.norm(A))


def scgendivdiffs(sigma, xi, beta, maxdgr, pff, isfunm=True):
    """rk_utils.jl:56-66"""
    if isfunm:
        return np.vstack([ratnewtoncoeffsm(f, sigma, xi, beta) for f in pff])
    return np.vstack([ratnewtoncoeffs_scalar(f, sigma, xi, beta) for f in pff])


class RKNEP:
    """rk_nep.jl:19-32,102-153"""

    def __init__(self, nep):
        self.nep = nep
        self.is_low_rank = False
        self.r = 0
        self.spmf = hasattr(nep, "get_Av")
        if not self.spmf:                      # RKNEP(::Type{T}, nep::NEP), rk_nep.jl:34-35: no structure known
            self.p, self.q, self.BC = 0, 0, []
            return
        Av = nep.get_Av()
        self.BC = Av
        if isinstance(nep, neps.PEP):
            self.p, self.q = len(Av) - 1, 0
        elif isinstance(nep, neps.SumNEP) and isinstance(nep.nep1, neps.PEP):
            self.p = len(nep.nep1.get_Av()) - 1; self.q = len(nep.nep2.get_Av())
            if self.q > 0 and isinstance(nep.nep2, neps.LowRankFactorizedNEP):       # :128-152
                self.is_low_rank = True
                self.L = nep.nep2.L
                self.UU = sp.csc_matrix(sp.hstack(nep.nep2.U))
                self.r = nep.nep2.rank
                self.iL = np.concatenate([np.full(L.shape[1], i, dtype=int) for i, L in enumerate(self.L)])
                self.Lall = sp.csc_matrix(sp.hstack(self.L))                          # the rows LL / iLr of :141-150, as one matrix
        else:
            self.p, self.q = -1, len(Av)

    def blk(self, j):
        """rows of block j (0-based) of the linearisation vectors: n for j < p, r from there on (method_nleigs.jl:206-211)"""
        n = self.nep.size(1)
        return n if (not self.is_low_rank or j < self.p) else self.r


def constructD(nb, P, sgdd):
    """method_nleigs.jl:380-396"""
    if not P.is_low_rank or nb <= P.p:
        D = None
        for ii in range(P.p + 1 + P.q):
            T = sgdd[ii, nb] * P.BC[ii]
            D = T if D is None else D + T
        return D
    return sp.csc_matrix(sp.hstack([sgdd[P.p + 1 + ii, nb] * P.L[ii] for ii in range(P.q)]))


class LinSolverCache:
    def __init__(self, nep, creator):
        self.nep, self.creator, self.solvers = nep, creator, {}

    def solve(self, sigma, y, add_to_cache):
        key = complex(sigma)
        if key in self.solvers:
            return self.solvers[key].lin_solve(y)
        s = self.creator.create_linsolver(self.nep, sigma)
        if add_to_cache:
            self.solvers[key] = s
        return s.lin_solve(y)


def backslash(wc, P, cache, reusefact, computeD, sigma, k, D, beta, N, xi, expand, kconv, sgdd):
    """method_nleigs.jl:399-518; k is the 1-based iteration counter.  Block j of a vector has P.blk(j) rows."""
    n = P.nep.size(1)
    p = P.p
    lr = P.is_low_rank
    shift = sigma[k]                      # sigma[k+1] in 1-based numbering
    off = np.concatenate([[0], np.cumsum([P.blk(j) for j in range(N + 2)])])
    B = lambda j: slice(off[j], off[j + 1])

    def Dmul(ii, x):
        if computeD:
            # (low rank, p = 2, first step: the reference indexes D[p+1] before it exists, :410 -- a BoundsError there;
            # the matrix is formed on the spot here, which is what its BBCC branch :412 computes)
            Dii = D[ii] if ii < len(D) else constructD(ii, P, sgdd)
            return np.asarray(Dii @ x).ravel()
        acc = np.zeros(n, dtype=complex)
        for j, A in enumerate(P.BC):
            acc += sgdd[j, ii] * (A @ x)
        return acc

    with np.errstate(all="ignore"):
        Bw = np.zeros(len(wc), dtype=complex)
        if lr:                                                                      # :407-415 first block
            Bw[:n] = -Dmul(p, wc[B(p - 1)]) / beta[p]
        for ii in range(1, N + 1):                                                  # :417-436
            if not lr or ii != p:
                Bw[B(ii)] = wc[B(ii - 1)] + beta[ii] / xi[ii - 1] * wc[B(ii)]
            else:
                Bw[B(ii)] = P.UU.conj().T @ wc[B(ii - 1)] + beta[ii] / xi[ii - 1] * wc[B(ii)]
        z = Bw.copy()                                                               # :438-491
        nu = beta[1] * (1 - shift / xi[0])
        z[B(1)] = 1 / nu * z[B(1)]
        for ii in range(1, N + 1):
            if not lr or ii < p:
                z[:n] -= Dmul(ii, z[B(ii)])
            elif ii == p and p >= 2:
                # Not in the reference, which skips ii == p altogether (:455-463).  The first block row of the low-rank
                # pencil holds D_{p-1} - sigma_{p-1} D_p / beta_p in A and -D_p / beta_p in B at block p-1 (that B entry is
                # the first block of Bw above), so eliminating it leaves (shift - sigma_{p-1})/beta_p D_p z_{p-1} on the
                # right-hand side.  For p = 1 this is D_1 z_0 = 0 -- every low-rank test of the reference has p = 1; for
                # p = 2 the Ritz values do not approach eigenvalues of the NEP without it (tests/test_oracle_kat.py).
                z[:n] -= Dmul(p, (shift - sigma[p - 1]) / (beta[p] * (1 - shift / xi[p - 1])) * z[B(p - 1)])
            elif ii > p:
                if computeD:
                    z[:n] -= np.asarray(D[ii] @ z[B(ii)]).ravel()
                else:
                    z[:n] -= P.Lall @ (sgdd[p + 1 + P.iL, ii] * z[B(ii)])           # :464-471
            if ii < N:
                mu = shift - sigma[ii]
                nu = beta[ii + 1] * (1 - shift / xi[ii])
                if not lr or ii != p - 1:
                    z[B(ii + 1)] = 1 / nu * z[B(ii + 1)] + mu / nu * z[B(ii)]
                else:
                    z[B(ii + 1)] = 1 / nu * z[B(ii + 1)] + mu / nu * (P.UU.conj().T @ z[B(ii)])
        w = np.zeros(len(wc), dtype=complex)
        add_to_cache = ((not expand or k > kconv) and reusefact == 1) or reusefact == 2
        w[:n] = cache.solve(shift, z[:n] / beta[0], add_to_cache)
        for ii in range(1, N + 1):                                                  # :498-515
            mu = shift - sigma[ii - 1]
            nu = beta[ii] * (1 - shift / xi[ii - 1])
            if not lr or ii != p:
                w[B(ii)] = mu / nu * w[B(ii - 1)] + 1 / nu * Bw[B(ii)]
            else:
                w[B(ii)] = mu / nu * (P.UU.conj().T @ w[B(ii - 1)]) + 1 / nu * Bw[B(ii)]
    return w


class NleigsSolutionDetails:
    """method_nleigs.jl:538-561"""

    def __init__(self, Lam, Res, sigma, xi, beta, nrmD, kconv):
        self.Lam, self.Res, self.sigma, self.xi, self.beta, self.nrmD, self.kconv = Lam, Res, sigma, xi, beta, nrmD, kconv


def nleigs(nep, Sigma, Xi=(np.inf,), maxdgr=100, minit=20, maxit=200, linsolvercreator=None, tol=1e-10,
           tollin=None, v=None, errmeasure=None, leja=1, nodes=(), reusefact=1, blksize=20, check_error_every=5,
           info=None, static=False, return_details=False, isfunm=True):
    """method_nleigs.jl:60-377 (isfunm=true; dynamic and static variants, optional solution details).
    With return_details=True the return value is (lam, X, res, NleigsSolutionDetails) as in the reference."""
    import warnings
    eps = np.finfo(float).eps
    if tollin is None:
        tollin = max(tol / 10, 100 * eps)
    if linsolvercreator is None:
        linsolvercreator = solvers.DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = solvers.ResidualErrmeasure(nep)
    Sigma = np.asarray(Sigma, dtype=complex); Xi = np.asarray(Xi, dtype=float)
    nodes = np.asarray(nodes, dtype=complex)
    P = RKNEP(nep)
    n = nep.size(1)
    if n == 1:
        maxdgr = maxit + 1
    computeD = n <= 400 or not P.spmf           # `!P.spmf || computeD` in every branch of the reference
    cache = LinSolverCache(nep, linsolvercreator)
    v = np.array(v, dtype=complex)

    if leja == 0:
        if len(nodes) == 0:
            raise ValueError("Interpolation nodes must be provided via 'nodes' when no Leja-Bagby points ('leja' == 0) are used.")
        gamma, _ = discretizepolygon(Sigma)
        max_count = (maxit + maxdgr + 2) if static else max(maxit, maxdgr) + 2
        sigma = np.tile(nodes, int(np.ceil(max_count / len(nodes))))
        _, xi, beta = lejabagby(sigma[:maxdgr + 2], Xi, gamma, maxdgr + 2, True, P.p)
    elif leja == 1:
        if len(nodes) == 0:
            gamma, nodes = discretizepolygon(Sigma, True)
        else:
            gamma, _ = discretizepolygon(Sigma)
        nodes = np.tile(nodes, int(np.ceil((maxit + 1) / len(nodes))))
        sigma, xi, beta = lejabagby(gamma, Xi, gamma, maxdgr + 2, False, P.p)
    else:
        gamma, _ = discretizepolygon(Sigma)
        max_count = (maxit + maxdgr + 2) if static else max(maxit, maxdgr) + 2
        sigma, xi, beta = lejabagby(gamma, Xi, gamma, max_count, False, P.p)
    sigma = np.array(sigma, dtype=complex); xi = np.array(xi, dtype=float); beta = np.array(beta, dtype=float)
    xi[maxdgr + 1] = np.nan

    rng_ = slice(0, maxdgr + 2)
    if not isfunm and len(np.unique(sigma)) != len(sigma):                          # :142-145
        raise ValueError("All interpolation nodes must be distinct when no matrix functions are used for computing "
                         "the generalized divided differences.")
    if not P.spmf:                                                                  # :149-153
        if len(np.unique(sigma[rng_])) != len(sigma[rng_]):
            raise ValueError("All interpolation nodes must be distinct when no matrix functions are used for computing "
                             "the generalized divided differences.")
        Dall = ratnewtoncoeffs(lambda lam_: nep.compute_Mder(lam_), sigma[rng_], xi[rng_], beta[rng_])
        D = [Dall[0]]
        nrmD = [_fro(Dall[0])]
        sgdd = None
    else:
        sgdd = scgendivdiffs(sigma[rng_], xi[rng_], beta[rng_], maxdgr, nep.get_fv(), isfunm)
        D = []
        if computeD:
            D.append(constructD(0, P, sgdd))
        nrmD = [float(np.max(abs(sgdd[:, 0])))]
    if not np.isfinite(nrmD[0]):
        raise ValueError("The generalized divided differences must be finite.")

    kmax = maxit + maxdgr if static else maxit                                      # :176
    vrows = (min(kmax, maxdgr + 1) + 2) * n if static else (kmax + 2) * n         # static: N <= maxdgr+1 blocks, zero padded
    V = np.zeros((vrows, maxit + 2), dtype=complex, order="F")
    H = np.zeros((maxit + 2, maxit + 1), dtype=complex); K = np.zeros((maxit + 2, maxit + 1), dtype=complex)
    Lam = np.zeros((maxit + 1, maxit + 1), dtype=complex); Res = np.zeros((maxit + 1, maxit + 1))
    v = cache.solve(sigma[0], v / np.linalg.norm(v), reusefact == 2)
    V[:n, 0] = v / np.linalg.norm(v)
    expand = True
    kconv = np.iinfo(np.int64).max // 2
    kn = n; l = 0; N = 0; nbconv = 0; nblamin = 0
    lam = np.zeros(0, dtype=complex); X = np.zeros((n, 0), dtype=complex); res = np.zeros(0); conv = np.zeros(0, dtype=bool)
    k = 1
    nfact = 0
    while k <= kmax:
        if expand:
            kn += P.blk(k)                                                          # :206-211
            if not P.spmf:
                D.append(Dall[k])
            elif computeD:
                D.append(constructD(k, P, sgdd))
            N += 1
            nrmD.append(_fro(D[k]) if not P.spmf else float(np.max(abs(sgdd[:, k]))))
            if not np.isfinite(nrmD[k]):
                raise ValueError("The generalized divided differences must be finite.")
            if n > 1 and k >= 5 and k < kconv:
                if sum(nrmD[k - 4:k + 1]) < 5 * tollin:
                    kconv = k - 1
                    if static:
                        kmax = maxit + kconv                                        # :236-238
                    expand = False
                    if leja == 1:
                        if len(sigma) < kmax + 1:
                            sigma = np.concatenate([sigma, np.zeros(kmax + 1 - len(sigma), dtype=complex)])
                        sigma[k:kmax + 1] = nodes[:kmax - k + 1]
                    if computeD:
                        D = D[:k]
                    xi = xi[:k]; beta = beta[:k]; nrmD = nrmD[:k]
                    if static:
                        kn -= P.blk(k)                                              # :250-257 (V is zero padded already)
                    N -= 1
                elif k == maxdgr + 1:
                    kconv = k
                    expand = False
                    warnings.warn("NLEIGS: Linearization not converged after %d iterations" % maxdgr)
                    if leja == 1:
                        if len(sigma) < kmax + 1:
                            sigma = np.concatenate([sigma, np.zeros(kmax + 1 - len(sigma), dtype=complex)])
                        sigma[k:kmax + 1] = nodes[:kmax - k + 1]
                    N -= 1
        l = k - N if static else k                                                 # :283
        if not static or not expand:                                                # :285-298
            t = np.zeros(l); t[l - 1] = 1
            wc = V[:kn, l - 1].copy()
            w = backslash(wc, P, cache, reusefact, computeD, sigma, k, D, beta, N, xi, expand, kconv, sgdd)
            H[l, l - 1] = solvers.dgks(V[:kn, :l], w, H[:l, l - 1])
            K[:l, l - 1] = H[:l, l - 1] * sigma[k] + t
            K[l, l - 1] = H[l, l - 1] * sigma[k]
            V[:kn, l] = w

        def check_convergence(all_):
            nonlocal lam, X, res, conv, nbconv, nblamin
            lambda_, S = sla.eig(K[:l, :l], H[:l, :l])
            if not all_:
                lamin = in_Sigma(lambda_, Sigma, tol)
                ilam = np.nonzero(lamin)[0]
                lam = lambda_[ilam]
            else:                                                                   # :309-313
                ilam = np.nonzero(np.isfinite(lambda_))[0]
                lam = lambda_[ilam]
                lamin = in_Sigma(lam, Sigma, tol)
            nblamin = int(np.sum(lamin))
            S = S.copy()
            for i in ilam:
                S[:, i] /= np.linalg.norm(H[:l + 1, :l] @ S[:, i])
            X = V[:n, :l + 1] @ (H[:l + 1, :l] @ S[:, ilam])
            if X.shape[1]:
                X = X / np.linalg.norm(X, axis=0)[None, :]
            res = np.array([errmeasure(lam[i], X[:, i]) for i in range(len(lam))])
            conv = abs(res) < tol
            if all_:                                                                # :328-336
                resall = np.full(l, np.nan)
                resall[ilam] = res
                si = sorted(range(l), key=lambda i: (abs(lambda_[i]), np.angle(lambda_[i])))
                Res[:l, l - 1] = resall[si]
                Lam[:l, l - 1] = lambda_[si]
                conv = conv & lamin
            nbconv = int(np.sum(conv)) if len(conv) else 0
            if info is not None and "_history" in info:                             # diagnostic hook: one row per check
                info["_history"].append((k, l, lam.copy(), res.copy(), [float(np.linalg.norm(S[:, i])) for i in ilam]))

        if not return_details and (
                (not expand and k >= N + minit and (k - (N + minit)) % check_error_every == 0) or
                (k >= kconv + minit and (k - (kconv + minit)) % check_error_every == 0) or k == kmax):
            check_convergence(False)
        elif return_details and (not static or not expand):
            check_convergence(True)
        if ((not expand and k >= N + minit) or k >= kconv + minit) and nblamin == nbconv \
                and not (info is not None and info.get("_nobreak")):               # (_nobreak: diagnostic hook only)
            break
        k += 1
    if info is not None:
        info.update(kconv=kconv, N=N, k=min(k, kmax), nfact=len(cache.solvers), nrmD=nrmD, nblamin=nblamin)
        if info.get("_debug"):          # diagnostic hook (scripts/diag): the Krylov relation of the final run
            info.update(V=V, H=H, K=K, sigma=sigma, xi=xi, beta=beta, kn=kn, l=l, P=P, sgdd=sgdd, D=D, cache=cache,
                        computeD=computeD)
    if return_details:                                                              # :363-374
        kk = min(k, kmax)
        if expand:
            xi = xi[:kk]; beta = beta[:kk]; nrmD = nrmD[:kk]
            warnings.warn("NLEIGS: Linearization not converged after %d iterations" % maxdgr)
        details = NleigsSolutionDetails(Lam[:l, :l], Res[:l, :l], sigma[:kk], xi, beta, nrmD, kconv)
        return lam[conv], X[:, conv], res[conv], details
    return lam[conv], X[:, conv], res[conv]
