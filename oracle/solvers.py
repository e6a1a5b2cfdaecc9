"""Oracle solvers (test infrastructure only): NumPy/SciPy restatement of

  src/errmeasure.jl:91-190                 Default/Residual/StandardSPMF error measures
  src/LinSolvers.jl:109-159                FactorizeLinSolver, BackslashLinSolver, lin_solve
  src/LinSolverCreators.jl:62-122          FactorizeLinSolverCreator (factorization reuse)
  IterativeSolvers 0.9.2 (Manifest.toml:55-59) orthogonalize_and_normalize! DGKS/CGS/MGS
       -- third-party, source not under /root/reference; published algorithm restated
  src/method_iar.jl:47-184                 iar
  src/method_tiar.jl:53-257                tiar
  src/method_newton.jl:142-226,380-445,598-609   resinv, quasinewton, armijo_rule
  src/compute_rf_wrapper.jl:25-54          compute_rf (ScalarNewtonInnerSolver)
  src/method_beyncontour.jl:49-185         contour_beyn
  src/method_contour_common.jl:61-94       integrate_interval (MatrixTrapezoidal)

LU: SciPy's bundled SuperLU stands in for UMFPACK (SuiteSparse 5.10.1), which is not
available in this image; differences appear at the 1e-13 level.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = np.finfo(float).eps


class NoConvergenceException(Exception):
    """NEPCore.jl:324-336."""

    def __init__(self, lam, v, errmeasure, msg):
        super().__init__(msg)
        self.lam, self.v, self.errmeasure, self.msg = lam, v, errmeasure, msg


class LostOrthogonalityException(Exception):
    """NEPCore.jl:350."""


# ----------------------------------------------------------------------------------
# error measures
class ResidualErrmeasure:
    def __init__(self, nep):
        self.nep = nep

    def __call__(self, lam, v):
        return np.linalg.norm(self.nep.compute_Mlincomb(lam, v)) / np.linalg.norm(v)


def _fro(A):
    return sla.norm(A.toarray() if sp.issparse(A) and A.shape[0] <= 64 else A.data) \
        if sp.issparse(A) else np.linalg.norm(A)


class StandardSPMFErrmeasure:
    """errmeasure.jl:174-190 (Frobenius norms of the A_i)."""

    def __init__(self, nep):
        self.nep = nep
        self.coeffs = [float(np.linalg.norm(A.data)) if sp.issparse(A) else float(np.linalg.norm(A))
                       for A in nep.get_Av()]

    def __call__(self, lam, v):
        fv = self.nep.get_fv()
        denom = sum(c * abs(f(lam)) for c, f in zip(self.coeffs, fv))
        return np.linalg.norm(self.nep.compute_Mlincomb(lam, v)) / (np.linalg.norm(v) * denom)


def DefaultErrmeasure(nep):
    """errmeasure.jl:91-101."""
    if hasattr(nep, "get_Av"):
        return StandardSPMFErrmeasure(nep)
    return ResidualErrmeasure(nep)


# ----------------------------------------------------------------------------------
# linear solvers
class FactorizeLinSolver:
    """LinSolvers.jl:109-137: factor M(lam) once, solve many."""

    def __init__(self, nep, lam, permc_spec="COLAMD"):
        A = nep.compute_Mder(lam)
        self.sparse = sp.issparse(A)
        if self.sparse:
            self.Afact = spla.splu(sp.csc_matrix(A, dtype=complex), permc_spec=permc_spec)
        else:
            self.Afact = sla.lu_factor(np.asarray(A, dtype=complex))

    def lin_solve(self, b, tol=0):
        b = np.asarray(b, dtype=complex)
        if self.sparse:
            return self.Afact.solve(b)
        return sla.lu_solve(self.Afact, b)


class BackslashLinSolver:
    """LinSolvers.jl:147-159: factor at every solve."""

    def __init__(self, nep, lam):
        self.A = nep.compute_Mder(lam)

    def lin_solve(self, b, tol=0):
        b = np.asarray(b, dtype=complex)
        if sp.issparse(self.A):
            return spla.splu(sp.csc_matrix(self.A, dtype=complex)).solve(b)
        return np.linalg.solve(np.asarray(self.A, dtype=complex), b)


class FactorizeLinSolverCreator:
    """LinSolverCreators.jl:62-122."""

    def __init__(self, max_factorizations=0, permc_spec="COLAMD"):
        self.recycled = {}
        self.max_factorizations = max_factorizations
        self.permc_spec = permc_spec

    def create_linsolver(self, nep, lam):
        if lam in self.recycled:
            return self.recycled[lam]
        s = FactorizeLinSolver(nep, lam, self.permc_spec)
        if len(self.recycled) < self.max_factorizations:
            self.recycled[lam] = s
        return s


class BackslashLinSolverCreator:
    def create_linsolver(self, nep, lam):
        return BackslashLinSolver(nep, lam)


DefaultLinSolverCreator = FactorizeLinSolverCreator


# ----------------------------------------------------------------------------------
# orthogonalisation (IterativeSolvers.orthogonalize_and_normalize!)
def dgks(V, w, h):
    """DGKS: h=V'w; w-=Vh; repeat while ||w|| < ||correction||/sqrt(2). In place on w,h.
    Returns ||w|| before normalisation."""
    h[:] = V.conj().T @ w
    w -= V @ h
    nrm = np.linalg.norm(w)
    eta = 1.0 / np.sqrt(2.0)
    projection_size = np.linalg.norm(h)
    while nrm < eta * projection_size:
        correction = V.conj().T @ w
        projection_size = np.linalg.norm(correction)
        w -= V @ correction
        h += correction
        nrm = np.linalg.norm(w)
    w *= 1.0 / nrm
    return nrm


def cgs(V, w, h):
    h[:] = V.conj().T @ w
    w -= V @ h
    nrm = np.linalg.norm(w)
    w *= 1.0 / nrm
    return nrm


def mgs(V, w, h):
    for i in range(V.shape[1]):
        h[i] = np.vdot(V[:, i], w)
        w -= h[i] * V[:, i]
    nrm = np.linalg.norm(w)
    w *= 1.0 / nrm
    return nrm


# ----------------------------------------------------------------------------------
def _eig(H):
    D, Z = sla.eig(H)
    return D, Z


def iar(nep, orthmethod=dgks, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6,
        errmeasure=None, sigma=0.0, gamma=1.0, v=None, check_error_every=1, errhist=None,
        timers=None, proj_solve=False, inner_solver_method=None):
    """method_iar.jl:47-184.  F-ordered V so that reshape(...) follows Julia's column-major semantics."""
    import time
    n = nep.size(1); m = maxit
    sigma = complex(sigma)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    v = np.array(v, dtype=complex)
    V = np.zeros((n * (m + 1), m + 1), dtype=complex, order="F")
    H = np.zeros((m + 1, m), dtype=complex)
    y = np.zeros((n, m + 1), dtype=complex, order="F")
    alpha = (complex(gamma) ** np.arange(m + 1)).astype(complex); alpha[0] = 0
    M0inv = linsolvercreator.create_linsolver(nep, sigma)
    err = np.full((m, m), np.nan)
    lam = np.zeros(m + 1, dtype=complex); Q = np.zeros((n, m + 1), dtype=complex)
    V[:n, 0] = v / np.linalg.norm(v)
    k = 1; conv_eig = 0
    tm = timers if timers is not None else {}
    for key in ("mlincomb", "solve", "orth", "ritz", "resid"):
        tm.setdefault(key, 0.0)
    while k <= m and conv_eig < neigs:
        VV = V[:n * (k + 1), :k]
        vv = V[:n * (k + 1), k]
        y[:, 1:k + 1] = VV[:n * k, k - 1].reshape((n, k), order="F")
        y[:, 1:k + 1] /= np.arange(1, k + 1)[None, :]
        t0 = time.perf_counter()
        y[:, 0] = nep.compute_Mlincomb(sigma, y[:, :k + 1], alpha[:k + 1])
        t1 = time.perf_counter()
        y[:, 0] = -M0inv.lin_solve(y[:, 0])
        t2 = time.perf_counter()
        vv[:] = y[:, :k + 1].reshape((k + 1) * n, order="F")
        H[k, k - 1] = orthmethod(VV, vv, H[:k, k - 1])
        t3 = time.perf_counter()
        tm["mlincomb"] += t1 - t0; tm["solve"] += t2 - t1; tm["orth"] += t3 - t2
        if (k % check_error_every == 0) or (k == m):
            D, Z = _eig(H[:k, :k])
            Q = V[:n, :k] @ Z
            lam = sigma + gamma / D
            if proj_solve:                                    # method_iar.jl:118-131
                QQ, RR = np.linalg.qr(V[:n, :k])
                pnep = Proj_SPMF_NEP(nep)
                pnep.set_projectmatrices(QQ, QQ)
                lam, Qproj = inner_solve(inner_solver_method, pnep, V=RR @ Z, lamv=lam.copy(), neigs=k,
                                         sigma=np.mean(lam))
                Q = QQ @ Qproj
            t4 = time.perf_counter()
            conv_eig = 0
            ne = min(len(lam), m)
            err[k - 1, :ne] = [errmeasure(lam[s], Q[:, s]) for s in range(ne)]
            t5 = time.perf_counter()
            tm["ritz"] += t4 - t3; tm["resid"] += t5 - t4
            if errhist is not None:
                errhist.append(np.sort(err[k - 1, :ne]).copy())
            conv_eig = int(np.sum(err[k - 1, :ne] < tol))
            idx = np.argsort(err[k - 1, :ne], kind="stable")
            err[k - 1, :ne] = err[k - 1, idx]
            if k == m or conv_eig >= neigs:
                nrof = int(min(len(lam), neigs))
                lam = lam[idx[:nrof]]
                Q = Q[:, idx[:len(lam)]]
        k += 1
    k -= 1
    if conv_eig < neigs and neigs != np.inf:
        raise NoConvergenceException(lam, Q, err[k - 1, :len(lam)],
                                     "Number of iterations exceeded. maxit=%d." % maxit)
    nc = min(len(lam), conv_eig)
    return lam[:nc], Q[:, :nc], V[:, :k]


def tiar(nep, orthmethod=dgks, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6,
         errmeasure=None, sigma=0.0, gamma=1.0, v=None, check_error_every=1, errhist=None, proj_solve=False,
         inner_solver_method=None):
    """method_tiar.jl:53-257. Note the Julia aliases f=g, ff=f
    (:147,:164): updates of f also change g; g is rebuilt every step."""
    n = nep.size(1); m = maxit
    sigma = complex(sigma)
    if n < m:
        raise LostOrthogonalityException("Loss of orthogonality in the matrix Z. The problem size "
                                         "is too small, use iar instead.")
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    v = np.array(v, dtype=complex)
    a = np.zeros((m + 1, m + 1, m + 1), dtype=complex)
    Z = np.zeros((n, m + 1), dtype=complex, order="F")
    t = np.zeros(m + 1, dtype=complex)
    g = np.zeros((m + 1, m + 1), dtype=complex)
    H = np.zeros((m + 1, m), dtype=complex)
    y = np.zeros((n, m + 1), dtype=complex, order="F")
    alpha = (complex(gamma) ** np.arange(m + 1)).astype(complex); alpha[0] = 0
    M0inv = linsolvercreator.create_linsolver(nep, sigma)
    err = np.full((m + 1, m + 4), np.nan)
    lam = np.zeros(m + 1, dtype=complex); Q = np.zeros((n, m + 1), dtype=complex)
    Z[:, 0] = v / np.linalg.norm(v)
    a[0, 0, 0] = 1
    conv_eig_hist = np.zeros(m + 1, dtype=int)
    k = 1; conv_eig = 0
    while k <= m and conv_eig < neigs:
        y[:, 1:k + 1] = Z[:, :k] @ a[:k, k - 1, :k].T
        y[:, 1:k + 1] /= np.arange(1, k + 1)[None, :]
        y[:, 0] = nep.compute_Mlincomb(sigma, y[:, :k + 1], alpha[:k + 1])
        y[:, 0] = -M0inv.lin_solve(y[:, 0])
        Z[:, k] = y[:, 0]
        t[k] = orthmethod(Z[:, :k], Z[:, k], t[:k])
        # G
        for l in range(k + 1):
            for i in range(1, k + 1):
                g[i, l] = a[i - 1, k - 1, l] / i
            g[0, l] = t[l]
        # h
        h = np.zeros(m + 1, dtype=complex)
        for l in range(k):
            h[:k] += a[:k, :k, l].conj().T @ g[:k, l]
        f = g  # alias, as in Julia
        for l in range(k):
            f[:k + 1, l] -= a[:k + 1, :k, l] @ h[:k]
        hh = np.zeros(m + 1, dtype=complex)
        for l in range(k):
            hh[:k] += a[:k, :k, l].conj().T @ f[:k, l]
        ff = f
        for l in range(k):
            ff[:k + 1, l] -= a[:k + 1, :k, l] @ hh[:k]
        h = h + hh; f = ff
        beta = np.linalg.norm(f[:k + 1, :k + 1])
        H[:k, k - 1] = h[:k]; H[k, k - 1] = beta
        a[:k + 1, k, :k + 1] = f[:k + 1, :k + 1] / beta
        if (k % check_error_every == 0) or (k == m):
            D, W = _eig(H[:k, :k])
            VV = Z[:, :k] @ a[0, :k, :k].T
            Q = VV @ W
            lam = sigma + gamma / D
            if proj_solve:                                    # method_tiar.jl:192-207
                pnep = Proj_SPMF_NEP(nep)
                pnep.set_projectmatrices(Z[:, :k], Z[:, :k])
                lamp, Qproj = inner_solve(inner_solver_method, pnep, lamv=lam.copy(), neigs=len(lam) + 3, sigma=sigma,
                                          tol=tol / 10)
                II = np.argsort(abs(lamp - sigma), kind="stable")
                lam = lamp[II]; Qproj = Qproj[:, II]
                Q = Z[:, :k] @ Qproj
            conv_eig = 0
            ne = len(lam)
            err[k - 1, :ne] = [errmeasure(lam[s], Q[:, s]) for s in range(ne)]
            if errhist is not None:
                errhist.append(np.sort(err[k - 1, :ne]).copy())
            conv_eig = int(np.sum(err[k - 1, :ne] < tol))
            idx = np.argsort(err[k - 1, :ne], kind="stable")
            err[k - 1, :ne] = err[k - 1, idx]
            if k == m or conv_eig >= neigs:
                nrof = int(min(len(lam), neigs))
                lam = lam[idx[:nrof]]
                Q = Q[:, idx[:nrof]]
            conv_eig_hist[k - 1] = conv_eig
        k += 1
    k -= 1
    if conv_eig < neigs and neigs != np.inf:
        raise NoConvergenceException(lam, Q, None, "Number of iterations exceeded. maxit=%d." % maxit)
    nc = min(len(lam), conv_eig)
    return lam[:nc], Q[:, :nc], Z[:, :k], conv_eig_hist


# ----------------------------------------------------------------------------------
def compute_rf(nep, x, y=None, target=0.0, lam=None, tol=EPS * 100, maxit=80):
    """compute_rf_wrapper.jl:25-54 (ScalarNewtonInnerSolver)."""
    if y is None:
        y = x
    lam_iter = complex(target if lam is None else lam)
    dlam = np.inf; count = 0
    while abs(dlam) > tol and count < maxit:
        count += 1
        z1 = nep.compute_Mlincomb(lam_iter, x.reshape(-1, 1))
        z2 = nep.compute_Mlincomb(lam_iter, x.reshape(-1, 1), [1.0], 1)
        dlam = -np.vdot(y, z1) / np.vdot(y, z2)
        lam_iter += dlam
    return np.array([lam_iter])


def armijo_rule(nep, errmeasure, err0, lam, v, dlam, dv, armijo_factor, armijo_max):
    """method_newton.jl:598-609."""
    j = 0
    if armijo_factor < 1:
        while errmeasure(lam + dlam, v + dv) > err0 and j < armijo_max:
            j += 1
            dv = dv * armijo_factor
            dlam = dlam * armijo_factor
    return dlam, dv, j, armijo_factor ** j


def resinv(nep, errmeasure=None, tol=EPS * 100, maxit=100, lam=0.0, v=None, c=None,
           linsolvercreator=None, armijo_factor=1, armijo_max=5, hist=None):
    """method_newton.jl:142-226."""
    lam = complex(lam)
    v = np.array(v, dtype=complex)
    c = v.copy() if c is None else np.array(c, dtype=complex)
    n = len(v)
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    linsolver = linsolvercreator.create_linsolver(nep, lam)
    use_v_as_rf_vector = np.linalg.norm(c) == 0
    sigma = lam
    err = np.inf
    for k in range(1, maxit + 1):
        v = v / np.linalg.norm(v)
        err = errmeasure(lam, v)
        if use_v_as_rf_vector:
            c = v.copy()
        if hist is not None:
            hist.append((k, err, lam))
        if err < tol:
            return lam, v
        lam_vec = compute_rf(nep, v, y=c, lam=lam, target=sigma)
        lam1 = lam_vec[np.argmin(abs(lam_vec - lam))]
        dlam = lam1 - lam
        dv = -linsolver.lin_solve(nep.compute_Mlincomb(lam1, v.reshape(n, 1)))
        dlam, dv, j, scaling = armijo_rule(nep, errmeasure, err, lam, v, dlam, dv,
                                           float(armijo_factor), armijo_max)
        lam += dlam
        v = v + dv
    raise NoConvergenceException(lam, v, err, "Number of iterations exceeded. maxit=%d." % maxit)


def quasinewton(nep, errmeasure=None, tol=EPS * 100, maxit=100, lam=0.0, v=None, ws=None,
                linsolvercreator=None, armijo_factor=1, armijo_max=5, hist=None):
    """method_newton.jl:380-445."""
    lam = complex(lam)
    v = np.array(v, dtype=complex)
    ws = v.copy() if ws is None else np.array(ws, dtype=complex)
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    linsolver = linsolvercreator.create_linsolver(nep, lam)
    err = np.inf
    for k in range(1, maxit + 1):
        err = errmeasure(lam, v)
        if hist is not None:
            hist.append((k, err, lam))
        if err < tol:
            return lam, v
        u = nep.compute_Mlincomb(lam, v, [1.0], 0)
        w = nep.compute_Mlincomb(lam, v, [1.0], 1)
        dlam = -np.vdot(ws, u) / np.vdot(ws, w)
        z = dlam * w + u
        dv = -linsolver.lin_solve(z, tol=tol)
        dlam, dv, j, scaling = armijo_rule(nep, errmeasure, err, lam, v, dlam, dv,
                                           float(armijo_factor), armijo_max)
        lam += dlam
        v = v + dv
    raise NoConvergenceException(lam, v, err, "Number of iterations exceeded. maxit=%d." % maxit)


# ----------------------------------------------------------------------------------
def integrate_interval_trapezoidal(f, gv, a, b, N):
    """method_contour_common.jl:61-94."""
    h = (b - a) / N
    t = a + h * np.arange(N)
    f1 = f(t[0])
    m = len(gv)
    S = np.zeros(f1.shape + (m,), dtype=complex)
    G = np.zeros((N, m), dtype=complex)
    for i in range(m):
        G[:, i] = [gv[i](tt) for tt in t]
    for i in range(N):
        temp = f1 if i == 0 else f(t[i])
        for j in range(m):
            S[:, :, j] += temp * G[i, j]
    return S * h


def probe_block(n, k, seed=10):
    """Deterministic replacement of `Random.seed!(10); randn(n,k)` (method_beyncontour.jl:85-86);
    Julia's stream is not reproducible outside Julia, so only counts/residuals are compared."""
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n, k)).astype(complex)


def contour_beyn(nep, tol=np.sqrt(EPS), sigma=0.0, linsolvercreator=None, neigs=2, k=None,
                 radius=1, N=1000, errmeasure=None, sanity_check=True, rank_drop_tol=None,
                 Vh=None, info=None):
    """method_beyncontour.jl:49-185."""
    if k is None:
        k = neigs + 1
    if rank_drop_tol is None:
        rank_drop_tol = tol
    if np.isscalar(radius):
        radius = (radius, radius)
    if linsolvercreator is None:
        linsolvercreator = BackslashLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    g = lambda t: complex(radius[0] * np.cos(t), radius[1] * np.sin(t))
    gp = lambda t: complex(-radius[0] * np.sin(t), radius[1] * np.cos(t))
    n = nep.size(1)
    if k > n:
        raise ValueError("Cannot compute more eigenvalues than the size of the NEP with contour_beyn()")
    if k <= 0:
        raise ValueError("k must be positive")
    if Vh is None:
        Vh = probe_block(n, k)

    def local_linsolve(lam):
        return linsolvercreator.create_linsolver(nep, lam + sigma).lin_solve(Vh)

    f = lambda t: local_linsolve(g(t)) * gp(t)
    AA = integrate_interval_trapezoidal(f, [lambda s: 1.0 + 0j, g], 0, 2 * np.pi, N)
    A0 = AA[:, :, 0] / (2j * np.pi)
    A1 = AA[:, :, 1] / (2j * np.pi)
    V, S, Wh = sla.svd(A0, full_matrices=False)
    W = Wh.conj().T
    p = int(np.sum(S / S[0] > rank_drop_tol))
    V0 = V[:, :p]; W0 = W[:, :p]
    B = (V0.conj().T @ A1 @ W0) @ np.diag(1.0 / S[:p])
    lam, VB = sla.eig(B)
    lam = lam + sigma
    V = V0 @ VB
    V = V / np.linalg.norm(V, axis=0)[None, :]
    if info is not None:
        info.update(p=p, S=S, A0=A0, A1=A1)

    def inside(l):
        return ((l - sigma).real / radius[0]) ** 2 + ((l - sigma).imag / radius[1]) ** 2 <= 1

    if not sanity_check:
        si = np.argsort(abs(sigma - lam), kind="stable")
        ins = inside(lam[si])
        perm = np.argsort(~ins, kind="stable")
        return lam[si[perm]], V[:, si[perm]]
    errs = np.array([errmeasure(lam[i], V[:, i]) for i in range(p)])
    good = np.nonzero(errs < tol)[0]
    sgi = good[np.argsort(abs(sigma - lam[good]), kind="stable")]
    ins = inside(lam[sgi])
    perm = np.argsort(~ins, kind="stable")
    sel = sgi[perm]
    if len(sel) > neigs:
        sel = sel[:neigs]
    return lam[sel], V[:, sel]


# ----------------------------------------------------------------------------------
def probe_block_uniform(n, L, seed=10):
    """Deterministic replacement of `Random.seed!(10); U = rand(T,n,L); V = rand(T,n,L)` (method_block_SS.jl:77-79):
    real and imaginary parts uniform in [0,1), U drawn before V."""
    rng = np.random.Generator(np.random.Philox(seed))
    U = rng.random((n, L)) + 1j * rng.random((n, L))
    V = rng.random((n, L)) + 1j * rng.random((n, L))
    return U, V


def contour_block_SS(nep, tol=np.sqrt(EPS), sigma=0.0, linsolvercreator=None, neigs=np.inf, k=3, radius=1, N=1000,
                     K=3, Shat_mode="native", rank_drop_tol=None, U=None, V=None, info=None):
    """method_block_SS.jl:47-214 (Asakura/Sakurai/Tadano/Ikegami/Kimura block SS; L probe columns = k, 2K moments)."""
    if rank_drop_tol is None:
        rank_drop_tol = tol
    if linsolvercreator is None:
        linsolvercreator = BackslashLinSolverCreator()
    n = nep.size(1)
    L = k
    if U is None or V is None:
        U, V = probe_block_uniform(n, L)

    def local_linsolve(lam):                       # :81-86
        return linsolvercreator.create_linsolver(nep, lam + sigma).lin_solve(V)

    if Shat_mode == "JSIAM":                       # :96-122 (circle only; omega_j = radius*exp(2 pi i (j+1/2)/N))
        if not np.isscalar(radius):
            raise ValueError("JSIAM Shat_mode does not support ellipses")
        w = np.exp(2j * np.pi * (0.5 + np.arange(N)) / N)
        Shat = np.zeros((n, L, 2 * K), dtype=complex)
        for j in range(N):
            X = local_linsolve(radius * w[j])
            for kk in range(2 * K):
                Shat[:, :, kk] += w[j] ** (kk + 1) * X / N
    elif Shat_mode == "native":                    # :124-145
        r1 = (radius, radius) if np.isscalar(radius) else tuple(radius)
        g = lambda t: complex(r1[0] * np.cos(t), r1[1] * np.sin(t))
        gp = lambda t: complex(-r1[0] * np.sin(t), r1[1] * np.cos(t))
        f = lambda t: local_linsolve(g(t)) * gp(t) / (2j * np.pi)
        gv = [(lambda s, kk=kk: g(s) ** kk) for kk in range(2 * K)]
        Shat = integrate_interval_trapezoidal(f, gv, 0, 2 * np.pi, N)
    else:
        raise ValueError("Unknown Shat_mode: %s" % Shat_mode)
    Mhat = np.stack([U.conj().T @ Shat[:, :, kk] for kk in range(2 * K)], axis=2)      # :151-153
    m = K * L
    Hhat = np.zeros((m, m), dtype=complex); Hhat2 = np.zeros((m, m), dtype=complex)    # :157-167
    for i in range(K):
        for j in range(K):
            Hhat[i * L:(i + 1) * L, j * L:(j + 1) * L] = Mhat[:, :, i + j]
            Hhat2[i * L:(i + 1) * L, j * L:(j + 1) * L] = Mhat[:, :, i + j + 1]
    UU, SS, VVh = sla.svd(Hhat)                                                       # :172-188
    VV = VVh.conj().T
    mprime = int(np.sum(SS / SS[0] > rank_drop_tol))
    UU1 = UU[:, :mprime]; VV1 = VV[:, :mprime]
    H1 = UU1.conj().T @ Hhat @ VV1                                                     # :191-193
    H2 = UU1.conj().T @ Hhat2 @ VV1
    xi, X = sla.eig(H2, H1)                                                            # :196-197
    S = np.concatenate([Shat[:, :, j] for j in range(K)], axis=1)                      # :202-206
    Vout = S @ VV1 @ X
    factor = radius if Shat_mode == "JSIAM" else 1.0                                   # :210-211
    if info is not None:
        info.update(mprime=mprime, SS=SS, Mhat=Mhat)
    return sigma + factor * xi, Vout


# ----------------------------------------------------------------------------------
class Proj_SPMF_NEP:
    """src/NEPTypes.jl:647-800: N(lam) = W^H M(lam) V as the SPMF with B_i = W^H A_i V"""

    def __init__(self, orgnep):
        self.orgnep = orgnep
        self.nep_proj = None

    def set_projectmatrices(self, W, V):
        from . import neps
        WH = np.asarray(W).conj().T
        self.nep_proj = neps.SPMF_NEP([np.asarray(WH @ (A @ V)) for A in self.orgnep.get_Av()], self.orgnep.get_fv())

    def size(self, d=None):
        return self.nep_proj.size(d)


class IARInnerSolver:
    """src/inner_solver.jl:133-144,308-347"""

    def __init__(self, tol=1e-13, maxit=80, normalize_DEPs=False):
        self.tol, self.maxit, self.normalize_DEPs = tol, maxit, normalize_DEPs


def inner_solve(solver, pnep, lamv=None, V=None, neigs=10, sigma=0.0, tol=None):
    """src/inner_solver.jl:308-347 (IARInnerSolver with iar, starting vector ones; partial results on NoConvergence)"""
    from . import neps
    cheb = False
    if solver is None:                               # DefaultInnerSolver (inner_solver.jl:243-256)
        if isinstance(pnep.orgnep, neps.DEP):
            solver = IARInnerSolver(normalize_DEPs=True); cheb = True
        else:
            solver = IARInnerSolver()
    cheb = cheb or getattr(solver, "cheb", False)
    nep = pnep.nep_proj
    k = nep.size(1)
    if isinstance(pnep.orgnep, neps.DEP) and solver.normalize_DEPs:
        AA = nep.get_Av()
        nep = neps.DEP([np.linalg.solve(AA[0], AA[1 + i]) for i in range(len(AA) - 1)], pnep.orgnep.tauv)
    try:
        fn = iar_chebyshev if cheb else iar
        out = fn(nep, sigma=sigma, neigs=neigs, tol=solver.tol, maxit=solver.maxit, v=np.ones(k))
        return out[0], out[1]
    except NoConvergenceException as e:
        Q = np.zeros((k, 0), dtype=complex) if e.v is None else np.asarray(e.v).reshape(k, -1)
        return np.asarray(e.lam, dtype=complex).reshape(-1), Q


# ----------------------------------------------------------------------------------
def _cheb_L(m, a, b):
    """method_iar_chebyshev.jl:129-131: integration map in the Chebyshev basis of [a, b]"""
    L = np.diag(np.concatenate([[2.0], 1.0 / np.arange(2, m + 1)])) + np.diag(-1.0 / np.arange(1, m - 1), -2)
    return L * (b - a) / 4


def _cheb_T_at(x, idx):
    """T_i(x) for real x outside or inside [-1, 1]  (method_iar_chebyshev.jl:245-253)"""
    idx = np.asarray(idx, dtype=float)
    if abs(x) <= 1:
        return np.cos(idx * np.arccos(x))
    if x >= 1:
        return np.cosh(idx * np.arccosh(x))
    return ((-1.0) ** idx) * np.cosh(idx * np.arccosh(-x))


def _dd0_mat_fun(f, S, sigma):
    """method_iar_chebyshev.jl:474-497: f[S, sigma I] through f([[S, I], [0, sigma I]])"""
    n = S.shape[0]
    A = np.zeros((2 * n, 2 * n), dtype=complex)
    A[:n, :n] = S; A[:n, n:] = np.eye(n); A[n:, n:] = sigma * np.eye(n)
    return np.asarray(f(A))[:n, n:]


def iar_chebyshev(nep, orthmethod=dgks, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6, errmeasure=None,
                  sigma=0.0, gamma=1.0, v=None, check_error_every=1, a=None, b=None, compute_y0_method="auto"):
    """method_iar_chebyshev.jl:66-218 with the DEP / PEP / SPMF versions of compute_y0_cheb (:309-370).  A DEP or PEP
    with sigma != 0 or gamma != 1 is handled by the SPMF version (which carries shift and scale in its divided
    differences) instead of the reference's explicit shift_and_scale -- same spectrum."""
    from . import neps
    n = nep.size(1); m = maxit
    sigma = complex(sigma); gamma = complex(gamma)
    isdep = isinstance(nep, neps.DEP); ispep = isinstance(nep, neps.PEP)
    if a is None:
        a = -float(np.max(nep.tauv)) if isdep else -1.0
    if b is None:
        b = 0.0 if isdep else 1.0
    if compute_y0_method == "auto":
        compute_y0_method = "DEP" if isdep else ("PEP" if ispep else "SPMF")
    if (sigma != 0 or gamma != 1) and compute_y0_method in ("DEP", "PEP"):
        compute_y0_method = "SPMF"
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    cc = (a + b) / (a - b); kk = 2 / (b - a)
    Av = nep.get_Av(); fv = nep.get_fv()
    L = _cheb_L(m, a, b)
    Tc = np.cos(np.arange(m + 1) * np.arccos(cc))
    if compute_y0_method == "DEP":
        Ttau = np.array([_cheb_T_at(-kk * tau + cc, np.arange(m + 2)) for tau in nep.tauv])
    else:
        Li = np.linalg.inv(L[:m, :m])
        D = np.vstack([np.zeros((1, m)), Li[:m - 1, :]])
        if compute_y0_method == "SPMF":
            DDf = [gamma * _dd0_mat_fun(f, sigma * np.eye(m) + gamma * D, sigma) for f in fv]
    v = np.array(v, dtype=complex)
    V = np.zeros((n * (m + 1), m + 1), dtype=complex, order="F")
    H = np.zeros((m + 1, m), dtype=complex)
    M0inv = linsolvercreator.create_linsolver(nep, sigma)
    err = np.ones((m, m))
    lam = np.zeros(m + 1, dtype=complex); Q = np.zeros((n, m + 1), dtype=complex)
    V[:n, 0] = v / np.linalg.norm(v)
    k = 1; conv_eig = 0
    while k <= m and conv_eig < neigs:
        X = V[:n * k, k - 1].reshape((n, k), order="F")
        y = np.zeros((n, k + 1), dtype=complex, order="F")
        y[:, 1:k + 1] = X @ L[:k, :k]
        N = k
        if compute_y0_method == "DEP":                                    # :309-321
            y0 = X @ Tc[:N]
            for j in range(len(nep.tauv)):
                y0 = y0 - Av[j + 1] @ (y @ Ttau[j, :N + 1])
            y0 = M0inv.lin_solve(y0)
        elif compute_y0_method == "PEP":                                  # :331-343
            d = len(Av) - 1
            vv_ = Tc[:N].astype(complex)
            y0 = np.zeros(n, dtype=complex)
            for j in range(d):
                y0 = y0 + Av[j + 1] @ (X @ vv_)
                vv_ = D[:N, :N] @ vv_
            y0 = -M0inv.lin_solve(y0) - y @ Tc[:N + 1]
        else:                                                             # :355-366
            y0 = np.zeros(n, dtype=complex)
            for i in range(len(fv)):
                y0 = y0 + Av[i] @ (X @ (DDf[i][:N, :N] @ Tc[:N]))
            y0 = -M0inv.lin_solve(y0) - y @ Tc[:N + 1]
        y[:, 0] = y0
        vv = y.reshape(-1, order="F").copy()
        H[k, k - 1] = orthmethod(V[:n * (k + 1), :k], vv, H[:k, k - 1])
        V[:n * (k + 1), k] = vv
        if ((k % check_error_every == 0) or (k == m)) and k > 2:
            Dv, Z = _eig(H[:k, :k])
            Q = V[:n, :k] @ Z
            lam = sigma + gamma / Dv
            err[k - 1, :k] = [errmeasure(lam[s], Q[:, s]) for s in range(k)]
            conv_eig = int(np.sum(err[k - 1, :k] < tol))
            idx = np.argsort(err[k - 1, :k], kind="stable")
            err[k - 1, :k] = err[k - 1, idx]
            if k == m or conv_eig >= neigs:
                nrof = int(min(len(lam), neigs))
                lam = lam[idx[:nrof]]
                Q = Q[:, idx[:nrof]]
        k += 1
    k -= 1
    if conv_eig < neigs and neigs != np.inf:
        raise NoConvergenceException(lam, Q, err[k - 1, :len(lam)], "Number of iterations exceeded. maxit=%d." % maxit)
    nc = min(len(lam), conv_eig)
    return lam[:nc], Q[:, :nc], V[:, :k]


# ----------------------------------------------------------------------------------
def _symmetrizer_coefficients(m):
    """method_ilan.jl:419-427"""
    G = np.zeros((m + 1, m + 1), dtype=complex)
    G[:, 0] = 1.0 / np.arange(1, m + 2)
    for j in range(m):
        for i in range(m + 1):
            G[i, j + 1] = G[i, j] * (j + 1) / (i + j + 2)
    return G


def ilan(nep, orthmethod=dgks, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6, errmeasure=None, sigma=0.0,
         gamma=1.0, v=None, check_error_every=30, inner_solver_method=None, proj_solve=True):
    """method_ilan.jl:56-258 (infinite Lanczos for symmetric NEPs) with the SPMF version of Bmult (:370-378) for every
    SPMF-type NEP -- the reference's DEP version (:384-405) is the same product in factored form (test/ilan.jl:45-62)."""
    n = nep.size(1); m = maxit
    sigma = complex(sigma); gamma = complex(gamma)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    Av = nep.get_Av(); fv = nep.get_fv(); p = len(fv)
    v = np.array(v, dtype=complex)
    V = np.zeros((n, m + 1), dtype=complex); Q = np.zeros((n, m + 1), dtype=complex)
    Qp = np.zeros_like(Q); Qn = np.zeros_like(Q); Z = np.zeros_like(Q); W = np.zeros_like(Q)
    H = np.zeros((m + 1, m), dtype=complex); HH = np.zeros((m + 1, m), dtype=complex)
    om = np.zeros(m + 1, dtype=complex)
    a = gamma ** np.arange(2 * m + 3); a[0] = 0
    M0inv = linsolvercreator.create_linsolver(nep, sigma)
    err = np.full((m, m + 1), np.nan)
    QQ = np.zeros((n, m + 1), dtype=complex)
    # precompute_data (:322-343): FDH_t[i,j] = fD[i+j+1, t], fD[:, t] = f_t(SS)[:, 0], SS = sigma I + gamma subdiag(1..)
    SS = np.diag(sigma * np.ones(2 * m + 2)) + np.diag(gamma * np.arange(1, 2 * m + 2), -1)
    fD = np.column_stack([np.asarray(f(SS))[:, 0] for f in fv])
    FDH = [np.array([[fD[i + j + 1, t] for j in range(m + 1)] for i in range(m + 1)]) for t in range(p)]
    G = _symmetrizer_coefficients(m)
    msum = lambda X, Y: np.sum(X * Y)                                     # mat_sum (:299-308): no conjugation
    Q[:, 0] = v / np.linalg.norm(v)
    om[0] = np.vdot(Q[:, 0], nep.compute_Mlincomb(0, np.column_stack([Q[:, 0], Q[:, 0]]), [0, 1]))
    V[:, 0] = Q[:, 0]
    lam = np.zeros(0, dtype=complex)
    k = 1; conv_eig = 0
    while k <= m and conv_eig < neigs:
        if not proj_solve:
            QQ[:, k - 1] = Q[:, 0]
        Qn[:, 1:k + 1] = Q[:, :k] / np.arange(1, k + 1)[None, :]
        Qn[:, 0] = nep.compute_Mlincomb(sigma, Qn[:, :k + 1].copy(), a[:k + 1])
        Qn[:, 0] = -M0inv.lin_solve(Qn[:, 0])
        Z[:] = 0                                                           # Bmult (:370-378)
        for t in range(p):
            Z[:, :k + 1] += Av[t] @ (Qn[:, :k + 1] @ (G[:k + 1, :k + 1] * FDH[t][:k + 1, :k + 1]))
        if k > 1:
            beta = msum(Z[:, :k], Qp[:, :k])
        alpha = msum(Z[:, :k], Q[:, :k])
        eta = msum(Z[:, :k + 1], Qn[:, :k + 1])
        H[k - 1, k - 1] = alpha / om[k - 1]
        if k > 1:
            H[k - 2, k - 1] = beta / om[k - 2]
        Qn[:, :k] -= H[k - 1, k - 1] * Q[:, :k]
        if k > 1:
            Qn[:, :k] -= H[k - 2, k - 1] * Qp[:, :k]
        H[k, k - 1] = np.linalg.norm(Qn)
        Qn[:, :k + 1] /= H[k, k - 1]
        om[k] = eta - 2 * alpha * H[k - 1, k - 1] + om[k - 1] * H[k - 1, k - 1] ** 2
        if k > 1:
            om[k] = om[k] - 2 * beta * H[k - 2, k - 1] + om[k - 2] * H[k - 2, k - 1] ** 2
        om[k] = om[k] / H[k, k - 1] ** 2
        V[:, k] = Qn[:, 0]
        vk = V[:, k].copy()
        orthmethod(V[:, :k], vk, HH[:k, k - 1])
        V[:, k] = vk
        if (check_error_every != np.inf and k % check_error_every == 0) or k == m:
            if not proj_solve:
                D, WR = _eig(H[:k, :k])
                W[:, :k] = QQ[:, :k] @ WR
                lam = sigma + gamma / D
            else:
                VV = V[:, :k + 1]
                pnep = Proj_SPMF_NEP(nep)
                pnep.set_projectmatrices(VV, VV)
                lamp, Wp = inner_solve(inner_solver_method, pnep, neigs=m, tol=tol)
                q = len(lamp)
                lam = lamp[:q]
                q = min(q, m)
                W[:, :q] = VV @ Wp[:, :q]
            ne = min(len(lam), m + 1)
            err[k - 1, :ne] = [errmeasure(lam[s], W[:, s]) for s in range(ne)]
            conv_eig = int(np.sum(err[k - 1, :ne] < tol))
            idx = np.argsort(err[k - 1, :k], kind="stable")
            err[k - 1, :k] = err[k - 1, idx]
            if k == m or conv_eig >= neigs:
                nrof = int(min(conv_eig, neigs))
                lam = lam[idx[:nrof]]
                W = W[:, idx[:len(lam)]]
        k += 1
        Qp[:] = Q; Q[:] = Qn; Qn[:] = 0
    k -= 1
    if conv_eig < neigs and neigs != np.inf:
        raise NoConvergenceException(lam, W, None, "Number of iterations exceeded. maxit=%d." % maxit)
    return lam, W[:, :len(lam)], V[:, :k + 1], H[:k, :k - 1], om[:k]


# ----------------------------------------------------------------------------------
def _discard_ritz_values(dd, D, R):
    dd = np.array(dd, dtype=complex)
    for j in range(len(D)):
        dd[np.abs(dd - D[j]) < R] = np.inf
    return dd


def residual_eigval_sorter(nep, dd, vv, sigma, D, R, Vk, errmeasure=None):
    """method_nlar.jl:185-196"""
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    dd = np.asarray(dd, dtype=complex)
    dd2 = _discard_ritz_values(dd, D, R)
    eig_res = np.array([errmeasure(dd[i], Vk @ vv[:, i]) for i in range(len(dd))])
    with np.errstate(all="ignore"):
        key = eig_res * np.abs(dd2 - sigma)
    key = np.where(np.isnan(key), np.inf, key)
    ii = np.argsort(key, kind="stable")
    return dd[ii], vv[:, ii]


def nlar(nep, neigs=10, errmeasure=None, tol=EPS * 100, maxit=100, lam=0.0, v=None, linsolvercreator=None, R=0.01,
         eigval_sorter=residual_eigval_sorter, max_subspace=100, num_restart_ritz_vecs=8, inner_solver_method=None,
         orthmethod=mgs):
    """method_nlar.jl:30-164 (qrfact_orth=false); after a restart the projected matrices are rebuilt in full (the
    reference only refreshes the last row and column)."""
    n = nep.size(1)
    maxit = min(maxit, n)
    num_restart_ritz_vecs = min(num_restart_ritz_vecs, neigs)
    if max_subspace < num_restart_ritz_vecs:
        max_subspace = num_restart_ritz_vecs + 20
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    sigma = complex(lam); nu = sigma
    V = np.zeros((n, max_subspace + 1), dtype=complex)
    X = np.zeros((n, neigs), dtype=complex)
    v = np.array(v, dtype=complex)
    V[:, 0] = v / np.linalg.norm(v)
    cbs = 1
    D = np.zeros(neigs, dtype=complex)
    m = 0; k = 1
    linsolver = linsolvercreator.create_linsolver(nep, sigma)
    err = np.inf; u = None
    while m < neigs and k < maxit:
        Vk = V[:, :cbs]
        pnep = Proj_SPMF_NEP(nep)
        pnep.set_projectmatrices(Vk, Vk)
        dd, vv = inner_solve(inner_solver_method, pnep, neigs=neigs, sigma=sigma)
        dd = np.asarray(dd, dtype=complex).reshape(-1); vv = np.asarray(vv, dtype=complex).reshape(cbs, -1)
        nuv, yv = eigval_sorter(nep, dd, vv, sigma, D[:m], R, Vk)
        nu = nuv[0]
        u = Vk @ yv[:, 0]; u = u / np.linalg.norm(u)
        res = nep.compute_Mlincomb(nu, u)
        err = errmeasure(nu, u)
        if err < tol:
            D[m] = nu; X[:, m] = u; m += 1
            if m >= neigs:
                break
            nuv, yv = eigval_sorter(nep, dd, vv, sigma, D[:m], R, Vk)
            u1 = Vk @ yv[:, 0]; u1 = u1 / np.linalg.norm(u1)
            res = nep.compute_Mlincomb(nuv[0], u1)
        if cbs >= max_subspace:
            nr = min(num_restart_ritz_vecs, yv.shape[1])
            Qh, _ = np.linalg.qr(np.column_stack([X[:, :m], Vk @ yv[:, :nr]]))
            cbs = Qh.shape[1]
            V[:, :cbs] = Qh
        else:
            dv = linsolver.lin_solve(res)
            h = np.zeros(cbs, dtype=complex)
            orthmethod(Vk, dv, h)
            V[:, cbs] = dv
            cbs += 1
        k += 1
    if k >= maxit and m < neigs:
        raise NoConvergenceException(nu, u, err, "Number of iterations exceeded. maxit=%d and only %d eigenvalues "
                                                 "converged out of %d." % (maxit, m, neigs))
    return D, X


# ----------------------------------------------------------------------------------
def jd_betcke(nep, maxit=100, neigs=1, projtype="PetrovGalerkin", inner_solver_method=None, orthmethod=dgks,
              errmeasure=None, linsolvercreator=None, tol=EPS * 100, lam=0.0, v=None, target=0.0):
    """method_jd.jl:52-175"""
    n = nep.size(1)
    if maxit > n:
        raise ValueError("maxit = %d is larger than size of NEP = %d." % (maxit, n))
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    lam = complex(lam); target = complex(target)
    lam_vec = np.zeros(neigs, dtype=complex); u_vec = np.zeros((n, neigs), dtype=complex)
    u = np.array(v, dtype=complex); u = u / np.linalg.norm(u)
    conveig = 0
    err = errmeasure(lam, u)
    if err < tol:
        lam_vec[conveig] = lam; u_vec[:, conveig] = u; conveig += 1
    if conveig == neigs:
        return lam_vec, u_vec
    Vm = np.zeros((n, maxit + 1), dtype=complex); Vm[:, 0] = u
    pg = projtype == "PetrovGalerkin"
    if pg:
        Wm = np.zeros((n, maxit + 1), dtype=complex)
        w0 = nep.compute_Mlincomb(lam, u); Wm[:, 0] = w0 / np.linalg.norm(w0)
    else:
        Wm = Vm
    for k in range(1, maxit + 1):
        V = Vm[:, :k]; W = Wm[:, :k]
        pnep = Proj_SPMF_NEP(nep)
        pnep.set_projectmatrices(W, V)
        lamv, sv = inner_solve(inner_solver_method, pnep, lamv=lam * np.ones(conveig + 1, dtype=complex), sigma=target,
                               neigs=conveig + 1)
        lamv = np.asarray(lamv, dtype=complex).reshape(-1); sv = np.asarray(sv).reshape(k, -1)
        NN = min(conveig + 1, len(lamv))
        c = np.argsort(abs(lamv - target), kind="stable")
        lam = lamv[c[NN - 1]]; s = sv[:, c[NN - 1]]; s = s / np.linalg.norm(s)
        u = V @ s
        err = errmeasure(lam, u)
        if err < tol and (conveig == 0 or np.all(abs(lam - lam_vec[:conveig]) / abs(lam_vec[:conveig]) > np.sqrt(np.sqrt(EPS)))):
            lam_vec[conveig] = lam; u_vec[:, conveig] = u; conveig += 1
        if conveig == neigs:
            return lam_vec, u_vec
        pk = nep.compute_Mlincomb(lam, u.reshape(-1, 1), [1.0], 1)
        vnew = linsolvercreator.create_linsolver(nep, lam).lin_solve(pk)
        h = np.zeros(k, dtype=complex)
        orthmethod(V, vnew, h)
        Vm[:, k] = vnew
        if pg:
            wnew = nep.compute_Mlincomb(lam, u)
            orthmethod(W, wnew, h)
            Wm[:, k] = wnew
    raise NoConvergenceException(np.concatenate([lam_vec[:conveig], [lam]]), np.column_stack([u_vec[:, :conveig], u]), err,
                                 "Number of iterations exceeded. maxit=%d and only %d eigenvalues converged out of %d."
                                 % (maxit, conveig, neigs))
