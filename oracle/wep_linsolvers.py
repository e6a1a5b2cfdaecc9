"""Oracle (test infrastructure only): the waveguide-specific linear solvers of the reference, restated with NumPy/SciPy.

  src/gallery_extra/waveguide/Waveguide.jl:159-162      Pinv (inverse of the boundary operator, Ringh (2.8))
  src/gallery_extra/waveguide/Waveguide.jl:394-425      SchurMatVec (Ringh (2.13), (3.3))
  src/gallery_extra/waveguide/Waveguide.jl:428-486      WEPGMRESLinSolver / WEPBackslashLinSolver / WEPFactorizedLinSolver
  src/gallery_extra/waveguide/Waveguide.jl:489-519      WEPLinSolverCreator, create_linsolver
  src/gallery_extra/waveguide/Waveguide.jl:523-550      construct_WEP_schur_complement (Ringh Prop. 3.1)
  src/gallery_extra/waveguide/Waveguide.jl:552-567      lin_solve through the Schur complement (Ringh Prop. 2.1)
  src/gallery_extra/waveguide/waveguide_preconditioner.jl:10-421   Sylvester solver by FFT diagonalisation (Ringh 5.3),
                                                        Sylvester-SMW preconditioner (Ringh section 4)

IterativeSolvers.gmres (0.9.2, absent from /root/reference) is replaced by a plain restarted, left-preconditioned GMRES
(Saad, Alg. 6.9 with Givens rotations), the published algorithm.  Pinned by test/wep_small.jl: the preconditioner with one
point per region is the exact inverse of the Schur complement (1e-14, :24-28); resinv with each of the three solver types
converges to a residual below 1e-10 (:41-61).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ---------------------------------------------------------------------------------------------- boundary operator
def Pinv(nep, lam, x):
    """Waveguide.jl:159-162: [R(Rinv(x_-) ./ s_-(lam)); R(Rinv(x_+) ./ s_+(lam))]"""
    wd = nep.wd; nz = wd.nz
    s = wd.S_scalar(lam)
    x = np.asarray(x, dtype=complex).ravel()
    return np.concatenate([wd.R(wd.Rinv(x[:nz]) / s[:nz]), wd.R(wd.Rinv(x[nz:]) / s[nz:])])


def P_inv_m(nep, lam, v):
    """Waveguide.jl:270-281"""
    wd = nep.wd
    return wd.R(wd.Rinv(v) / wd.S_scalar(lam)[:wd.nz])


def P_inv_p(nep, lam, v):
    """Waveguide.jl:283-294"""
    wd = nep.wd
    return wd.R(wd.Rinv(v) / wd.S_scalar(lam)[wd.nz:])


# ---------------------------------------------------------------------------------------------- Schur complement
class SchurMatVec:
    """Waveguide.jl:394-425: v -> vec(A(lam) X + X B + K .* X) - C1 Pinv(lam, C2T v),  X = reshape(v, nz, nx)"""

    def __init__(self, nep, lam):
        self.nep, self.lam = nep, complex(lam)
        self.A = sp.csc_matrix(nep._A(self.lam))
        self.shape = (nep.wd.nx * nep.wd.nz,) * 2

    def __call__(self, v):
        nep = self.nep; wd = nep.wd
        v = np.asarray(v, dtype=complex).ravel()
        X = v.reshape((wd.nz, wd.nx), order="F")
        Y = self.A @ X + (wd.Dxx.T @ X.T).T + nep.K_scaled * X
        return np.asarray(Y).ravel(order="F") - wd.C1 @ Pinv(nep, self.lam, wd.C2T @ v)


def construct_WEP_schur_complement(nep, lam):
    """Waveguide.jl:523-550 (Ringh Prop. 3.1): kron(B^T, I) + kron(I, A) + diag(K) - kron(E, Pinv_-) - kron(EE, Pinv_+)"""
    wd = nep.wd; nx, nz = wd.nx, wd.nz
    I = np.eye(nz)
    Pm = np.column_stack([P_inv_m(nep, lam, I[:, i]) for i in range(nz)])
    Pp = np.column_stack([P_inv_p(nep, lam, I[:, i]) for i in range(nz)])
    d1 = 2 / wd.hx; d2 = -1 / (2 * wd.hx)
    E = sp.lil_matrix((nx, nx)); E[0, 0] = d1 / wd.hx ** 2; E[0, 1] = d2 / wd.hx ** 2
    EE = sp.lil_matrix((nx, nx)); EE[nx - 1, nx - 1] = d1 / wd.hx ** 2; EE[nx - 1, nx - 2] = d2 / wd.hx ** 2
    S = (sp.kron(wd.Dxx.T, sp.identity(nz)) + sp.kron(sp.identity(nx), nep._A(lam)) + sp.diags(nep.K_scaled.ravel(order="F"))
         - sp.kron(E, Pm) - sp.kron(EE, Pp))
    return sp.csc_matrix(S, dtype=complex)


# ---------------------------------------------------------------------------------------------- GMRES
def gmres(matvec, b, reltol=None, tol=None, restart=20, maxiter=None, Pl=None, log=False):
    """restarted GMRES with left preconditioner Pl (a callable r -> Pl^{-1} r); stops when the preconditioned residual
    has dropped by reltol relative to the initial one (IterativeSolvers 0.9 `gmres(A, b; reltol, restart, maxiter, Pl)`)"""
    b = np.asarray(b, dtype=complex)
    n = len(b)
    rt = reltol if reltol is not None else (tol if tol is not None else np.sqrt(np.finfo(float).eps))
    maxiter = n if maxiter is None else maxiter
    prec = (lambda r: r) if Pl is None else Pl
    x = np.zeros(n, dtype=complex)
    r = prec(b.copy())
    beta0 = np.linalg.norm(r)
    hist = [beta0]
    if beta0 == 0:
        return (x, hist) if log else x
    its = 0
    beta = beta0
    while its < maxiter and beta > rt * beta0:
        m = min(restart, maxiter - its)
        V = np.zeros((n, m + 1), dtype=complex); H = np.zeros((m + 1, m), dtype=complex)
        cs = np.zeros(m, dtype=complex); sn = np.zeros(m, dtype=complex); g = np.zeros(m + 1, dtype=complex)
        V[:, 0] = r / beta; g[0] = beta
        k = 0
        for j in range(m):
            w = prec(matvec(V[:, j]))
            for i in range(j + 1):
                H[i, j] = np.vdot(V[:, i], w); w = w - H[i, j] * V[:, i]
            H[j + 1, j] = np.linalg.norm(w)
            if H[j + 1, j] != 0:
                V[:, j + 1] = w / H[j + 1, j]
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -np.conj(sn[i]) * H[i, j] + np.conj(cs[i]) * H[i + 1, j]
                H[i, j] = t
            den = np.sqrt(abs(H[j, j]) ** 2 + abs(H[j + 1, j]) ** 2)
            cs[j] = np.conj(H[j, j]) / den if den else 1.0; sn[j] = np.conj(H[j + 1, j]) / den if den else 0.0
            H[j, j] = cs[j] * H[j, j] + sn[j] * H[j + 1, j]; H[j + 1, j] = 0
            g[j + 1] = -np.conj(sn[j]) * g[j]; g[j] = cs[j] * g[j]
            its += 1; k = j + 1
            hist.append(abs(g[j + 1]))
            if abs(g[j + 1]) <= rt * beta0:
                break
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x = x + V[:, :k] @ y
        r = prec(b - matvec(x))
        beta = np.linalg.norm(r)
    return (x, hist) if log else x


# ---------------------------------------------------------------------------------------------- the three solvers
class _WEPSchurSolver:
    """Waveguide.jl:552-567 (Ringh Prop. 2.1): eliminate the boundary unknowns, solve with the Schur complement, recover them"""

    def __init__(self, nep, lam):
        self.nep, self.lam = nep, complex(lam)

    def lin_solve(self, x, tol=np.finfo(float).eps):
        nep = self.nep; wd = nep.wd
        N = wd.nx * wd.nz
        x = np.asarray(x, dtype=complex).ravel()
        x_int, x_ext = x[:N], x[N:]
        rhs = x_int - wd.C1 @ Pinv(nep, self.lam, x_ext)
        q = self.inner(rhs, tol)
        return np.concatenate([q, Pinv(nep, self.lam, -(wd.C2T @ q) + x_ext)])


class WEPBackslashLinSolver(_WEPSchurSolver):
    """Waveguide.jl:449-463"""

    def __init__(self, nep, lam, kwargs=()):
        super().__init__(nep, lam)
        self.schur_comp = construct_WEP_schur_complement(nep, lam)

    def inner(self, rhs, tol):
        return spla.splu(self.schur_comp).solve(rhs)


class WEPFactorizedLinSolver(_WEPSchurSolver):
    """Waveguide.jl:466-480"""

    def __init__(self, nep, lam, kwargs=()):
        super().__init__(nep, lam)
        self.schur_comp_fact = spla.splu(construct_WEP_schur_complement(nep, lam))

    def inner(self, rhs, tol):
        return self.schur_comp_fact.solve(rhs)


class WEPGMRESLinSolver(_WEPSchurSolver):
    """Waveguide.jl:428-446: matrix-free GMRES on the Schur complement; kwargs as ((name, value), ...)"""

    def __init__(self, nep, lam, kwargs=()):
        super().__init__(nep, lam)
        self.schur = SchurMatVec(nep, lam)
        self.kwargs = dict(kwargs)
        self.iterations = []

    def inner(self, rhs, tol):
        kw = dict(self.kwargs)
        kw.setdefault("reltol", tol)
        log = kw.pop("log", False)
        q, hist = gmres(self.schur, rhs, log=True, **kw)
        self.iterations.append(len(hist) - 1)
        return q


class WEPLinSolverCreator:
    """Waveguide.jl:489-519: solver_type in {"backslash", "factorized", "gmres"}"""

    def __init__(self, solver_type="factorized", kwargs=()):
        self.solver_type, self.kwargs = solver_type, kwargs

    def create_linsolver(self, nep, lam):
        if not hasattr(nep, "wd"):
            raise TypeError("WEPLinSolver can only be used in combination with WEPs: type(nep)=%s" % type(nep).__name__)
        if self.solver_type == "backslash":
            return WEPBackslashLinSolver(nep, lam, self.kwargs)
        if self.solver_type == "gmres":
            return WEPGMRESLinSolver(nep, lam, self.kwargs)
        if self.solver_type == "factorized":
            return WEPFactorizedLinSolver(nep, lam, self.kwargs)
        raise ValueError("Unknown type of solver_type in linsolvercreator:%s" % self.solver_type)


# ---------------------------------------------------------------------------------------------- Sylvester solver
def solve_wg_sylvester_fft(C, lam, k_bar, hx, hz):
    """waveguide_preconditioner.jl:120-160 (Ringh 5.3): X with A X + X B = C for A = Dzz + 2 lam Dz + (lam^2 + k_bar) I
    (circulant: diagonalised by the DFT along z) and B = Dxx (Dirichlet second difference: diagonalised by the sine
    transform along x).  C is nz x nx."""
    C = np.asarray(C, dtype=complex)
    nz, nx = C.shape
    alpha = lam ** 2 + k_bar
    v = np.zeros(nz, dtype=complex); v[0] = -2; v[1] = 1; v[nz - 1] = 1; v /= hz ** 2
    w = np.zeros(nz, dtype=complex); w[1] = 1; w[nz - 1] = -1; w *= lam / hz
    D = np.fft.fft(v + w) + alpha
    S = -(4.0 / hx ** 2) * np.sin(np.pi * np.arange(1, nx + 1) / (2 * (nx + 1))) ** 2
    Wm = np.sqrt(2.0 / (nx + 1)) * np.sin(np.pi * np.outer(np.arange(1, nx + 1), np.arange(1, nx + 1)) / (nx + 1))
    # change of variables: Vh(Wh(C')') -- inverse DFT along z (scaled to be unitary), sine transform along x
    T = np.fft.ifft(C @ Wm, axis=0) * np.sqrt(nz)
    Z = T / (D[:, None] + S[None, :])
    return np.fft.fft(Z, axis=0) / np.sqrt(nz) @ Wm


# ---------------------------------------------------------------------------------------------- Sylvester-SMW
class _Regions:
    """index sets of waveguide_preconditioner.jl:233-252: N x (N+4) regions; z-blocks of L = n/N points, x-blocks: the two
    single boundary columns on either side, N blocks of L interior columns in between"""

    def __init__(self, n, N):
        self.n, self.N, self.L = n, N, n // N
        self.mm = N * N + 4 * N

    def k2ij(self, k):                         # k = 0..mm-1  ->  (i, j), i = 0..N-1, j = 0..N+3
        return divmod(k, self.N + 4)

    def II(self, i):
        return slice(i * self.L, (i + 1) * self.L)

    def JJ(self, j):                           # interior x-block j = 2..N+1
        return slice((j - 2) * self.L + 2, (j - 1) * self.L + 2)

    def JJ2(self, j):                          # boundary column
        n = self.n
        return {0: 0, 1: 1, self.N + 2: n + 2, self.N + 3: n + 3}[j]

    def boundary(self, j):
        return j in (0, 1, self.N + 2, self.N + 3)

    def functionals(self, X):
        """block means of waveguide_preconditioner.jl:297-304 for all regions (region k = i (N+4) + j)"""
        N, L, n = self.N, self.L, self.n
        b = np.zeros((N, N + 4), dtype=complex)
        b[:, 2:N + 2] = X[:, 2:n + 2].reshape(N, L, N, L).mean(axis=(1, 3))
        for j in (0, 1, N + 2, N + 3):
            b[:, j] = X[:, self.JJ2(j)].reshape(N, L).mean(axis=1)
        return b.ravel()

    def expand(self, alpha, K):
        """sum_k alpha_k * (K restricted to region k) as an nz x nx matrix, and the two boundary vectors
        sum_k alpha_k e_k of waveguide_preconditioner.jl:382-412 (their P^{-1} images are added by the caller, once each,
        which is the reference's per-region loop by linearity)"""
        N, L, n = self.N, self.L, self.n
        A = np.asarray(alpha).reshape(N, N + 4)
        Y = np.zeros_like(K, dtype=complex)
        Y[:, 2:n + 2] = np.kron(A[:, 2:N + 2], np.ones((L, L))) * K[:, 2:n + 2]
        for j in (0, 1, N + 2, N + 3):
            Y[:, self.JJ2(j)] = np.repeat(A[:, j], L) * K[:, self.JJ2(j)]
        return Y, A


class WEPPreconditioner:
    """waveguide_preconditioner.jl:10-109,221-421: the coefficient matrix K and the boundary operators are replaced by
    their region-wise constant parts, which makes the Schur complement a Sylvester operator plus a rank-mm correction;
    Sherman-Morrison-Woodbury with mm = N^2 + 4 N unknowns.  Callable: r -> approximate Schur^{-1} r."""

    def __init__(self, nep, N, sigma):
        wd = nep.wd
        if wd.nz + 4 != wd.nx:
            raise ValueError("This implementation requires nx = nz + 4. Provided NEP has nz = %d and nx = %d" % (wd.nz, wd.nx))
        if wd.nz % N != 0:
            raise ValueError("This implementation is uniform in the blocking and therefore requires nz/N to be an integer. "
                             "Provided data is nz = %d with N = %d" % (wd.nz, N))
        self.nep, self.sigma, self.N = nep, complex(sigma), N
        self.reg = _Regions(wd.nz, N)
        self.dd1 = (2 / wd.hx) / wd.hx ** 2
        self.dd2 = (-1 / (2 * wd.hx)) / wd.hx ** 2
        self.K = nep.K_scaled
        self.Linv = lambda rhs: solve_wg_sylvester_fft(rhs, self.sigma, nep.k_bar, wd.hx, wd.hz)
        self.Pm = lambda v: -P_inv_m(nep, self.sigma, v)          # minus sign as in Ringh (4.10)
        self.Pp = lambda v: -P_inv_p(nep, self.sigma, v)
        self.M = self._generate()

    def _Etilde(self, k, coef=1.0, out=None):
        """coef * (k-th correction term) accumulated into out (nz x nx): waveguide_preconditioner.jl:263-288 / :382-412"""
        wd = self.nep.wd; reg = self.reg; nz, nx = wd.nz, wd.nx
        E = np.zeros((nz, nx), dtype=complex) if out is None else out
        i, j = reg.k2ij(k)
        II = reg.II(i)
        if reg.boundary(j):
            c = reg.JJ2(j)
            E[II, c] += coef * self.K[II, c]
            ek = np.zeros(nz, dtype=complex)
            ek[II] = self.dd1 if j in (0, reg.N + 3) else self.dd2
            if j in (0, 1):
                E[:, 0] += coef * self.Pm(ek)
            else:
                E[:, nx - 1] += coef * self.Pp(ek)
        else:
            E[II, reg.JJ(j)] += coef * self.K[II, reg.JJ(j)]
        return E

    def _generate(self):
        """waveguide_preconditioner.jl:221-313: M[kk, k] = functional_kk(Linv(E_k)) + identity; factorised"""
        import scipy.linalg as sla
        reg = self.reg
        M = np.zeros((reg.mm, reg.mm), dtype=complex)
        for k in range(reg.mm):
            M[:, k] = reg.functionals(self.Linv(self._Etilde(k)))
        M += np.eye(reg.mm)
        return sla.lu_factor(M)

    def __call__(self, r):
        """solve_smw, waveguide_preconditioner.jl:323-421: Linv C - Linv(sum_k alpha_k E_k), alpha = M \\ functionals(Linv C)"""
        import scipy.linalg as sla
        wd = self.nep.wd
        C = self.Linv(np.asarray(r, dtype=complex).reshape((wd.nz, wd.nx), order="F"))
        alpha = sla.lu_solve(self.M, self.reg.functionals(C))
        reg = self.reg; L = reg.L; N = reg.N
        Y, A = reg.expand(alpha, self.K)
        Y[:, 0] += self.Pm(self.dd1 * np.repeat(A[:, 0], L) + self.dd2 * np.repeat(A[:, 1], L))
        Y[:, wd.nx - 1] += self.Pp(self.dd2 * np.repeat(A[:, N + 2], L) + self.dd1 * np.repeat(A[:, N + 3], L))
        return (C - self.Linv(Y)).ravel(order="F")


def wep_generate_preconditioner(nep, N, sigma):
    """waveguide_preconditioner.jl:36-47"""
    return WEPPreconditioner(nep, N, sigma)
