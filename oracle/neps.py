"""Oracle NEP types (test infrastructure only): a NumPy/SciPy restatement of

  src/NEPCore.jl:113-160,218-228     generic compute_Mlincomb(!) semantics, _from_MM
  src/NEPTypes.jl:162-394            SPMF_NEP: compute_MM, compute_Mder, linear combination
  src/NEPTypes.jl:427-513            DEP
  src/types_poly.jl:31-98            PEP
  src/NEPTypes.jl:828-898            SumNEP
  src/NEPTypes.jl:940-1045           compute_Mlincomb for DEP / SPMF_NEP / PEP
  src/NEPTypes.jl:1055-1160          DerSPMF
  src/NEPTransformations.jl:92-105   shift_and_scale(::SPMF_NEP)

Like the reference, scalar functions f_i are callables valid for scalars AND for
square matrices (matrix functions); derivatives are obtained with the bidiagonal /
Jordan-matrix trick exactly as the reference does (no closed forms here -- the
product uses closed-form derivative tables, so the two routes are independent).
"""
import math
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


# ----------------------------------------------------------------------------------
# scalar / matrix functions
def _ismat(S):
    return isinstance(S, np.ndarray) and S.ndim == 2


def f_one():
    return lambda S: np.eye(S.shape[0], dtype=complex) if _ismat(S) else 1.0 + 0 * S


def f_id():
    return lambda S: S


def f_neg():
    return lambda S: -S


def f_pow(j):
    return lambda S: np.linalg.matrix_power(S, j) if _ismat(S) else S ** j


def f_exp(c):
    """S -> exp(c*S)"""
    return lambda S: sla.expm(c * S) if _ismat(S) else np.exp(c * S)


def f_isqrt(shift):
    """S -> 1im*sqrt(S + shift*one(S))   (NLEVP_native.jl:13-14)"""
    def f(S):
        if _ismat(S):
            return 1j * sla.sqrtm(S.astype(complex) + shift * np.eye(S.shape[0]))
        return 1j * np.sqrt(complex(S) + shift)
    return f


def f_compose_affine(f, scale, shift):
    """S -> f(scale*S + shift*one(S))   (NEPTransformations.jl:100)"""
    def g(S):
        if _ismat(S):
            return f(scale * S + shift * np.eye(S.shape[0]))
        return f(scale * S + shift)
    return g


def _matmul(A, X):
    return A @ X


def _issparse(A):
    return sp.issparse(A)


def _bidiag(lam, k, sub):
    """diagm(0 => fill(lam,k), -1 => sub)"""
    S = np.zeros((k, k), dtype=complex)
    S[np.arange(k), np.arange(k)] = lam
    if k > 1:
        S[np.arange(1, k), np.arange(k - 1)] = sub
    return S


# ----------------------------------------------------------------------------------
def result_type(nep_is_real, *args):
    """Element type the reference returns from compute_Mder / compute_Mlincomb / compute_MM (test/compute_types.jl:62-75,
    95-106, 146-155): promote_type of the NEP's element type and of every argument, made complex when the NEP is not real.
    Restricted to the two precisions NumPy and the device share: float64 and complex128 (Julia's Float16/32 and BigFloat rows
    of that test have no counterpart here; a lower-precision complex argument such as ComplexF16 promotes to complex)."""
    cplx = (not nep_is_real) or any(a is not None and np.iscomplexobj(a) for a in args)
    return np.complex128 if cplx else np.float64


class NEP:
    def size(self, d=None):
        return (self.n, self.n) if d is None else self.n

    # NEPCore.jl:113-125: manual scaling of the columns by a
    def compute_Mlincomb(self, lam, V, a=None, startder=0):
        V = np.array(V, dtype=complex, copy=True)
        if V.ndim == 1:
            V = V.reshape(-1, 1)
        if a is None:
            a = np.ones(V.shape[1], dtype=complex)
        a = np.array(a, dtype=complex, copy=True)
        if startder > 0:
            # NEPCore.jl:156-160
            a = np.concatenate([np.zeros(startder, dtype=complex), a])
            V = np.hstack([np.zeros((V.shape[0], startder), dtype=complex), V])
        return self._mlincomb(lam, V, a)

    def compute_Mlincomb_from_MM(self, lam, V, a=None):
        """NEPCore.jl:218-228."""
        V = np.array(V, dtype=complex, copy=True)
        if V.ndim == 1:
            V = V.reshape(-1, 1)
        k = V.shape[1]
        a = np.ones(k, dtype=complex) if a is None else np.array(a, dtype=complex, copy=True)
        z0 = a == 0
        V[:, z0] = 0
        a[z0] = 1
        S = _bidiag(lam, k, (a[1:k] / a[0:k - 1]) * np.arange(1, k))
        z = self.compute_MM(S, V)[:, 0]
        return a[0] * z


class Mder_NEP(NEP):
    """nep_type_helpers.jl:6-12,106-146 (Mder_NEP): a NEP known only through a function lam -> M(lam) (and, optionally,
    derivatives i <= maxder); what test/nleigs/nleigs_nep_types.jl calls "Custom NEP type"."""

    def __init__(self, n, Mder_fun, maxder=0):
        self.n = n
        self.Mder_fun = Mder_fun
        self.maxder = maxder

    def compute_Mder(self, lam, i=0):
        if i > self.maxder:
            raise ValueError("Derivatives higher than %d are not available" % self.maxder)
        return self.Mder_fun(lam) if self.maxder == 0 else self.Mder_fun(lam, i)

    def _mlincomb(self, lam, V, a):
        """NEPCore.jl:164-172 compute_Mlincomb_from_Mder"""
        z = np.zeros(self.n, dtype=complex)
        for j in range(V.shape[1]):
            if a[j] != 0:
                z = z + a[j] * (self.compute_Mder(lam, j) @ V[:, j])
        return np.asarray(z).ravel()


class AbstractSPMF(NEP):
    def compute_Mder(self, lam, i=0):
        """NEPTypes.jl:362-394 (generic AbstractSPMF route)."""
        Av = self.get_Av(); fv = self.get_fv()
        if i == 0:
            x = [f(lam) for f in fv]
        else:
            k = i + 1
            S = _bidiag(lam, k, np.arange(1, k))
            x = [np.asarray(f(S))[-1, 0] for f in fv]
        Z = None
        for A, c in zip(Av, x):
            T = A * c
            Z = T if Z is None else Z + T
        return Z


class SPMF_NEP(AbstractSPMF):
    """NEPTypes.jl:162-237."""

    def __init__(self, AA, fii):
        if len(AA) != len(fii):
            raise ValueError("Inconsistency: Number of supplied matrices = %d but the number of "
                             "supplied functions are = %d" % (len(AA), len(fii)))
        sps = [_issparse(A) for A in AA]
        if not (all(sps) or not any(sps)):
            raise ValueError("Mixing sparse and dense matrices is not allowed in SPMF_NEP.")
        for A in AA[1:]:
            if A.shape != AA[0].shape:
                raise ValueError("The dimensions of the matrices mismatch")
        self.A = [sp.csc_matrix(A) if _issparse(A) else np.asarray(A) for A in AA]
        self.fi = list(fii)
        self.n = AA[0].shape[0]

    def get_Av(self):
        return self.A

    def get_fv(self):
        return self.fi

    def issparse(self):
        return _issparse(self.A[0])

    def compute_MM(self, S, V):
        """NEPTypes.jl:276-319."""
        S = np.atleast_2d(np.asarray(S, dtype=complex))
        V = np.asarray(V, dtype=complex)
        n, p = self.n, S.shape[0]
        Z = np.zeros((n, p), dtype=complex)
        isdiag = np.count_nonzero(S - np.diag(np.diag(S))) == 0
        for A, f in zip(self.A, self.fi):
            if isdiag:
                Sd = np.diag(S)
                Fi = np.diag(np.array([f(s) for s in Sd], dtype=complex))
            else:
                Fi = np.asarray(f(S), dtype=complex)
            Z += _matmul(A, V @ Fi)
        return Z

    def _mlincomb(self, lam, V, a):
        """NEPTypes.jl:972-1011 (compute_Mlincomb!)."""
        n, k = V.shape
        z0 = a == 0
        V[:, z0] = 0
        a[z0] = 1
        z = np.zeros(n, dtype=complex)
        if k == 1:
            for A, f in zip(self.A, self.fi):
                z += _matmul(A, V[:, 0] * f(lam))
            return a[0] * z
        S = _bidiag(lam, k, (a[1:k] / a[0:k - 1]) * np.arange(1, k))
        for A, f in zip(self.A, self.fi):
            Fi1 = np.asarray(f(S), dtype=complex)[:, 0]
            z += _matmul(A, V @ Fi1)
        return a[0] * z


def low_rank_lu_factors(A):
    """rk_nep.jl:70-83 + compactlu :96-100: LU of the dense block that holds the non-zeros of A, keeping only the columns of
    L / rows of U that carry anything; returns L (n x r), U (n x r) with A = L U^H.  The reference destructures
    `L, U = lu(B)` and drops the row permutation (it relies on no interchange taking place, true for the gun W1, W2 and
    for diagonal blocks); here the permutation is applied to L so that A = L U^H always holds."""
    import scipy.linalg as sla
    A = sp.csc_matrix(A)
    n = A.shape[0]
    coo = A.tocoo()
    keep = coo.data != 0
    if not np.any(keep):
        return sp.csc_matrix((n, 0)), sp.csc_matrix((n, 0))
    r0, r1 = coo.row[keep].min(), coo.row[keep].max() + 1
    c0, c1 = coo.col[keep].min(), coo.col[keep].max() + 1
    B = A[r0:r1, c0:c1].toarray()
    Pm, L, U = sla.lu(B)
    m = min(B.shape)
    sel = [i for i in range(m) if np.count_nonzero(L[i:, i]) > 1 or np.count_nonzero(U[i, i:]) > 0]
    Lc = (Pm @ L)[:, sel]; Uc = U[sel, :]
    Lf = sp.lil_matrix((n, len(sel)), dtype=B.dtype); Lf[r0:r1, :] = Lc
    Uf = sp.lil_matrix((n, len(sel)), dtype=B.dtype); Uf[c0:c1, :] = Uc.conj().T
    return sp.csc_matrix(Lf), sp.csc_matrix(Uf)


class LowRankMatrixAndFunction:
    """rk_nep.jl:41-53"""

    def __init__(self, A, f, L=None, U=None):
        self.f = f
        if L is not None and U is not None:
            self.L, self.U = sp.csc_matrix(L), sp.csc_matrix(U)
            self.A = sp.csc_matrix(A) if A is not None and A.shape[0] else sp.csc_matrix(self.L @ self.U.conj().T)
        else:
            self.A = sp.csc_matrix(A)
            self.L, self.U = low_rank_lu_factors(self.A)


class LowRankFactorizedNEP(SPMF_NEP):
    """rk_nep.jl:58-67 (NEPTypes.jl LowRankFactorizedNEP): SPMF whose matrices carry A_i = L_i U_i^H, rank = sum r_i"""

    def __init__(self, Amf):
        super().__init__([M.A for M in Amf], [M.f for M in Amf])
        self.L = [M.L for M in Amf]
        self.U = [M.U for M in Amf]
        self.rank = int(sum(M.U.shape[1] for M in Amf))


class DEP(AbstractSPMF):
    """NEPTypes.jl:427-443."""

    def __init__(self, AA, tauv=(0.0, 1.0)):
        self.A = [sp.csc_matrix(A) if _issparse(A) else
                  np.asarray(A, dtype=complex if np.iscomplexobj(A) else float) for A in AA]   # projected DEPs are complex
        self.tauv = np.array(tauv, dtype=float)
        self.n = AA[0].shape[0]

    def issparse(self):
        return _issparse(self.A[0])

    def compute_Mder(self, lam, i=0):
        """NEPTypes.jl:446-467."""
        n = self.n
        J = sp.identity(n, format="csc") if self.issparse() else np.eye(n)
        M = 0 * J
        if i == 0:
            M = -lam * J
        if i == 1:
            M = -1.0 * J
        for A, tau in zip(self.A, self.tauv):
            a = np.exp(-tau * lam) * (-tau) ** i
            M = M + A * a
        return M

    def compute_MM(self, S, V):
        """NEPTypes.jl:473-483."""
        S = np.atleast_2d(np.asarray(S, dtype=complex)); V = np.asarray(V, dtype=complex)
        Z = -V @ S
        for A, tau in zip(self.A, self.tauv):
            Z = Z + _matmul(A, V @ sla.expm(-tau * S))
        return Z

    def get_Av(self):
        n = self.n
        J = sp.identity(n, format="csc") if self.issparse() else np.eye(n)
        return [J] + list(self.A)

    def get_fv(self):
        """NEPTypes.jl:497-513."""
        fv = [f_neg()]
        for tau in self.tauv:
            fv.append(f_one() if tau == 0 else f_exp(-tau))
        return fv

    def _mlincomb(self, lam, V, a):
        """NEPTypes.jl:940-970."""
        n, k = V.shape
        z = np.zeros(n, dtype=complex)
        for A, tau in zip(self.A, self.tauv):
            w = np.exp(-lam * tau) * np.power(-tau, np.arange(k)).astype(complex)
            z += _matmul(A, V @ (a * w))
        if k == 1:
            z -= a[0] * lam * V[:, 0]
        else:
            z += -lam * a[0] * V[:, 0] - a[1] * V[:, 1]
        return z


class PEP(AbstractSPMF):
    """types_poly.jl:31-98."""

    def __init__(self, AA):
        self.A = [sp.csc_matrix(A) if _issparse(A) else np.asarray(A) for A in AA]
        self.n = AA[0].shape[0]

    def issparse(self):
        return _issparse(self.A[0])

    def compute_MM(self, S, V):
        S = np.atleast_2d(np.asarray(S, dtype=complex)); V = np.asarray(V, dtype=complex)
        Z = np.zeros(V.shape, dtype=complex)
        Si = np.eye(S.shape[0], dtype=complex)
        for A in self.A:
            Z = Z + _matmul(A, V @ Si)
            Si = Si @ S
        return Z

    def compute_Mder(self, lam, i=0):
        Z = 0 * self.A[0]
        for j in range(i + 1, len(self.A) + 1):
            Z = Z + self.A[j - 1] * (lam ** (j - i - 1) * math.factorial(j - 1) / math.factorial(j - i - 1))
        return Z

    def get_Av(self):
        return self.A

    def get_fv(self):
        fv = []
        for i in range(len(self.A)):
            fv.append(f_one() if i == 0 else (f_id() if i == 1 else f_pow(i)))
        return fv

    def _mlincomb(self, lam, V, a):
        """NEPTypes.jl:1016-1045."""
        n = self.n
        z = np.zeros(n, dtype=complex)
        d = len(self.A) - 1
        k = min(V.shape[1], d + 1)
        if lam == 0:
            for j in range(k):
                z += a[j] * math.factorial(j) * _matmul(self.A[j], V[:, j])
        else:
            for j in range(k):
                for i in range(j, d + 1):
                    z += a[j] * lam ** (i - j) * (math.factorial(i) / math.factorial(i - j)) * \
                        _matmul(self.A[i], V[:, j])
        return z


class SumNEP(AbstractSPMF):
    """NEPTypes.jl:845-898 (SPMFSumNEP)."""

    def __init__(self, nep1, nep2):
        self.nep1, self.nep2 = nep1, nep2
        self.n = nep1.n

    def issparse(self):
        return self.nep1.issparse()

    def compute_Mlincomb(self, lam, V, a=None, startder=0):
        # NEPCore.jl:113-125 scales the columns, then NEPTypes.jl:889-890 delegates
        V = np.array(V, dtype=complex, copy=True)
        if V.ndim == 1:
            V = V.reshape(-1, 1)
        if a is not None:
            V = V * np.asarray(a, dtype=complex)[None, :]
        if startder > 0:
            # NEPCore.jl:156-160: zero columns in front (their coefficient is irrelevant)
            V = np.hstack([np.zeros((V.shape[0], startder), dtype=complex), V])
        return self.nep1.compute_Mlincomb(lam, V) + self.nep2.compute_Mlincomb(lam, V)

    def compute_Mder(self, lam, i=0):
        return self.nep1.compute_Mder(lam, i) + self.nep2.compute_Mder(lam, i)

    def compute_MM(self, S, V):
        return self.nep1.compute_MM(S, V) + self.nep2.compute_MM(S, V)

    def get_Av(self):
        return list(self.nep1.get_Av()) + list(self.nep2.get_Av())

    def get_fv(self):
        return list(self.nep1.get_fv()) + list(self.nep2.get_fv())


class DerSPMF(AbstractSPMF):
    """NEPTypes.jl:1055-1160: derivative table fD precomputed at sigma."""

    def __init__(self, spmf, sigma, m):
        self.spmf = spmf
        self.sigma = sigma
        self.n = spmf.n
        fv = spmf.get_fv()
        SS = _bidiag(sigma, 2 * m + 2, np.arange(1, 2 * m + 2))
        self.fD = np.zeros((2 * m + 2, len(fv)), dtype=complex)
        for t, f in enumerate(fv):
            self.fD[:, t] = np.asarray(f(SS), dtype=complex)[:, 0]

    def issparse(self):
        return self.spmf.issparse()

    def get_Av(self):
        return self.spmf.get_Av()

    def get_fv(self):
        return self.spmf.get_fv()

    def compute_Mder(self, lam, i=0):
        return self.spmf.compute_Mder(lam, i)

    def compute_MM(self, S, V):
        return self.spmf.compute_MM(S, V)

    def _mlincomb(self, lam, V, a):
        if lam != self.sigma:
            return self.spmf._mlincomb(lam, V, a)
        n, k = V.shape
        VafD = V @ (a[:, None] * self.fD[:k, :])
        z = np.zeros(n, dtype=complex)
        for j, A in enumerate(self.get_Av()):
            z += _matmul(A, VafD[:, j])
        return z


def shift_and_scale(orgnep, shift=0, scale=1):
    """NEPTransformations.jl:92-105 (SPMF_NEP version)."""
    fv = [f_compose_affine(f, scale, shift) for f in orgnep.get_fv()]
    return SPMF_NEP(orgnep.get_Av(), fv)
