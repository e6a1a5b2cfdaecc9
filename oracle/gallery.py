"""Oracle gallery: problem constructors (test infrastructure only).

Follows
  src/gallery_extra/basic_random_examples.jl:2-9,73-105   (dep0, MSWS RNG)
  src/utils/Serialization.jl:20-31                        (text sparse format)
  src/gallery_extra/NLEVP_native.jl:4-18                  (gun = PEP + SPMF)
  src/gallery_extra/gallery_examples.jl:75-88             (qdep0)
  test/nlar.jl:19-27                                      (gun as shifted/scaled SPMF)
"""
import os
import numpy as np
import scipy.sparse as sp

from . import neps

_HERE = os.path.dirname(os.path.abspath(__file__))
_GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")

_M64 = (1 << 64) - 1
_M128 = (1 << 128) - 1


class MSWS_RNG:
    """Middle-Square-Weyl-Sequence RNG in UInt128 arithmetic.
    basic_random_examples.jl:73-91."""

    def __init__(self, seed=0):
        base = 0x9EF09A97AC0F9ECAEF01C4F2DB0958C9
        self.s = ((seed << 1) + base) & _M128
        self.x = 0x1DE568E1A1CA1B593CBF13F7407CF43E
        self.w = 0xD4AC5C288559E14A5FAFC1B7DF9F9E0E

    def gen_int(self):
        self.x = (self.x * self.x) & _M128
        self.w = (self.w + self.s) & _M128
        self.x = (self.x + self.w) & _M128
        self.x = ((self.x >> 64) | (self.x << 64)) & _M128
        return self.x & _M64

    def gen_float(self):
        # Float64(gen_rng_int(rng)/typemax(UInt64)): UInt64/UInt64 in Julia promotes
        # both to Float64 first, then divides.
        return float(self.gen_int()) / float(_M64)


def gen_rng_mat(rng, n, m):
    """basic_random_examples.jl:93-101 (column-major fill)."""
    A = np.zeros((n, m))
    for c in range(m):
        for r in range(n):
            A[r, c] = 1 - 2 * rng.gen_float()
    return A


def dep0(n=5):
    """basic_random_examples.jl:2-9."""
    rng = MSWS_RNG()
    A0 = gen_rng_mat(rng, n, n)
    A1 = gen_rng_mat(rng, n, n)
    return neps.DEP([A0, A1], [0.0, 1.0])


def read_sparse_matrix(filename):
    """utils/Serialization.jl:20-31; duplicates are summed (Julia `sparse`)."""
    with open(filename) as f:
        data = f.read().split()
    m = int(data[0]); n = int(data[1])
    c = (len(data) - 2) // 3
    I = np.array(data[2:2 + c], dtype=np.int64) - 1
    J = np.array(data[2 + c:2 + 2 * c], dtype=np.int64) - 1
    V = np.array(data[2 + 2 * c:2 + 3 * c], dtype=np.float64)
    return sp.csc_matrix((V, (I, J)), shape=(m, n))


def write_sparse_matrix(filename, M):
    """utils/Serialization.jl:8-17."""
    M = sp.coo_matrix(sp.csc_matrix(M))
    # findnz of a CSC matrix is column-major ordered
    order = np.lexsort((M.row, M.col))
    with open(filename, "w") as f:
        f.write("%d\n%d\n" % M.shape)
        for x in M.row[order] + 1:
            f.write("%d\n" % x)
        for x in M.col[order] + 1:
            f.write("%d\n" % x)
        for x in M.data[order]:
            f.write(repr(float(x)) + "\n")


def load_npz_csc(path, key):
    d = np.load(path)
    return sp.csc_matrix((d[key + "_data"], d[key + "_indices"], d[key + "_indptr"]),
                         shape=tuple(d[key + "_shape"]))


def gun_W():
    """The two gun data files that exist in the reference checkout
    (converted_nlevp/gun_W1.txt, gun_W2.txt), stored as golden data."""
    p = os.path.join(_GOLDEN, "gun_W.npz")
    return load_npz_csc(p, "W1"), load_npz_csc(p, "W2")


GUN_NK = 1.474544889815002e+05   # test/rk_helper/gun_test_utils.jl:50
GUN_NM = 2.726114618171165e-02   # :51
GUN_SIGMA2 = 108.8774


def _onenorm(A):
    return abs(A).sum(axis=0).max()


def gun_standin_KM(nx=76, ny=131):
    """Deterministic gun-like stand-in for the missing gun_K/gun_M blobs
    (SURVEY.md section 8d, C2): 5-point Laplacian K and tensor-product consistent
    mass M on an nx*ny grid (76*131 = 9956), scaled to the reference's gun 1-norms."""
    def T(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])

    def B(n):
        return sp.diags([np.ones(n - 1), 4 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]) / 6.0

    K = sp.kron(sp.identity(nx), T(ny)) + sp.kron(T(nx), sp.identity(ny))
    M = sp.kron(B(nx), B(ny))
    K = sp.csc_matrix(K); M = sp.csc_matrix(M)
    K = K * (GUN_NK / _onenorm(K))
    M = M * (GUN_NM / _onenorm(M))
    return sp.csc_matrix(K), sp.csc_matrix(M)


def gun_matrices(n=9956):
    """[K, M, W1, W2]; real files if NEPMI_GUN_DIR is set, else the stand-in.
    For reduced-size twins (n < 9956, tests) W1/W2 are index-folded into range."""
    W1, W2 = gun_W()
    d = os.environ.get("NEPMI_GUN_DIR")
    if d and n == 9956:
        K = read_sparse_matrix(os.path.join(d, "gun_K.txt"))
        M = read_sparse_matrix(os.path.join(d, "gun_M.txt"))
        return K, M, W1, W2
    if n == 9956:
        K, M = gun_standin_KM()
        return K, M, W1, W2
    # reduced twin: nx*ny = n with ny fixed small
    nx, ny = _twin_grid(n)
    K, M = gun_standin_KM(nx, ny)
    return K, M, _fold(W1, n, tail=True), _fold(W2, n, tail=False)


def _twin_grid(n):
    for ny in (131, 61, 31, 25, 20, 16, 10, 8, 5, 4, 2, 1):
        if n % ny == 0:
            return n // ny, ny
    return n, 1


def _fold(W, n, tail):
    """Shrink a gun W matrix to n x n keeping its pattern shape: W1 lives in the
    last rows/cols, W2 in the first ones."""
    W = sp.coo_matrix(W)
    N = W.shape[0]
    if tail:
        r = W.row - (N - n); c = W.col - (N - n)
    else:
        r = W.row; c = W.col
    keep = (r >= 0) & (r < n) & (c >= 0) & (c < n)
    return sp.csc_matrix((W.data[keep], (r[keep], c[keep])), shape=(n, n))


def nlevp_native_gun(n=9956):
    """NLEVP_native.jl:4-18: SumNEP(PEP([K,-M]), SPMF([W1,W2],[i sqrt(S), i sqrt(S-s2^2 I)]))."""
    K, M, W1, W2 = gun_matrices(n)
    pep = neps.PEP([K, -M])
    sq = neps.SPMF_NEP([W1, W2], [neps.f_isqrt(0.0), neps.f_isqrt(-GUN_SIGMA2 ** 2)])
    return neps.SumNEP(pep, sq)


GUN_SHIFT = 250.0 ** 2
GUN_SCALE = 330.0 ** 2 - 220.0 ** 2


def gun_spmf_scaled(n=9956):
    """test/nlar.jl:26-27: SPMF_NEP(get_Av,get_fv) then shift_and_scale."""
    nep = nlevp_native_gun(n)
    spmf = neps.SPMF_NEP(nep.get_Av(), nep.get_fv())
    return neps.shift_and_scale(spmf, shift=GUN_SHIFT, scale=GUN_SCALE)


def gun_spmf(n=9956):
    nep = nlevp_native_gun(n)
    return neps.SPMF_NEP(nep.get_Av(), nep.get_fv())


def qdep0():
    """gallery_examples.jl:75-88 (data: converted_misc/qdep_infbilanczos_A{0,1}.txt)."""
    p = os.path.join(_GOLDEN, "qdep0.npz")
    A0 = load_npz_csc(p, "A0"); A1 = load_npz_csc(p, "A1")
    n = A0.shape[0]
    AA = [-sp.identity(n, format="csc"), A0, A1]
    fi = [neps.f_pow(2), neps.f_one(), neps.f_exp(-1.0)]
    return neps.SPMF_NEP(AA, fi)


def dep_symm_double(n=100):
    """src/gallery_extra/gallery_examples.jl:15-30: DEP with sparse symmetric matrices, double eigenvalues, tau = 2"""
    import scipy.sparse as sp
    from . import neps
    LL = -sp.diags(2 * np.ones(n)) + sp.diags(np.ones(n - 1), -1) + sp.diags(np.ones(n - 1), 1)
    x = np.linspace(0, np.pi, n)
    h = x[1] - x[0]
    LL = sp.kron(LL / h ** 2, LL / h ** 2)
    bb = -100 * np.abs(np.sin(x[:, None] + x[None, :]))
    aa = 8 * np.sin(x)[:, None] * np.sin(x)[None, :]
    B = sp.diags(bb.reshape(-1, order="F"))
    A = LL + sp.diags(aa.reshape(-1, order="F"))
    return neps.DEP([sp.csc_matrix(A), sp.csc_matrix(B)], [0.0, 2.0])


def pep0(n=200):
    """basic_random_examples.jl:36-44"""
    rng = MSWS_RNG()
    return neps.PEP([gen_rng_mat(rng, n, n) for _ in range(3)])


def _f_exp_sqrt(m, coeff, below):
    """test/nleigs/particle_test_utils.jl:148-155: exp(i sqrt(m (lam - c)))  (branch points below the interval) or
    exp(-sqrt(m (c - lam)))  (from the interval on)"""
    import scipy.linalg as sla

    def f(S):
        if neps._ismat(S):
            I = np.eye(S.shape[0])
            if below:
                return sla.expm(1j * sla.sqrtm((m * (S.astype(complex) - coeff * I))))
            return sla.expm(-sla.sqrtm((m * (-S.astype(complex) + coeff * I))))
        z = complex(S)
        return np.exp(1j * np.sqrt(m * (z - coeff))) if below else np.exp(-np.sqrt(m * (-z + coeff)))
    return f


def particle_nep(interval):
    """test/nleigs/particle_test_utils.jl:37-165 ("particle in a canyon", after W. Vandenberghe): H - lam I plus one
    rank-2 term per branch point, given by its factors only; returns (nep, brpts, U0)"""
    meter = 1 / 5.2917725e-11
    nm = 1e-9 * meter
    eV = 1 / 13.6
    xmax, zmax, xstep, zstep = 5, 2, 0.05, 0.05
    x_x = np.arange(-xmax, xmax + xstep / 2, xstep) * nm
    z_z = np.arange(-zmax, zmax + zstep / 2, zstep) * nm
    nx, nz = len(x_x), len(z_z)
    dx = np.min(np.diff(x_x)); dz = np.min(np.diff(z_z))
    x = np.kron(x_x, np.ones(nz)); z = np.kron(np.ones(nx), z_z)
    w1, w2, l, U0 = 1 * nm, 1.1 * nm, 4 * nm, 3 * eV
    U = np.zeros(len(x))
    U[np.abs(z) < w1] = -U0
    U[(np.abs(z) < w2) & (np.abs(x) < l / 2)] = -U0
    m = 0.2
    n = nx * nz
    tri = lambda k, d0, d1: np.diag(np.full(k, d0)) + np.diag(np.full(k - 1, d1), 1) + np.diag(np.full(k - 1, d1), -1)
    Dxx_x = tri(nx, -2 / dx ** 2, 1 / dx ** 2)
    Dzz_z = tri(nz, -2 / dx ** 2, 1 / dz ** 2)
    H_L = -1 / m * Dzz_z + np.diag(U[:nz])
    H_R = -1 / m * Dzz_z + np.diag(U[-nz:])
    if np.linalg.norm(H_L - H_R, 2) != 0:
        raise NotImplementedError("asymmetric potential (not reached by the reference's parameters)")
    D, V = np.linalg.eigh(H_L)
    i = np.argsort(D, kind="stable")
    d = D[i]; V = V[:, i]
    H = -1 / m * (sp.kron(sp.csc_matrix(Dxx_x), sp.identity(nz)) + sp.kron(sp.identity(nx), sp.csc_matrix(Dzz_z))) + sp.diags(U)
    brpts = []
    SL = []
    for j in range(len(d)):                                   # p[j] == 0: left and right column for every eigenvector
        cols = np.zeros((n, 2)); cols[:nz, 0] = V[:, j]; cols[n - nz:, 1] = V[:, j]
        if j > 0 and d[j - 1] == d[j]:
            SL[-1] = np.hstack([SL[-1], cols])
        else:
            SL.append(cols); brpts.append(d[j])
    brpts = np.array(brpts)
    SU = [sp.csc_matrix(S) for S in SL]
    SL = [sp.csc_matrix(-1 / m / dx ** 2 * S) for S in SL]
    f = [_f_exp_sqrt(m, brpts[j], j < interval - 1) for j in range(len(brpts))]
    C = [neps.LowRankMatrixAndFunction(None, f[k], L=SL[k], U=SU[k]) for k in range(len(f))]
    nep = neps.SumNEP(neps.PEP([sp.csc_matrix(H), -sp.identity(n, format="csc")]), neps.LowRankFactorizedNEP(C))
    return nep, brpts, U0


def particle_init(interval):
    """test/nleigs/particle_test_utils.jl:7-35: (nep, Sigma, Xi, v, nodes, xmin, xmax)"""
    nep, brpts, U0 = particle_nep(interval)
    sep = 1e-4
    if interval == 1:
        xmin = -U0; xmax = brpts[0] - sep
        Xi = 10.0 ** np.linspace(-6, 6, 10000) + brpts[0]
    elif interval > 1:
        xmin = brpts[interval - 2] + sep; xmax = brpts[interval - 1] - sep
        Xi = np.concatenate([-10.0 ** np.linspace(-6, 6, 5000) + brpts[interval - 2], 10.0 ** np.linspace(-6, 6, 5000) + brpts[interval - 1]])
    else:
        raise ValueError("Invalid interval: %d" % interval)
    Sigma = np.array([xmin + 0j, xmax + 0j])
    A0 = pep0(200).get_Av()[0]
    v = np.concatenate([A0[:, :81].reshape(-1, order="F"), A0[:81, 81]]).astype(complex)
    nodes = np.linspace(xmin, xmax, 11)[1::2] + 0j
    return nep, Sigma, Xi, v, nodes, xmin, xmax
