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
