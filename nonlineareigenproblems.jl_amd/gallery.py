"""Problem gallery for the backend (inputs only): `nep_gallery(name, ...)` mirrors src/Gallery.jl:193-221
for the problems on the hot path.

  "dep0"              src/gallery_extra/basic_random_examples.jl:2-9 (MSWS RNG :73-105)
  "qdep0"             src/gallery_extra/gallery_examples.jl:75-88 (matrices: data/qdep0.npz)
  "nlevp_native_gun"  src/gallery_extra/NLEVP_native.jl:4-18.  gun_K/gun_M are missing from the
                      reference checkout (.MISSING_LARGE_BLOBS); they are read from $NEPMI_GUN_DIR in
                      the reference's text format (src/utils/Serialization.jl:20-31) when given,
                      otherwise a deterministic gun-like stand-in with the reference's 1-norms is used
                      (SURVEY.md section 8d C2).  W1, W2 are the reference's data (data/gun_W.npz).
  "gun_spmf" / "gun_spmf_scaled"   the SPMF form used with Krylov methods (test/nlar.jl:26-27)
"""
import os

import numpy as np
import scipy.sparse as sp

from . import funcs
from .nep import DEP, PEP, SPMF_NEP, SumNEP, LowRankMatrixAndFunction, LowRankFactorizedNEP, shift_and_scale

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_M64 = (1 << 64) - 1
_M128 = (1 << 128) - 1


class MSWS_RNG:
    """Middle Square Weyl Sequence RNG, UInt128 arithmetic (basic_random_examples.jl:73-91)."""

    def __init__(self, seed=0):
        self.s = ((seed << 1) + 0x9EF09A97AC0F9ECAEF01C4F2DB0958C9) & _M128
        self.x = 0x1DE568E1A1CA1B593CBF13F7407CF43E
        self.w = 0xD4AC5C288559E14A5FAFC1B7DF9F9E0E

    def gen_rng_int(self):
        self.x = (self.x * self.x) & _M128
        self.w = (self.w + self.s) & _M128
        self.x = (self.x + self.w) & _M128
        self.x = ((self.x >> 64) | (self.x << 64)) & _M128
        return self.x & _M64

    def gen_rng_float(self):
        return float(self.gen_rng_int()) / float(_M64)


def gen_rng_mat(rng, n, m):
    A = np.zeros((n, m))
    for c in range(m):
        for r in range(n):
            A[r, c] = 1 - 2 * rng.gen_rng_float()
    return A


def read_sparse_matrix(filename):
    """src/utils/Serialization.jl:20-31"""
    with open(filename) as f:
        data = f.read().split()
    m = int(data[0]); n = int(data[1])
    c = (len(data) - 2) // 3
    I = np.array(data[2:2 + c], dtype=np.int64) - 1
    J = np.array(data[2 + c:2 + 2 * c], dtype=np.int64) - 1
    V = np.array(data[2 + 2 * c:2 + 3 * c], dtype=np.float64)
    return sp.csc_matrix((V, (I, J)), shape=(m, n))


def write_sparse_matrix(filename, M):
    """src/utils/Serialization.jl:8-17"""
    M = sp.coo_matrix(sp.csc_matrix(M))
    order = np.lexsort((M.row, M.col))
    with open(filename, "w") as f:
        f.write("%d\n%d\n" % M.shape)
        for x in M.row[order] + 1:
            f.write("%d\n" % x)
        for x in M.col[order] + 1:
            f.write("%d\n" % x)
        for x in M.data[order]:
            f.write(repr(float(x)) + "\n")


def _load_csc(path, key):
    d = np.load(path)
    return sp.csc_matrix((d[key + "_data"], d[key + "_indices"], d[key + "_indptr"]), shape=tuple(d[key + "_shape"]))


GUN_NK = 1.474544889815002e+05   # test/rk_helper/gun_test_utils.jl:50
GUN_NM = 2.726114618171165e-02   # :51
GUN_SIGMA2 = 108.8774
GUN_SHIFT = 250.0 ** 2           # test/nlar.jl:19-20
GUN_SCALE = 330.0 ** 2 - 220.0 ** 2


def _onenorm(A):
    return abs(A).sum(axis=0).max()


def gun_standin_KM(nx=76, ny=131):
    def T(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])

    def B(n):
        return sp.diags([np.ones(n - 1), 4 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]) / 6.0

    K = sp.csc_matrix(sp.kron(sp.identity(nx), T(ny)) + sp.kron(T(nx), sp.identity(ny)))
    M = sp.csc_matrix(sp.kron(B(nx), B(ny)))
    K = sp.csc_matrix(K * (GUN_NK / _onenorm(K)))
    M = sp.csc_matrix(M * (GUN_NM / _onenorm(M)))
    return K, M


def _twin_grid(n):
    for ny in (131, 61, 31, 25, 20, 16, 10, 8, 5, 4, 2, 1):
        if n % ny == 0:
            return n // ny, ny
    return n, 1


def _fold(W, n, tail):
    W = sp.coo_matrix(W)
    N = W.shape[0]
    r = W.row - (N - n) if tail else W.row
    c = W.col - (N - n) if tail else W.col
    keep = (r >= 0) & (r < n) & (c >= 0) & (c < n)
    return sp.csc_matrix((W.data[keep], (r[keep], c[keep])), shape=(n, n))


def gun_matrices(n=9956):
    p = os.path.join(_DATA, "gun_W.npz")
    W1 = _load_csc(p, "W1"); W2 = _load_csc(p, "W2")
    d = os.environ.get("NEPMI_GUN_DIR")
    if n == 9956 and d:
        K = read_sparse_matrix(os.path.join(d, "gun_K.txt"))
        M = read_sparse_matrix(os.path.join(d, "gun_M.txt"))
        _check_gun_files(K, M)
        return K, M, W1, W2
    if n == 9956:
        K, M = gun_standin_KM()
        return K, M, W1, W2
    nx, ny = _twin_grid(n)
    K, M = gun_standin_KM(nx, ny)
    return K, M, _fold(W1, n, True), _fold(W2, n, False)


def _check_gun_files(K, M):
    """files given through NEPMI_GUN_DIR must be THE gun matrices: shape 9956 x 9956 and the 1-norms the reference pins in
    test/rk_helper/gun_test_utils.jl:50-51 (a wrong or truncated file would otherwise give a silently different problem)"""
    for name, A, ref in (("gun_K.txt", K, GUN_NK), ("gun_M.txt", M, GUN_NM)):
        if A.shape != (9956, 9956):
            raise ValueError("NEPMI_GUN_DIR/%s: shape %s, expected (9956, 9956)" % (name, A.shape))
        nrm = _onenorm(A)
        if not abs(nrm - ref) <= 1e-12 * ref:
            raise ValueError("NEPMI_GUN_DIR/%s: 1-norm %.16e differs from the reference value %.16e "
                             "(test/rk_helper/gun_test_utils.jl:50-51)" % (name, nrm, ref))


def nlevp_native_gun(n=9956):
    K, M, W1, W2 = gun_matrices(n)
    pep = PEP([K, -M])
    sqrtnep = SPMF_NEP([W1, W2], [funcs.ISqrt(1.0, 0.0), funcs.ISqrt(1.0, -GUN_SIGMA2 ** 2)])
    return SumNEP(pep, sqrtnep)


def gun_spmf(n=9956):
    nep = nlevp_native_gun(n)
    return SPMF_NEP(nep.get_Av(), nep.get_fv())


def gun_spmf_scaled(n=9956):
    return shift_and_scale(gun_spmf(n), shift=GUN_SHIFT, scale=GUN_SCALE)


def dep0(n=5):
    rng = MSWS_RNG()
    A0 = gen_rng_mat(rng, n, n)
    A1 = gen_rng_mat(rng, n, n)
    return DEP([A0, A1], [0.0, 1.0])


def qdep0():
    p = os.path.join(_DATA, "qdep0.npz")
    A0 = _load_csc(p, "A0"); A1 = _load_csc(p, "A1")
    n = A0.shape[0]
    return SPMF_NEP([-sp.identity(n, format="csc"), A0, A1], [funcs.Monomial(2), funcs.one(), funcs.Exp(-1.0)])


def _wep(**kw):
    from .wep import WEP
    return WEP(**kw)


GALLERY = {
    "WEP": _wep,
    "dep0": dep0,
    "qdep0": qdep0,
    "nlevp_native_gun": nlevp_native_gun,
    "gun_spmf": gun_spmf,
    "gun_spmf_scaled": gun_spmf_scaled,
}


def dep_symm_double(n=100):
    """src/gallery_extra/gallery_examples.jl:15-30: DEP with sparse symmetric matrices, double eigenvalues, tau = 2"""
    LL = -sp.diags(2 * np.ones(n)) + sp.diags(np.ones(n - 1), -1) + sp.diags(np.ones(n - 1), 1)
    x = np.linspace(0, np.pi, n)
    h = x[1] - x[0]
    LL = sp.kron(LL / h ** 2, LL / h ** 2)
    bb = -100 * np.abs(np.sin(x[:, None] + x[None, :]))
    aa = 8 * np.sin(x)[:, None] * np.sin(x)[None, :]
    B = sp.diags(bb.reshape(-1, order="F"))
    A = LL + sp.diags(aa.reshape(-1, order="F"))
    return DEP([sp.csc_matrix(A), sp.csc_matrix(B)], [0.0, 2.0])


GALLERY["dep_symm_double"] = dep_symm_double


def pep0(n=200):
    """src/gallery_extra/basic_random_examples.jl:36-44"""
    rng = MSWS_RNG()
    return PEP([gen_rng_mat(rng, n, n) for _ in range(3)])


GALLERY["pep0"] = pep0


def particle_nep(interval):
    """The "particle in a canyon" problem of test/nleigs/particle_test_utils.jl:37-165 (after W. Vandenberghe): a 2-D
    Schroedinger operator H - lam I on a 201 x 81 grid plus, per branch point (eigenvalue of the lead Hamiltonian), a
    rank-2 boundary term with the wave-number function exp(+-sqrt(...)) -- 83 terms, the nonlinear ones given by their
    factors L_k, U_k only.  Returns (nep, brpts, U0); nep = PEP + LowRankFactorizedNEP (n = 16281, r = 162)."""
    meter = 1 / 5.2917725e-11
    nm = 1e-9 * meter
    eV = 1 / 13.6
    xmax, zmax, xstep, zstep = 5, 2, 0.05, 0.05
    x_x = np.arange(-xmax, xmax + xstep / 2, xstep) * nm
    z_z = np.arange(-zmax, zmax + zstep / 2, zstep) * nm
    nx, nz = len(x_x), len(z_z)
    dx = np.min(np.diff(x_x)); dz = np.min(np.diff(z_z))
    xg = np.kron(x_x, np.ones(nz)); zg = np.kron(np.ones(nx), z_z)
    w1, w2, ell, U0 = 1 * nm, 1.1 * nm, 4 * nm, 3 * eV
    U = np.zeros(len(xg))
    U[np.abs(zg) < w1] = -U0
    U[(np.abs(zg) < w2) & (np.abs(xg) < ell / 2)] = -U0
    m = 0.2
    n = nx * nz
    tri = lambda k, d0, d1: sp.diags([np.full(k - 1, d1), np.full(k, d0), np.full(k - 1, d1)], [-1, 0, 1], format="csc")
    Dxx = tri(nx, -2 / dx ** 2, 1 / dx ** 2)
    Dzz = tri(nz, -2 / dx ** 2, 1 / dz ** 2)            # (the reference scales the diagonal with dx, :80)
    H_L = (-1 / m * Dzz + sp.diags(U[:nz])).toarray()
    H_R = (-1 / m * Dzz + sp.diags(U[-nz:])).toarray()
    if np.linalg.norm(H_L - H_R, 2) != 0:
        raise NotImplementedError("asymmetric lead potential (not reached with the reference's parameters)")
    d, V = np.linalg.eigh(H_L)
    order = np.argsort(d, kind="stable")
    d = d[order]; V = V[:, order]
    H = -1 / m * (sp.kron(Dxx, sp.identity(nz)) + sp.kron(sp.identity(nx), Dzz)) + sp.diags(U)
    brpts, SL = [], []
    for j in range(len(d)):                                 # symmetric leads: a left and a right column per lead mode
        cols = np.zeros((n, 2)); cols[:nz, 0] = V[:, j]; cols[n - nz:, 1] = V[:, j]
        if j > 0 and d[j - 1] == d[j]:
            SL[-1] = np.hstack([SL[-1], cols])
        else:
            SL.append(cols); brpts.append(d[j])
    brpts = np.array(brpts)
    fv = [funcs.ExpSqrt(1j, m, -m * c) if j < interval - 1 else funcs.ExpSqrt(-1.0, -m, m * c) for j, c in enumerate(brpts)]
    C = [LowRankMatrixAndFunction(None, fv[k], L=sp.csc_matrix(-1 / m / dx ** 2 * SL[k]), U=sp.csc_matrix(SL[k]))
         for k in range(len(fv))]
    nep = SumNEP(PEP([sp.csc_matrix(H), -sp.identity(n, format="csc")]), LowRankFactorizedNEP(C))
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
        Xi = np.concatenate([-10.0 ** np.linspace(-6, 6, 5000) + brpts[interval - 2],
                             10.0 ** np.linspace(-6, 6, 5000) + brpts[interval - 1]])
    else:
        raise ValueError("Invalid interval: %d" % interval)
    A0 = pep0(200).get_Av()[0]
    v = np.concatenate([A0[:, :81].reshape(-1, order="F"), A0[:81, 81]]).astype(complex)
    nodes = np.linspace(xmin, xmax, 11)[1::2] + 0j
    return nep, np.array([xmin + 0j, xmax + 0j]), Xi, v, nodes, xmin, xmax


def nep_gallery(name, *args, **kwargs):
    if name not in GALLERY:
        raise KeyError("unknown gallery problem %r (available: %s)" % (name, ", ".join(sorted(GALLERY))))
    return GALLERY[name](*args, **kwargs)
