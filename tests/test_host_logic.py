"""CPU tests of the product's host logic (no GPU compute): closed-form derivative tables against the
oracle's matrix-function route, gallery equality, C-ABI symbol export, coefficient blocks."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import nep_amd as na
from oracle import gallery as og, neps as oneps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "nepmi355.h")).read()
    names = set(re.findall(r"\b(nep_[a-z0-9_]+)\s*\(", hdr))
    names -= {"nep_cdouble"}
    assert len(names) >= 30
    lib = ctypes.CDLL(na.LIB_PATH)
    for nme in sorted(names):
        assert hasattr(lib, nme), "missing export " + nme
    assert set(na._lib.SIGNATURES) == names          # the ctypes binding covers the whole header
    assert lib.nep_version() == 101
    # the binary names the sources it was built from; the binding refuses a library whose digest differs (_lib._load)
    from nep_amd import build
    assert na._lib.lib.nep_src_digest().decode() == build.source_digest() == build.built_digest()
    assert not build.needs_build()


def test_no_cpu_fallback_without_gpu():
    if na.device_count() > 0:
        pytest.skip("GPU present")
    nep = na.nep_gallery("dep0")
    with pytest.raises((na.NepError, RuntimeError)):
        nep.compute_Mlincomb(1.0, np.ones(5))
    with pytest.raises((na.NepError, RuntimeError)):
        na.iar(nep, v=np.ones(5))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nonlineareigenproblems.jl_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.parametrize("lam", [0.0, 0.3 - 0.2j, 1.5])
def test_gun_scaled_derivative_table_vs_matrix_function(lam):
    """closed form (product) vs f(S)[:,1] of the bidiagonal matrix (oracle = reference route,
    src/NEPTypes.jl:1108-1128) for the shifted/scaled gun functions, 42 derivatives."""
    m = 20
    o = oneps.DerSPMF(og.gun_spmf_scaled(655), lam, m)
    p = na.nep_gallery("gun_spmf_scaled", 655)
    fD = np.column_stack([f.derivs(lam, 2 * m + 2) for f in p.get_fv()])
    assert fD.shape == o.fD.shape
    rel = np.abs(fD - o.fD) / np.maximum(np.abs(o.fD), 1e-300)
    rel[np.abs(o.fD) < 1e-250] = 0
    assert rel.max() < 1e-9


def test_derivative_table_dynamic_range():
    p = na.nep_gallery("gun_spmf_scaled", 655)
    d = np.column_stack([f.derivs(0.0, 101) for f in p.get_fv()])
    assert np.all(np.isfinite(d))
    assert 1e160 < np.abs(d).max() < 1e170          # 3.1e164 at order 100


@pytest.mark.parametrize("name", ["Exp", "Monomial", "ISqrt", "Affine", "WEPSqrt"])
def test_funcs_derivs_vs_numeric(name):
    f = {"Exp": na.funcs.Exp(-0.7), "Monomial": na.funcs.Monomial(5), "ISqrt": na.funcs.ISqrt(2.0, -3.0 + 1j),
         "Affine": na.funcs.ISqrt(1.0, 0.5).affine(3.0, 0.25 + 0.5j),
         "WEPSqrt": na.funcs.WEPSqrt(0.3 + 0.1j, 2.0 + 0.7j, 0.4)}[name]
    lam = 0.8 + 0.3j
    d = f.derivs(lam, 4)
    h = 0.05
    # Cauchy-integral style numerical derivatives on a small circle
    N = 64
    th = 2 * np.pi * np.arange(N) / N
    vals = np.array([f(lam + h * np.exp(1j * t)) for t in th])
    import math
    for j in range(4):
        num = math.factorial(j) * np.mean(vals * np.exp(-1j * j * th)) / h ** j
        assert abs(num - d[j]) <= 1e-6 * max(1.0, abs(d[j]))


def test_gallery_matches_oracle():
    a = na.nep_gallery("dep0"); b = og.dep0()
    for X, Y in zip(a.A, b.A):
        assert np.array_equal(X, Y)
    assert a.compute_Mder(3.0)[0, 0].real == pytest.approx(-2.942777908030041, abs=1e-15)
    ga = na.nep_gallery("nlevp_native_gun"); gb = og.nlevp_native_gun()
    for X, Y in zip(ga.get_Av(), gb.get_Av()):
        assert (sp.csc_matrix(X) != sp.csc_matrix(Y)).nnz == 0
    lam = 250.0 ** 2 + 3j
    D = sp.csc_matrix(ga.compute_Mder(lam)) - sp.csc_matrix(gb.compute_Mder(lam))
    assert abs(D).max() <= 1e-12 * abs(sp.csc_matrix(gb.compute_Mder(lam))).max()
    D1 = sp.csc_matrix(ga.compute_Mder(lam, 1)) - sp.csc_matrix(gb.compute_Mder(lam, 1))
    assert abs(D1).max() <= 1e-12
    q = na.nep_gallery("qdep0")
    assert q.n == 1000 and [A.nnz for A in q.get_Av()] == [1000, 9945, 9963]


def test_coeff_block_matches_reference_identity():
    """C[j,i] = a_j f_i^(j-1)(lam) equals a_1 * f_i(S)[:,1] of the reference's bidiagonal S
    (src/NEPTypes.jl:981-1010) wherever a has no zeros."""
    p = na.nep_gallery("gun_spmf_scaled", 655)
    o = og.gun_spmf_scaled(655)
    rng = np.random.default_rng(0)
    k = 6
    a = rng.standard_normal(k) + 0j
    lam = 0.2 + 0.1j
    Cm = p.coeff_block(lam, a)
    S = oneps._bidiag(lam, k, (a[1:k] / a[0:k - 1]) * np.arange(1, k))
    for i, f in enumerate(o.get_fv()):
        ref = a[0] * np.asarray(f(S))[:, 0]
        assert np.allclose(Cm[:, i], ref, rtol=1e-10)
    a[2] = 0
    assert np.all(p.coeff_block(lam, a)[2] == 0)


def test_serialization_roundtrip(tmp_path):
    # test/serialization.jl:5-17
    A = sp.random(30, 20, 0.2, random_state=1, format="csc")
    fn = str(tmp_path / "m.txt")
    na.gallery.write_sparse_matrix(fn, A)
    B = na.gallery.read_sparse_matrix(fn)
    assert (A != B).nnz == 0
    assert (og.read_sparse_matrix(fn) != A).nnz == 0


def test_spmf_constructor_errors():
    A = sp.identity(3, format="csc")
    with pytest.raises(ValueError):
        na.SPMF_NEP([A, A], [na.funcs.one()])
    with pytest.raises(ValueError):
        na.SPMF_NEP([A, np.eye(3)], [na.funcs.one(), na.funcs.ident()])
    with pytest.raises(ValueError):
        na.SPMF_NEP([A, sp.identity(4, format="csc")], [na.funcs.one(), na.funcs.ident()])


def test_rk_helper_matches_oracle():
    """host setup of nleigs (polygon discretisation, Leja-Bagby points, scalar divided differences, point-in-polygon)"""
    from oracle import nleigs as onl
    rk = na.rk_helper
    Sigma = np.array([-10 - 2j, 10 - 2j, 10 + 2j, -10 + 2j])
    g1, z1 = rk.discretizepolygon(Sigma, True); g2, z2 = onl.discretizepolygon(Sigma, True)
    assert np.array_equal(g1, g2) and np.array_equal(z1, z2)
    a1, b1, c1 = rk.lejabagby(g1, np.array([np.inf]), g1, 12, False, 2); a2, b2, c2 = onl.lejabagby(g2, np.array([np.inf]), g2, 12, False, 2)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(c1, c2)
    fv_p = [na.funcs.one(), na.funcs.ident(), na.funcs.ISqrt(1.0, 0.0)]
    fv_o = [oneps.f_one(), oneps.f_id(), oneps.f_isqrt(0.0)]
    sig = 62500 + 1e4 * np.exp(2j * np.pi * np.arange(8) / 8); xi = -np.logspace(0, 3, 8); be = np.linspace(1, 2, 8)
    s1 = rk.scgendivdiffs(sig, xi, be, fv_p); s2 = onl.scgendivdiffs(sig, xi, be, 6, fv_o)
    assert np.allclose(s1, s2, rtol=1e-10, atol=1e-14 * abs(s2).max())
    pts = np.array([0, 9.99 + 1.99j, 10 + 2j, 10.01, -10 - 2j, 3 - 2j, 3 - 2.0001j])
    assert list(rk.in_Sigma(pts, Sigma, 1e-10)) == list(onl.in_Sigma(pts, Sigma, 1e-10)) == [True, True, True, False, True, True, False]
    assert rk.rk_structure(na.nep_gallery("nlevp_native_gun", 655)) == (1, 2)
    assert rk.rk_structure(na.PEP([np.eye(2)] * 3)) == (2, 0)
    assert rk.rk_structure(na.nep_gallery("dep0")) == (-1, 3)


def test_hostlu_factor_strategy():
    """UMFPACK-like strategy selection and factor layout (torch-free worker module)"""
    import nep_amd_hostlu as hl
    import scipy.sparse.linalg as spla
    A = sp.csc_matrix(og.gun_spmf(655).compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    F = hl.factor(A.data, A.indices, A.indptr, A.shape)
    assert F["strategy"]["symmetric_mode"] and F["strategy"]["permc_spec"] == "MMD_AT_PLUS_A"
    n = A.shape[0]
    assert F["fmt"] == "csc"                      # SuperLU's own layout goes to nep_lu_create_csc unconverted
    L = sp.csc_matrix((F["Lx"], F["Li"], F["Lp"]), shape=(n, n)); U = sp.csc_matrix((F["Ux"], F["Ui"], F["Up"]), shape=(n, n))
    Pr = sp.csc_matrix((np.ones(n), (F["perm_r"], np.arange(n)))); Pc = sp.csc_matrix((np.ones(n), (np.arange(n), F["perm_c"])))
    assert abs(Pr @ A @ Pc - L @ U).max() <= 1e-10 * abs(A).max()
    Fr = hl.factor(A.data, A.indices, A.indptr, A.shape, csr=True)
    Lr = sp.csr_matrix((Fr["Lx"], Fr["Li"], Fr["Lp"]), shape=(n, n)); Ur = sp.csr_matrix((Fr["Ux"], Fr["Ui"], Fr["Up"]), shape=(n, n))
    # second factorisation of the pattern: the cached column ordering is used (matrix handed over pre-permuted, NATURAL);
    # same pivots and structure, values equal up to the summation order inside a column
    assert np.array_equal(Fr["perm_r"], F["perm_r"]) and np.array_equal(Fr["perm_c"], F["perm_c"])
    assert Fr["fmt"] == "csr" and abs(Lr - L).max() <= 1e-13 * abs(L).max() and abs(Ur - U).max() <= 1e-13 * abs(U).max()
    B = sp.csc_matrix(sp.triu(A, 0) + sp.identity(n))                     # upper triangular pattern: symmetry 0
    F2 = hl.factor(B.data, B.indices, B.indptr, B.shape)
    assert not F2["strategy"]["symmetric_mode"] and F2["strategy"]["permc_spec"] == "COLAMD"
    Bz = A.copy().tolil(); Bz[3, 3] = 0.0; Bz = sp.csc_matrix(Bz); Bz.eliminate_zeros()   # zero on the diagonal
    assert not hl.pattern_symmetric(Bz)


def _ml_reference_partition(n, L, U, bmax):
    """NumPy restatement of the symbolic analysis of csrc/trsv_ml.hip (elimination tree of struct(L)+struct(U)^T by
    Liu's algorithm, multilevel partition into subtrees of at most bmax nodes): levels, block ids"""
    Lr = sp.csr_matrix(L); UT = sp.csr_matrix(sp.csc_matrix(U).T)
    parent = np.full(n, -1); anc = np.full(n, -1)
    for i in range(n):
        for M in (Lr, UT):
            for k in M.indices[M.indptr[i]:M.indptr[i + 1]]:
                while 0 <= k < i:
                    nx = anc[k]; anc[k] = i
                    if nx < 0:
                        parent[k] = i
                        break
                    k = nx
    lvl = np.zeros(n, int); rsz = np.ones(n, int); pmax = np.full(n, -1); psum = np.zeros(n, int)
    for j in range(n):
        M = max(pmax[j], 0); s_ = psum[j] if pmax[j] >= 0 else 0
        if s_ + 1 <= bmax:
            lvl[j], rsz[j] = M, s_ + 1
        else:
            lvl[j], rsz[j] = M + 1, 1
        p = parent[j]
        if p >= 0:
            if lvl[j] > pmax[p]:
                pmax[p], psum[p] = lvl[j], rsz[j]
            elif lvl[j] == pmax[p]:
                psum[p] += rsz[j]
    bid = np.arange(n)
    for j in range(n - 1, -1, -1):
        if parent[j] >= 0 and lvl[parent[j]] == lvl[j]:
            bid[j] = bid[parent[j]]
    return parent, lvl, bid


@pytest.mark.parametrize("case", ["gun_sym", "random_unsym", "chain"])
def test_lu_block_schedule_analysis(case, monkeypatch):
    """host-only symbolic analysis of the K5 block schedule (nep_lu_analyze, no device): level / block counts equal the
    NumPy restatement; every dependency of L (U) stays inside its diagonal block or points to an earlier (later) level;
    the block solve built from that partition reproduces the direct solution"""
    import ctypes as C
    import nep_amd_hostlu as hl
    monkeypatch.setenv("NEP_ML_BMAX", "128")
    rng = np.random.default_rng(3)
    if case == "gun_sym":
        A = sp.csc_matrix(og.gun_spmf(1310).compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    elif case == "random_unsym":
        n0 = 900
        A = (sp.random(n0, n0, 0.004, random_state=rng, format="csc") + sp.diags(0.05 + rng.random(n0))
             + 1j * sp.random(n0, n0, 0.002, random_state=rng, format="csc")).tocsc()
    else:
        A = sp.diags([np.ones(699), 4 * np.ones(700), np.ones(699)], [-1, 0, 1], format="csc").astype(complex)
    F = hl.factor(A.data, A.indices, A.indptr, A.shape)
    n = F["n"]
    out = (C.c_int64 * 8)()
    L_ = na._lib.lib
    hp = na._lib.hptr
    assert L_.nep_lu_analyze(n, 1, hp(F["Lp"]), hp(F["Li"]), hp(F["Up"]), hp(F["Ui"]), out) == 0
    L = sp.csc_matrix((F["Lx"], F["Li"], F["Lp"]), shape=(n, n)); U = sp.csc_matrix((F["Ux"], F["Ui"], F["Up"]), shape=(n, n))
    parent, lvl, bid = _ml_reference_partition(n, L, U, 128)
    assert out[0] == lvl.max() + 1 and out[1] == len(np.unique(bid)) and out[2] == np.bincount(bid).max() <= 128
    # CSR input gives the same partition
    Lr = sp.csr_matrix(L); Ur = sp.csr_matrix(U); Lr.sort_indices(); Ur.sort_indices()
    out2 = (C.c_int64 * 8)()
    assert L_.nep_lu_analyze(n, 0, hp(Lr.indptr.astype(np.int32)), hp(Lr.indices.astype(np.int32)), hp(Ur.indptr.astype(np.int32)),
                             hp(Ur.indices.astype(np.int32)), out2) == 0
    assert list(out2) == list(out)
    # dependencies respect the partition
    Lc = sp.coo_matrix(sp.tril(L, -1)); Uc = sp.coo_matrix(sp.triu(U, 1))
    same = bid[Lc.row] == bid[Lc.col]
    assert np.all(same | (lvl[Lc.col] < lvl[Lc.row]))
    same = bid[Uc.row] == bid[Uc.col]
    assert np.all(same | (lvl[Uc.col] > lvl[Uc.row]))
    assert out[3] == int(np.sum(bid[Lc.row] != bid[Lc.col])) and out[5] == int(np.sum(bid[Uc.row] != bid[Uc.col]))
    # block solve with explicitly inverted diagonal blocks (what the device kernels do), level by level
    order = np.lexsort((np.arange(n), bid, lvl))
    Ld = L.toarray()[np.ix_(order, order)]; Ud = U.toarray()[np.ix_(order, order)]
    assert np.allclose(np.triu(Ld, 1), 0) and np.allclose(np.tril(Ud, -1), 0)
    bnew = bid[order]
    starts = np.flatnonzero(np.r_[True, bnew[1:] != bnew[:-1]]); ends = np.r_[starts[1:], n]
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    bw = np.empty(n, complex); bw[F["perm_r"]] = b
    bn = bw[order]; y = np.zeros(n, complex); x = np.zeros(n, complex)
    for s_, e_ in zip(starts, ends):                       # blocks come level by level in the new order
        y[s_:e_] = np.linalg.inv(Ld[s_:e_, s_:e_]) @ (bn[s_:e_] - Ld[s_:e_, :s_] @ y[:s_])
    for s_, e_ in zip(starts[::-1], ends[::-1]):
        x[s_:e_] = np.linalg.inv(Ud[s_:e_, s_:e_]) @ (y[s_:e_] - Ud[s_:e_, e_:] @ x[e_:])
    xw = np.empty(n, complex); xw[order] = x
    xo = xw[F["perm_c"]]
    assert np.linalg.norm(A @ xo - b) <= 1e-9 * np.linalg.norm(b)
    # a pattern that is not triangular is rejected
    bad = F["Li"].copy(); bad[:] = bad[::-1]
    assert L_.nep_lu_analyze(n, 1, hp(F["Lp"]), hp(bad), hp(F["Up"]), hp(F["Ui"]), out) in (-2, -5)


def test_c_abi_argument_errors_without_gpu():
    """every entry point validates its arguments before touching the device: status NEP_ERR_ARG (-2) + message"""
    import ctypes as C
    L = na._lib.lib
    assert L.nep_mlincomb(None, 1, None, None, 1, None, None) == -2
    assert b"invalid argument" in L.nep_last_error()
    assert L.nep_orth(None, 1, 1, 1, None, None, None, None, 0, None, None) == -2
    assert L.nep_gemm_ts(None, 1, 1, 1, None, 1, 1, None, 1, 0, None) == -2
    assert L.nep_lu_solve(None, 1, None, 1, None, 1, 1.0, None) == -2
    h = C.c_void_p()
    assert L.nep_spmf_create(0, 1, None, None, None, None, C.byref(h)) == -2 and not h.value
    assert L.nep_lu_create(0, None, None, None, None, None, None, None, None, C.byref(h)) == -2
    assert L.nep_resid_batch(None, 0, None, None, 0, None, None, None) == -2
    # CSC -> CSR helper is pure host code
    A = sp.random(7, 7, 0.4, random_state=0, format="csc")
    rp = np.zeros(8, dtype=np.int32); ci = np.zeros(A.nnz, dtype=np.int32); vv = np.zeros(A.nnz)
    cp = A.indptr.astype(np.int64) + 1; rv = A.indices.astype(np.int64) + 1        # Julia's 1-based Int64 CSC
    st = L.nep_csc_to_csr(7, cp.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p), A.data.ctypes.data_as(C.c_void_p), 0, 1,
                          rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p))
    assert st == 0
    R = sp.csr_matrix((vv, ci, rp), shape=(7, 7))
    assert (R != A).nnz == 0


def test_low_rank_types_host_side():
    """LowRankMatrixAndFunction / LowRankFactorizedNEP (rk_nep.jl:41-67): A = L U^H on the reference's gun W1, W2
    (ranks 19 and 65, test/rk_helper/gun_test_utils.jl) and the PEP + LowRankFactorizedNEP composition of
    test/nleigs/nleigs_nep_types.jl:40"""
    import scipy.sparse as sp
    import nep_amd as na
    d = np.load(os.path.join(ROOT, "nonlineareigenproblems.jl_amd", "data", "gun_W.npz"))
    ranks = []
    for nm in ("W1", "W2"):
        W = sp.csc_matrix((d[nm + "_data"], d[nm + "_indices"], d[nm + "_indptr"]), shape=tuple(d[nm + "_shape"]))
        c = na.LowRankMatrixAndFunction(W, na.funcs.ident())
        assert abs(c.L @ c.U.conj().T - W).max() < 1e-13 and c.L.shape == (W.shape[0], c.U.shape[1])
        ranks.append(c.U.shape[1])
    assert ranks == [19, 65]
    B = [np.array([[1.0, 3], [5, 6]]), np.array([[3.0, 4], [6, 6]])]
    lr = na.LowRankFactorizedNEP([na.LowRankMatrixAndFunction(sp.csc_matrix(np.eye(2)), na.funcs.Monomial(2))])
    nep = na.SumNEP(na.PEP(B), lr)
    assert lr.rank == 2 and len(nep.get_Av()) == 3 and na.rk_helper.rk_structure(nep) == (1, 1)


def test_inpolygon_reference_kat_and_vectorised_form():
    """test/rk_helper/inpolygon.jl:6-31: 96 of the 13 x 13 lattice points lie in or on the M-shaped polygon, either
    orientation, non-finite points are outside; and the all-edges-at-once form of the package equals the scalar
    restatement of the oracle on vertices, edge points, horizontal edges and random points of several polygons"""
    from oracle import nleigs as onl
    inp = na.rk_helper.inpolygon
    px = [0, 0, 5, 10, 10]; py = [0, 10, 5, 10, 0]
    pts = [(x, y) for x in range(-1, 12) for y in range(-1, 12)]
    for f in (inp, onl.inpolygon):
        assert sum(bool(f(x, y, px, py)) for x, y in pts) == 96
        assert sum(bool(f(x, y, px[::-1], py[::-1])) for x, y in pts) == 96
        assert not any(f(x, y, px, py) for x, y in ((np.nan, 0.0), (0.0, np.nan), (np.inf, 0.0), (0.0, np.inf)))
    rng = np.random.default_rng(0)
    th = np.linspace(0, np.pi, 200)
    a = np.sort(rng.uniform(0, 2 * np.pi, 37)); rr = rng.uniform(.5, 1.5, 37)
    polys = [(np.array([-1, 1, 1, -1.]), np.array([-1, -1, 1, 1.])), (np.r_[np.cos(th), -1.0], np.r_[np.sin(th), 0.0]),
             (rr * np.cos(a), rr * np.sin(a)), (np.array([0, 2, 2, 1, 1, 0.]), np.array([0, 0, 2, 2, 1, 1.]))]
    for qx, qy in polys:
        m = len(qx)
        P = [(x, y) for x in np.linspace(-1.6, 2.1, 24) for y in np.linspace(-1.6, 2.1, 24)]
        P += list(zip(qx, qy)) + [((qx[i] + qx[(i + 1) % m]) / 2, (qy[i] + qy[(i + 1) % m]) / 2) for i in range(m)]
        P += [(0.5, 0.0), (1.0, 1.5), (1.5, 1.0), (0.0, 0.0)]
        assert all(bool(inp(x, y, qx, qy)) == bool(onl.inpolygon(x, y, qx, qy)) for x, y in P)


def test_compute_Mder_union_pattern_equals_term_by_term_sum():
    """compute_Mder (src/NEPTypes.jl:343-394) assembles M^(i)(lam) on a cached union sparsity pattern: same matrix as the
    term-by-term sparse sum for gun (disjoint and overlapping patterns), qdep0, a random SPMF with unsorted / duplicate
    entries, the waveguide problem (whose dense corner is added on top), and dense NEPs fall back to the plain sum"""
    rng = np.random.default_rng(0)
    def naive(nep, lam, i):
        Z = None
        for A, f in zip(nep.get_Av(), nep.get_fv()):
            T = A * f.derivs(lam, i + 1)[i]
            Z = T if Z is None else Z + T
        return Z
    rows = rng.integers(0, 30, 200); cols = rng.integers(0, 30, 200)
    A0 = sp.coo_matrix((rng.standard_normal(200), (rows, cols)), shape=(30, 30))            # duplicates, unsorted
    A1 = sp.random(30, 30, 0.1, random_state=1, format="csr") + 1j * sp.random(30, 30, 0.1, random_state=2, format="csr")
    cases = [(na.nep_gallery("gun_spmf", 655), 62500.0 + 300j), (na.nep_gallery("nlevp_native_gun", 655), 5e4 - 20j),
             (na.nep_gallery("qdep0"), 0.3 + 0.1j),
             (na.SPMF_NEP([A0, A1, sp.identity(30, format="csc")], [na.funcs.one(), na.funcs.Exp(-0.5), na.funcs.Monomial(2)]), -0.7 + 0.2j)]
    for nep, lam in cases:
        for i in (0, 1, 2):
            Z = nep.compute_Mder(lam, i); R = naive(nep, lam, i)
            assert sp.issparse(Z) and abs(Z - R).max() <= 1e-15 * max(abs(R).max(), 1e-300)
    d = na.nep_gallery("dep0")
    assert isinstance(d.compute_Mder(0.2), np.ndarray) and np.allclose(d.compute_Mder(0.2), naive(d, 0.2, 0))
    w = na.nep_gallery("WEP", nx=11, nz=7)
    from oracle import wep as ow
    lam = -1.3 - 0.31j
    assert abs(w.compute_Mder(lam) - ow.WEP_FD(11, 7, "TAUSCH").compute_Mder(lam)).max() < 1e-12


def test_asan_host_analysis():
    """AddressSanitizer + UBSan build of the library's host side (tests/sanitize/Makefile) running the K5 symbolic analysis
    on random patterns from 4 threads: no report, no leak (sanitizer target)"""
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("make") is None:
        pytest.skip("needs hipcc + make")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["make", "-C", os.path.join(root, "tests", "sanitize"), "run"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "asan_driver ok" in p.stdout


def test_compute_Mder_union_pattern_with_stored_zeros_and_many_terms():
    """the union pattern of compute_Mder is built from entry keys: explicitly stored zeros keep their slot (a sparse add of
    pattern matrices dropped them -> IndexError when the entry had the largest key), and 300 overlapping terms do not wrap"""
    n = 40
    rng = np.random.default_rng(8)
    A0 = sp.random(n, n, 0.1, random_state=rng, format="csc") + sp.identity(n, format="csc")
    A1 = sp.csc_matrix(([0.0, 2.0], ([n - 1, 3], [n - 1, 5])), shape=(n, n))     # stored zero at the largest key, no eliminate_zeros
    A1 = sp.csc_matrix((np.array([2.0, 0.0]), np.array([3, n - 2]), np.r_[np.zeros(6), np.ones(n - 6), 2].astype(int)), shape=(n, n))
    assert A1.nnz == 2 and 0.0 in A1.data
    nep = na.SPMF_NEP([A0, A1], [na.funcs.one(), na.funcs.ident()])
    lam = 0.3 + 0.1j
    M = nep.compute_Mder(lam)
    ref = A0 + lam * A1
    assert abs(M - ref).max() < 1e-15
    many = [sp.identity(n, format="csc") * (i + 1.0) for i in range(300)]
    nep2 = na.SPMF_NEP(many, [na.funcs.one()] * 300)
    assert abs(nep2.compute_Mder(0.0) - sp.identity(n) * sum(range(1, 301))).max() < 1e-9


def test_recorded_refinement_rule_replay():
    """FactorizeLinSolver.review_recorded replays UMFPACK's stopping rule (the checked loop of solve_dev) on the omegas a
    native iar step records: accepted when the iterate kept is the one the rule returns (or at least as good), a miss when
    the rule would have continued or taken a worsening sweep back"""
    from nep_amd.linsolvers import FactorizeLinSolver
    eps = np.finfo(float).eps
    s = FactorizeLinSolver.__new__(FactorizeLinSolver)
    s.umfpack_refinements = 10; s._recorded_plan = None; s.last_omega = None
    assert s.blind_plan_recorded() == 2                       # before any record: two sweeps
    assert s.review_recorded(np.array([1e-13, 1e-16, 5e-17, 0.0]), 2)     # rule stops after 1 sweep, x_2 is as good
    assert s._recorded_plan == 1 and s.blind_plan_recorded() == 1
    assert s.review_recorded(np.array([1e-13, 1e-16, 0.0, 0.0]), 1)       # exactly what the rule does
    assert s.review_recorded(np.array([1e-17, 0.0, 0.0, 0.0]), 0) and s._recorded_plan == 1   # never below one sweep
    assert not s.review_recorded(np.array([1e-13, 0.0, 0.0, 0.0]), 0)     # omega_0 > 2 eps and no sweep taken: miss
    assert not s.review_recorded(np.array([1e-9, 1e-12, 0.0, 0.0]), 1)    # still halving after the sweeps taken: miss
    assert s.review_recorded(np.array([9.4e-11, 4.4885e-16, 0.0, 0.0]), 1)  # ... unless the kept iterate sits at the noise level of omega
    assert s.review_recorded(np.array([1e-13, 8e-14, 0.0, 0.0]), 1)       # stagnation (> half): the rule stops there too
    assert not s.review_recorded(np.array([1e-13, 5e-13, 0.0, 0.0]), 1)   # the sweep made it worse: rule takes it back
    assert s.review_recorded(np.array([3e-16, 3.5e-16, 0.0, 0.0]), 1) == (3.5e-16 <= 4 * eps)   # ... unless at noise level
    assert not s.review_recorded(np.array([1e-13, np.nan, 0.0, 0.0]), 1)
    s.umfpack_refinements = 0
    assert s.blind_plan_recorded() == 0
    # the settled count travels with the NEP object: the next solver of the same NEP starts with it, a miss withdraws it
    class _Nep: pass
    nep = _Nep()
    a = FactorizeLinSolver.__new__(FactorizeLinSolver); a.umfpack_refinements = 10; a._recorded_plan = None; a.last_omega = None; a.nep = nep
    assert a.blind_plan_recorded() == 2
    assert a.review_recorded(np.array([1e-13, 1e-16, 5e-17, 0.0]), 2) and nep._refine_hint == 1
    b = FactorizeLinSolver.__new__(FactorizeLinSolver); b.umfpack_refinements = 10; b._recorded_plan = None; b.last_omega = None; b.nep = nep
    assert b.blind_plan_recorded() == 1
    assert not b.review_recorded(np.array([1e-9, 1e-12, 0.0, 0.0]), 1) and nep._refine_hint is None
    c = FactorizeLinSolver.__new__(FactorizeLinSolver); c.umfpack_refinements = 10; c._recorded_plan = None; c.last_omega = None; c.nep = nep
    assert c.blind_plan_recorded() == 2
    # ... for good: a later clean record does not bring the hint back on this NEP object
    assert c.review_recorded(np.array([1e-13, 1e-16, 5e-17, 0.0]), 2) and nep._refine_hint is None
    d = FactorizeLinSolver.__new__(FactorizeLinSolver); d.umfpack_refinements = 10; d._recorded_plan = None; d.last_omega = None; d.nep = nep
    assert d.blind_plan_recorded() == 2
    # the hint belongs to ONE shift: a solver of the same NEP at another sigma neither starts with the count nor counts as settled
    nep2 = _Nep()
    def mk(lam):
        x = FactorizeLinSolver.__new__(FactorizeLinSolver); x.umfpack_refinements = 10; x._recorded_plan = None; x.last_omega = None
        x.nep = nep2; x.lam = lam
        return x
    e = mk(0.0)
    assert not e.settled_plan()
    assert e.review_recorded(np.array([1e-13, 1e-16, 5e-17, 0.0]), 2) and nep2._refine_hint == 1 and nep2._refine_hint_lam == 0j
    assert mk(0.0).blind_plan_recorded() == 1 and mk(0.0).settled_plan()
    assert mk(0.3 + 0.1j).blind_plan_recorded() == 2 and not mk(0.3 + 0.1j).settled_plan()
    # a step whose kept iterate was NOT recorded (7 of 8 once settled): the rule is replayed on x_0 .. x_{plan-1}
    f = mk(0.0)
    assert f.review_recorded(np.array([1e-13, 0.0, 0.0, 0.0]), 1, final_recorded=False)            # still improving when the sweep was taken
    assert f.review_recorded(np.array([1e-17, 0.0, 0.0, 0.0]), 1, final_recorded=False)            # x_0 converged already: accepted, not a miss
    assert f._recorded_plan == 1 and nep2._refine_hint == 1
    assert f.review_recorded(np.array([1e-13, 1e-17, 0.0, 0.0]), 2, final_recorded=False) and f._recorded_plan == 1
    assert not f.review_recorded(np.array([1e-13, 9e-14, 0.0, 0.0]), 2, final_recorded=False)      # stagnation before the last sweep: miss
    assert nep2._refine_hint is None


def test_hosteig_hessenberg_route():
    """_hosteig.eig(H, hessenberg=True): eigenvalues without Schur vectors + inverse iteration on the Hessenberg matrix give
    the same decomposition as zgeev (unit 2-norm columns, residual at round-off level); small n and failures fall back"""
    import scipy.linalg as sl
    from nep_amd import _hosteig
    rng = np.random.default_rng(5)
    A = rng.standard_normal((160, 160)) + 1j * rng.standard_normal((160, 160))
    for n in (6, 60, 100):
        H = np.triu(sl.hessenberg(A)[:n, :n], -1)
        w, V = _hosteig.eig(H, hessenberg=True)
        w0, V0 = _hosteig.eig(H)
        assert np.abs(H @ V - V * w[None, :]).max() <= 1e-10 * np.linalg.norm(H, 2)
        assert np.allclose(np.linalg.norm(V, axis=0), 1.0, atol=1e-12)
        order = [int(np.argmin(abs(w - x))) for x in w0]
        assert sorted(order) == list(range(n)) and np.abs(w[order] - w0).max() <= 1e-10 * np.abs(w0).max()
        for j, i in enumerate(order):                      # same eigenvector up to a phase
            assert 1 - abs(np.vdot(V0[:, j], V[:, i])) <= 1e-8


def test_device_lu_plan_is_independent_of_the_enumeration_threads():
    """nep_lu_refac_analyze (host-only dry run of the plan builder of csrc/lufac.hip: symbolic partition, two-pass product
    enumeration on worker threads, classification, placement): the product count equals sum_k |L(:,k)| |U(k,:)| computed
    independently, the classes add up, and the hash of the plan arrays is the same for 1, 2, 5 and 8 enumeration threads"""
    import ctypes as C
    import nep_amd_hostlu as hl
    from nep_amd._lib import lib, hptr
    nep = na.nep_gallery("gun_spmf_scaled", 1310)
    A = sp.csc_matrix(nep.compute_Mder(0.0)).astype(np.complex128); A.sort_indices()
    F = hl.factor(A.data, A.indices, A.indptr, A.shape)
    assert F["strategy"].get("symmetric_mode") and np.array_equal(F["perm_r"], F["perm_c"])
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    arrs = [i32(F[k]) for k in ("Lp", "Li", "Up", "Ui", "perm_r", "perm_c")] + [i32(A.indptr), i32(A.indices)]
    n = A.shape[0]
    expect = int(np.sum((np.diff(arrs[0]) - 1).astype(np.int64) * (np.bincount(arrs[3], minlength=n) - 1)))
    res = []
    old = os.environ.get("NEP_LU_PLAN_THREADS")
    try:
        for thr in (1, 2, 5, 8):
            os.environ["NEP_LU_PLAN_THREADS"] = str(thr)
            out = (C.c_int64 * 8)()
            assert lib.nep_lu_refac_analyze(n, *[hptr(a) for a in arrs], out) == 0
            res.append(list(out))
    finally:
        if old is None:
            os.environ.pop("NEP_LU_PLAN_THREADS", None)
        else:
            os.environ["NEP_LU_PLAN_THREADS"] = old
    assert all(r == res[0] for r in res)
    prod, internal, external, segs, wide, steps, levels, _ = res[0]
    assert prod == expect == internal + external + wide and segs >= 1 and levels >= 2 and steps >= 1
    # malformed input is rejected, not read out of bounds
    bad = [a.copy() for a in arrs]; bad[1][0] = n + 5
    assert lib.nep_lu_refac_analyze(n, *[hptr(a) for a in bad], (C.c_int64 * 8)()) != 0


def test_gun_loader_from_directory(tmp_path, monkeypatch):
    """the NEPMI_GUN_DIR path of gallery.gun_matrices (src/gallery_extra/NLEVP_native.jl:4-18 reads gun_K.txt / gun_M.txt in
    the text format of src/utils/Serialization.jl:8-31): files written in that format are read back bit for bit, pass the
    1-norm known-answer check of test/rk_helper/gun_test_utils.jl:50-51, and a wrong file is refused instead of silently
    defining another problem.  (The physical matrices are absent from the reference checkout; the files here hold the
    stand-in, which carries the reference's norms by construction.)"""
    import scipy.sparse as sp
    from nep_amd import gallery
    K, M = gallery.gun_standin_KM()
    gallery.write_sparse_matrix(str(tmp_path / "gun_K.txt"), K)
    gallery.write_sparse_matrix(str(tmp_path / "gun_M.txt"), M)
    monkeypatch.setenv("NEPMI_GUN_DIR", str(tmp_path))
    K2, M2, W1, W2 = gallery.gun_matrices()
    assert (abs(K2 - K)).max() == 0.0 and (abs(M2 - M)).max() == 0.0
    assert abs(gallery._onenorm(K2) - 1.474544889815002e+05) <= 1e-12 * 1.474544889815002e+05
    assert abs(gallery._onenorm(M2) - 2.726114618171165e-02) <= 1e-12 * 2.726114618171165e-02
    assert abs(gallery._onenorm(W1) - 2.328612251920476) < 1e-14 and abs(gallery._onenorm(W2) - 3.793375498194695) < 1e-14
    nep = gallery.nlevp_native_gun()
    assert nep.n == 9956
    gallery.write_sparse_matrix(str(tmp_path / "gun_M.txt"), sp.csc_matrix(M * 1.001))
    with pytest.raises(ValueError, match="1-norm"):
        gallery.gun_matrices()
    gallery.write_sparse_matrix(str(tmp_path / "gun_M.txt"), sp.csc_matrix(M[:100, :100]))
    with pytest.raises(ValueError, match="shape"):
        gallery.gun_matrices()


@pytest.mark.parametrize("case", ["gun", "wep", "qdep0", "random", "dense_row"])
def test_k1_footprint_tiles_host_dryrun(case, monkeypatch):
    """csrc/spmv_tile.hip, host side (nep_spmf_tiles_analyze, no GPU): every row belongs to exactly one block, owned rows are
    flagged in their block's footprint, and z = sum_t A_t (V c_t) walked through the tiles (footprint -> W -> 16-bit entries
    -> row map) equals the direct evaluation; grid stride detection on the gun / waveguide stencils; blocks whose footprint
    exceeds the LDS budget are split; a matrix with a row that can never fit gets no tiles"""
    import scipy.sparse as sp
    from nep_amd.nep import tiles_analyze
    from nep_amd import gallery, wep
    if case == "gun":
        K, M, W1, W2 = gallery.gun_matrices()
        d = tiles_analyze([K, -M, W1, W2], k=5)
        assert d["stride"] == 131 and d["blocks"] >= 128 and d["max_footprint"] <= 128
        assert d["stream_bytes"] < 12 * (K.nnz + M.nnz + W1.nnz + W2.nnz) + 4 * 4 * 9957      # below the stacked CSR's bytes
    elif case == "wep":
        Av = wep.WaveguideData(109, 105, "JARLEBRING").big_matrices()
        d = tiles_analyze(Av, k=3)
        assert d["stride"] == 105 and d["blocks"] > 0
        monkeypatch.setenv("NEP_K1_TILE_XP", "8"); monkeypatch.setenv("NEP_K1_TILE_ZP", "64")
        d2 = tiles_analyze(Av, k=3)
        assert d2["max_rel_err"] <= 1e-14 and d2["blocks"] < d["blocks"]
        monkeypatch.setenv("NEP_K1_TILE_LDS_KB", "8")               # 8 KiB / (16 B * 3 terms) = 170 columns: patches must split
        d3 = tiles_analyze(Av, k=3)
        assert d3["max_footprint"] <= 170 and d3["blocks"] > d2["blocks"] and d3["max_rel_err"] <= 1e-14
    elif case == "qdep0":
        d = tiles_analyze(na.nep_gallery("qdep0").get_Av(), k=2)
        assert d["stride"] == 0 and d["blocks"] > 0
    elif case == "random":
        A = sp.random(4000, 4000, density=0.002, random_state=1, format="csr") + sp.identity(4000, format="csr")
        B = sp.random(4000, 4000, density=0.001, random_state=2, format="csr") * (1 + 2j)
        d = tiles_analyze([A, B], k=4)
        assert d["blocks"] > 0
    else:
        A = sp.lil_matrix((6000, 6000)); A.setdiag(1.0); A[17, :] = 1.0      # one dense row: 6000 columns > any LDS budget
        d = tiles_analyze([sp.csr_matrix(A)], k=2)
        assert d["blocks"] == 0
        return
    assert d["max_rel_err"] <= 1e-14


def test_julia_binding_symbols_exist():
    """julia/NEPMI355X.jl (the reference-side binding of INTEGRATION.md, shipped as a file; Julia is not installed here, so it
    cannot be executed): every `ccall` target is declared in include/nepmi355.h and exported by the library, and the file is
    the code block of INTEGRATION.md verbatim (one source of truth)"""
    jl = open(os.path.join(ROOT, "julia", "NEPMI355X.jl")).read()
    ex = open(os.path.join(ROOT, "julia", "usage_examples.jl")).read()
    names = set(re.findall(r"\(:(nep_[a-z0-9_]+)\s*,\s*LIB\)", jl + ex))
    assert len(names) >= 25
    hdr = open(os.path.join(ROOT, "include", "nepmi355.h")).read()
    lib = ctypes.CDLL(na.LIB_PATH)
    for nme in sorted(names):
        assert re.search(r"\b%s\s*\(" % nme, hdr), "not in the header: " + nme
        assert hasattr(lib, nme), "not exported: " + nme
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    first = md.split("```julia\n", 1)[1].split("\n```", 1)[0]
    assert first.strip() in jl


def _split_top(txt):
    """split at top-level commas (parentheses, brackets and braces nest; string literals are skipped)"""
    out, depth, cur, i = [], 0, "", 0
    while i < len(txt):
        ch = txt[i]
        if ch == '"':
            j = txt.index('"', i + 1); cur += txt[i:j + 1]; i = j + 1; continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(txt, start):
    """text between the parenthesis at `start` and its partner"""
    depth, i = 0, start
    while True:
        if txt[i] == '"':
            i = txt.index('"', i + 1)
        elif txt[i] == "(":
            depth += 1
        elif txt[i] == ")":
            depth -= 1
            if depth == 0:
                return txt[start + 1:i]
        i += 1


def _julia_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Csize_t": "i64", "Float64": "f64", "ComplexF64": "c128"}[t]


def _c_class(t):
    t = t.strip()
    if "*" in t or "[" in t or re.match(r"(const\s+)?(nep_stream|nep_fv_eval)\b", t):   # (arrays decay; nep_fv_eval: a function-pointer typedef)
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).split()[0]
    return {"int32_t": "i32", "int": "i32", "int64_t": "i64", "size_t": "i64", "double": "f64", "nep_cdouble": "c128"}[t]


def _c_prototypes(hdr):
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(nep_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [] if args.strip() in ("", "void") else [a for a in _split_top(args)]
        protos[name] = (_c_class(ret + " x"), [_c_class(a) for a in args])
    return protos


def _julia_ccalls(src):
    src = "\n".join(ln.split("#")[0] if '"' not in ln else ln for ln in src.splitlines())      # comments off (no '#' in strings here)
    calls = []
    for m in re.finditer(r"\bccall\(", src):
        parts = _split_top(_balanced(src, m.end() - 1))
        name = re.match(r"\(:(nep_[a-z0-9_]+)\s*,\s*LIB\)", parts[0]).group(1)
        tup = parts[2]
        assert tup.startswith("(") and tup.endswith(")"), (name, tup)
        types = [x for x in _split_top(tup[1:-1]) if x]
        calls.append((name, _julia_class(parts[1]), [_julia_class(x) for x in types], len(parts) - 3))
    return calls


def test_julia_ccall_signatures_match_the_header():
    """every `ccall` of julia/NEPMI355X.jl and julia/usage_examples.jl against its prototype in include/nepmi355.h: same
    number of arguments in the type tuple AND in the call, same scalar width / pointer-ness per position, same return kind
    (Julia cannot be run here, and a wrong tuple is silent memory corruption there).  The checker itself is checked on a
    deliberately broken tuple."""
    protos = _c_prototypes(open(os.path.join(ROOT, "include", "nepmi355.h")).read())
    assert len(protos) >= 90 and protos["nep_axpy"] == ("i32", ["i64", "c128", "ptr", "ptr", "ptr"])
    n = 0
    for fn in ("NEPMI355X.jl", "usage_examples.jl"):
        for name, ret, types, nargs in _julia_ccalls(open(os.path.join(ROOT, "julia", fn)).read()):
            cret, cargs = protos[name]
            assert ret == cret, (fn, name, "return", ret, cret)
            assert types == cargs, (fn, name, types, cargs)
            assert nargs == len(types), (fn, name, "call passes %d values for %d types" % (nargs, len(types)))
            n += 1
    assert n >= 40
    # also the ccalls quoted in INTEGRATION.md outside the file's own block
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for block in md.split("```julia\n")[2:]:
        for name, ret, types, nargs in _julia_ccalls(block.split("\n```", 1)[0]):
            assert (ret, types) == protos[name], ("INTEGRATION.md", name, types, protos[name][1])
            assert nargs == len(types), ("INTEGRATION.md", name)
    # the checker fails on a broken tuple: Int32 where the header has int64_t, and a missing argument
    bad = 'chk(ccall((:nep_axpy, LIB), Cint, (Int32, ComplexF64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), n, a, x, y, C_NULL))'
    (name, ret, types, nargs), = _julia_ccalls(bad)
    assert types != protos[name][1]
    bad2 = 'chk(ccall((:nep_axpy, LIB), Cint, (Int64, ComplexF64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), n, a, x, y))'
    (name, ret, types, nargs), = _julia_ccalls(bad2)
    assert types == protos[name][1] and nargs != len(types)


def test_no_vendor_blas_or_fft_behind_the_abi():
    """the product's library carries no reference to rocBLAS / hipBLAS / rocFFT (until round 3 nep_zgemm / nep_dgemm dlopen'ed
    rocBLAS for the dense-transform fallback of the waveguide preconditioner): every GEMM / DFT behind the C ABI is this
    library's own kernel.  (RCCL is the one vendor library it loads: the collective of the sharded contour integrators.)"""
    blob = open(na.LIB_PATH, "rb").read().lower()
    for name in (b"rocblas", b"hipblas", b"rocfft", b"hipfft", b"rocsparse", b"rocsolver", b"librocprim", b"libhipcub", b"rocprim", b"hipcub"):
        assert name not in blob, name           # (header-only rocPRIM / hipCUB kernels would show up by their mangled names)
    assert b"rccl" in blob
    # no source includes a vendor primitive header: until round 5 csrc/lufac.hip compiled hipCUB's scan and radix sort in for the
    # ONE-OFF enumeration of a device-LU plan (nep_lu_refac_create); round 6 replaced them by csrc/devprims.h
    csrc = os.path.join(ROOT, "nonlineareigenproblems.jl_amd", "csrc")
    users = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h"))
                   and re.search(r"#include\s*<(hipcub|rocprim|rocblas|hipblas|rocfft|hipfft|rocsparse|rocsolver|rocwmma|ck|thrust)[/_.]", open(os.path.join(csrc, f)).read()))
    assert users == [], users


def test_discretizepolygon_native_walk_equals_the_interpreted_one(monkeypatch):
    """nep_amd.rk_helper.discretizepolygon hands the boundary walk of src/rk_helper/discretizepolygon.jl to the library's host
    code (nep_discretize_polygon: every operation separately rounded, no fused multiply-adds) where oracle/nleigs.py walks it in
    the interpreter: the SAME bits for every point (the Leja-Bagby selection behind it takes an argmax over these candidates) --
    polygons of 3-11 vertices over nine decades of size, a 1500-vertex arc like gun's target set, near-degenerate edges, point
    counts that end mid-edge; NEP_RK_NATIVE=0 is the interpreted walk"""
    from nep_amd import rk_helper as rk
    from oracle import nleigs as on
    rng = np.random.default_rng(0)
    cases = [np.array([-1 - 1j, -1 + 1j, 1 + 1j, 1 - 1j]),
             146.71 ** 2 + (300.0 ** 2 - 146.71 ** 2) / 2 * (1 + np.exp(1j * np.linspace(0, np.pi, 9))),
             62500.0 + 28000.0 * np.exp(1j * np.linspace(0, np.pi, 1500))]
    for _ in range(60):
        k = int(rng.integers(3, 12))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k)); r = rng.uniform(0.1, 5, k) * 10.0 ** int(rng.integers(-3, 6))
        cases.append(r * np.exp(1j * ang) + 3 * rng.standard_normal())
    for _ in range(10):
        z = rng.standard_normal(5) + 1j * rng.standard_normal(5); z[2] = z[1] + 1e-13
        cases.append(z)
    calls = {"n": 0}
    orig = rk._discretize_native

    def counting(*a):
        r = orig(*a)
        calls["n"] += r is not None
        return r
    monkeypatch.setattr(rk, "_discretize_native", counting)
    for z in cases:
        for npts in (10000, 1000, 37, 3):
            a, _ = rk.discretizepolygon(z, False, npts); b, _ = on.discretizepolygon(z, False, npts)
            assert a.shape == b.shape and np.array_equal(a.view(np.float64), b.view(np.float64))
    assert calls["n"] == 4 * len(cases)                      # the library did the walks
    a, Za = rk.discretizepolygon(cases[0], True); b, Zb = on.discretizepolygon(cases[0], True)
    assert np.array_equal(a, b) and np.array_equal(Za, Zb)
    monkeypatch.setenv("NEP_RK_NATIVE", "0")
    n0 = calls["n"]
    a, _ = rk.discretizepolygon(cases[2]); b, _ = on.discretizepolygon(cases[2])
    assert calls["n"] == n0 and np.array_equal(a.view(np.float64), b.view(np.float64))


def test_lejabagby_in_place_form_equals_the_oracle_bit_for_bit():
    """nep_amd.rk_helper.lejabagby issues the reference's update  s * betainv * (X - a_j) / (1 - X * binv)  (src/rk_helper/lejabagby.jl)
    as the same ufunc calls into work arrays: nodes, poles and scaling factors equal oracle/nleigs.py's to the last bit -- finite and
    infinite poles, forced infinities, kept nodes, A and C the same array or not"""
    from nep_amd import rk_helper as rk
    from oracle import nleigs as on
    rng = np.random.default_rng(1)
    for t in range(12):
        nA = int(rng.integers(5, 1500)); nC = int(rng.integers(5, 1500)); m = int(rng.integers(2, 30))
        A = (rng.standard_normal(nA) + 1j * rng.standard_normal(nA)) * 10.0 ** int(rng.integers(-2, 5))
        C = A if t % 3 == 0 else rng.standard_normal(nC) + 1j * rng.standard_normal(nC)
        for B in ([np.inf], list(rng.standard_normal(7) * 50), [np.inf, 3.0, -2.5]):
            for keepA in (False, True):
                for force in (0, 3):
                    if keepA and len(A) < m:
                        continue
                    r1 = rk.lejabagby(A, B, C, m, keepA, force); r2 = on.lejabagby(A, B, C, m, keepA, force)
                    for x, y in zip(r1, r2):
                        assert np.array_equal(np.asarray(x).view(np.float64), np.asarray(y).view(np.float64), equal_nan=True)


def test_bench_gpus_flag_and_launcher_must_agree():
    """bench.py: `--gpus N` is what decides the rank count.  Under a launcher (WORLD_SIZE set) a different N is an error, not a
    silently ignored flag; without a launcher and N > 1 the file starts its own ranks (the GPU suite runs that for real,
    tests/test_gpu_dist2.py::test_bench_gpus_flag_launches_its_own_ranks) -- here: on a box without GPUs that is a loud exit too."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "disagree" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NEP_BENCH_SHARE_GPU")}
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "GPU(s)" in r.stderr
    src = open(bench).read()
    assert "args.gpus" in src and "torch.distributed.run" in src


# ---- a small model of Julia's Array / SubArray addressing, to evaluate the pointer expressions of julia/NEPMI355X.jl -------------
class _JArray:
    """column-major Array{ComplexF64,N} at byte address `base`"""
    elsize = 16

    def __init__(self, dims, base):
        self.dims = tuple(dims); self.base = base
        self.strides = tuple(int(np.prod(self.dims[:d])) for d in range(len(self.dims)))
        self.first = 1                        # linear index (1-based) of the first element in the parent = itself

    def parent(self):
        return self

    def addr_linear(self, i):                 # pointer(A, i)
        return self.base + self.elsize * (i - 1)


class _JSub:
    """SubArray of a _JArray, indices = ints or (start, step, length) ranges (1-based): what `view(A, ...)` builds.  `first` is
    Base.first_index(V) (parent-space linear index of V[1,1,...]), `strides` are in parent ELEMENTS."""
    elsize = 16

    def __init__(self, par, idx):
        assert len(idx) == len(par.dims)
        self.par = par; dims = []; strides = []; first = 1
        for d, ix in enumerate(idx):
            if isinstance(ix, int):
                first += (ix - 1) * par.strides[d]
            else:
                st, step, ln = ix
                assert 1 <= st and st + (ln - 1) * step <= par.dims[d]
                first += (st - 1) * par.strides[d]
                dims.append(ln); strides.append(step * par.strides[d])
        self.dims = tuple(dims); self.strides = tuple(strides); self.first = first

    def parent(self):
        return self.par

    def addr_first(self):                     # unsafe_convert(Ptr{T}, V) = pointer(parent) + _byte_offset(V)
        return self.par.base + self.elsize * (self.first - 1)

    def addr_linear(self, i):                 # pointer(V, i::Int): i is a linear index IN THE VIEW (Base._memory_offset via _to_subscript_indices)
        sub = []; r = i - 1
        for ln in self.dims[:-1]:
            sub.append(r % ln); r //= ln
        sub.append(r)
        return self.addr_first() + self.elsize * sum(s * st for s, st in zip(sub, self.strides))

    def addr_cart(self, *I):
        return self.addr_first() + self.elsize * sum((i - 1) * st for i, st in zip(I, self.strides))


def _jl_env():
    def pointer(A, i=None):
        if i is None:
            return A.addr_first() if isinstance(A, _JSub) else A.base
        return A.addr_linear(i)

    def size(A, d=None):
        return A.dims if d is None else A.dims[d - 1]
    return {"pointer": pointer, "size": size, "stride": lambda A, d: A.strides[d - 1], "parent": lambda A: A.parent(), "__builtins__": {}}


def _jl_to_py(expr):
    """Julia arithmetic -> Python: juxtaposed numeric literal coefficients (`16ldmax`, `16S.rows`, `16stride(V, 2)`) get their `*`"""
    return re.sub(r"(?<![A-Za-z_0-9.])(\d+)\s*(?=[A-Za-z_(])", r"\1*", expr)


def _jl_eval(expr, **names):
    env = _jl_env(); env.update(names)
    return eval(_jl_to_py(expr), env)


def _julia_call_args(src, symbol, within=None):
    """argument VALUES (texts) of every ccall of `symbol`, optionally only inside the function whose text contains `within`"""
    src = "\n".join(ln.split("#")[0] if '"' not in ln else ln for ln in src.splitlines())
    if within is not None:
        a = src.index(within); src = src[a:src.index("\nend", a)]
    out = []
    for m in re.finditer(r"\bccall\(", src):
        parts = _split_top(_balanced(src, m.end() - 1))
        if re.match(r"\(:%s\s*,\s*LIB\)" % symbol, parts[0]):
            out.append(parts[3:])
    return out


def test_julia_glue_pointer_arithmetic_on_subarray_model():
    """every address julia/NEPMI355X.jl forms by arithmetic, evaluated on the model above for the shapes the reference's drivers pass
    (iar: method_iar.jl:96-107, a NON-contiguous view with rows < leading dimension; tiar: method_tiar.jl:128, contiguous columns;
    Beyn: column blocks of DevBufs) and compared with the byte address the C side expects.  Julia cannot run here; the by-reading
    defect of round 4 (`pointer(V, (j-1)*stride(V, 2) + 1)`: parent-space offset used as a view-space linear index) is the
    known-bad expression this test must reject."""
    from types import SimpleNamespace as NS
    jl = open(os.path.join(ROOT, "julia", "NEPMI355X.jl")).read()
    ups = _julia_call_args(jl, "nep_upload", within="function IterativeSolvers.orthogonalize_and_normalize!")
    assert len(ups) == 2                                       # the column loop and w
    (dst_col, src_col, nbytes_col, _), (dst_w, src_w, nbytes_w, _) = ups
    assert "pointer(V, " not in src_col                        # no linear-index form
    orth = _julia_call_args(jl, "nep_orth", within="function IterativeSolvers.orthogonalize_and_normalize!")[0]
    assert orth[0] == "m.buf.ptr" and orth[1] == "ldmax" and orth[2] == "rows" and orth[3] == "k" and orth[5] == "m.w.ptr"
    bad = "pointer(V, (j-1)*stride(V, 2) + 1)"
    BUF = 0x7000_0000_0000; WB = 0x7100_0000_0000
    seen_bad = False

    def check_orth_call(V, w, par_base, ld_parent, col0, k):
        """V: the view passed as basis (k columns starting at parent column col0), w: the view passed as new vector"""
        nonlocal seen_bad
        rows = V.dims[0]
        names = dict(V=V, w=w, rows=rows, k=k, ldmax=_jl_eval("size(parent(V), 1)", V=V), m=NS(buf=NS(ptr=BUF), w=NS(ptr=WB)))
        assert names["ldmax"] == ld_parent
        assert _jl_eval("stride(V, 1)", V=V) == 1 and _jl_eval("stride(w, 1)", w=w) == 1
        for j in range(1, k + 1):
            want_src = par_base + 16 * ((col0 - 1 + j - 1) * ld_parent)                 # parent element (1, col0 + j - 1)
            assert _jl_eval(src_col, j=j, **names) == want_src == V.addr_cart(1, j)
            assert _jl_eval(dst_col, j=j, **names) == BUF + 16 * ld_parent * (j - 1)    # column j of the mirror, ldv = ldmax as nep_orth is told
            if _jl_eval(bad, j=j, **names) != want_src:
                seen_bad = True
        assert _jl_eval(nbytes_col, **names) == 16 * rows == _jl_eval(nbytes_w, **names)
        assert rows <= ld_parent

    # iar (method_iar.jl:96-97): V = zeros(n(m+1), m+1); VV = view(V, 1:1:n(k+1), 1:k); vv = view(V, 1:1:n(k+1), k+1)
    n, m = 7, 5
    base = 0x1000_0000
    Vp = _JArray((n * (m + 1), m + 1), base)
    for k in range(1, m + 1):
        VV = _JSub(Vp, ((1, 1, n * (k + 1)), (1, 1, k)))
        vv = _JSub(Vp, ((1, 1, n * (k + 1)), k + 1))
        assert VV.dims == (n * (k + 1), k) and vv.dims == (n * (k + 1),)
        check_orth_call(VV, vv, base, n * (m + 1), 1, k)
        assert _jl_eval("pointer(w)", w=vv) == base + 16 * k * n * (m + 1)               # what `w` passed as Ptr{ComplexF64} converts to
    assert seen_bad, "the round-4 expression must be wrong on the iar view for some k < m, j > 1"
    # tiar (method_tiar.jl:128): Z n x (m+1); view(Z, :, 1:k), view(Z, :, k+1): contiguous, leading dimension n
    Zp = _JArray((n, m + 1), base)
    for k in range(1, m + 1):
        check_orth_call(_JSub(Zp, ((1, 1, n), (1, 1, k))), _JSub(Zp, ((1, 1, n), k + 1)), base, n, 1, k)
    # a view that does not start at the parent's first column or row stays right as well (pointer(V) carries the offset)
    Vo = _JSub(Vp, ((3, 1, n), (2, 1, 3)))
    for j in (1, 2, 3):
        assert _jl_eval(src_col, V=Vo, j=j) == base + 16 * ((j) * n * (m + 1) + 2)
    # a plain Matrix (no view): parent(V) === V
    M = _JArray((n, 4), base)
    assert _jl_eval("size(parent(V), 1)", V=M) == n and _jl_eval(src_col, V=M, j=3) == base + 16 * 2 * n
    # Beyn / lin_solve!: column j of the right-hand-side block and of the partial-sum block
    assert "b = dB.ptr + 16n*(j-1)" in jl and "S.ptr + 16S.rows*(j-1)" in jl
    DB = 0x7200_0000_0000
    for j in (1, 2, 32):
        assert _jl_eval("dB.ptr + 16n*(j-1)", dB=NS(ptr=DB), n=9956, j=j) == DB + 16 * 9956 * (j - 1)
        assert _jl_eval("S.ptr + 16S.rows*(j-1)", S=NS(ptr=DB, rows=9956 * 32), j=j) == DB + 16 * 9956 * 32 * (j - 1)
    # the model itself: linear indexing of a non-contiguous view walks the VIEW (Base semantics), cartesian agrees with strides
    Vs = _JSub(Vp, ((1, 1, 2 * n), (1, 1, 3)))
    assert Vs.addr_linear(2 * n + 1) == Vs.addr_cart(1, 2) == base + 16 * n * (m + 1)
    assert Vs.addr_linear(n * (m + 1) + 1) != Vs.addr_cart(1, 2)


def test_pattern_digest_cache_notices_an_in_place_edit():
    """_DeviceRefactor.key remembers the digest of index arrays it has hashed (same memory, same shape) -- and a strided content sample
    with it: a pattern edited IN PLACE (here: two column indices of every row swapped, nnz unchanged) gets a new digest, not the plan
    of the old pattern"""
    from nep_amd.linsolvers import _DeviceRefactor
    rng = np.random.default_rng(5)
    A = (sp.random(400, 400, density=0.02, random_state=3, format="csr") + sp.identity(400)).tocsr()
    A.sort_indices()
    k1 = _DeviceRefactor.key(A, ("x",))
    assert _DeviceRefactor.key(A, ("x",)) == k1                       # cached: same arrays, same contents
    ip, ix = A.indptr, A.indices
    for r in range(400):
        if ip[r + 1] - ip[r] >= 2:
            ix[ip[r]], ix[ip[r] + 1] = ix[ip[r] + 1], ix[ip[r]]      # in place: same address, same length
    k2 = _DeviceRefactor.key(A, ("x",))
    assert k2 != k1
    B = A.copy()
    assert _DeviceRefactor.key(B, ("x",)) == k2                       # content decides, not the address


def _struct_fields_c(hdr, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        for nm in names.split(","):
            out.append((nm.strip(), {"int32_t": "i32", "double": "f64", "nep_cdouble": "c128"}[typ]))
    return out


def test_julia_and_ctypes_structs_match_the_header():
    """nep_iar_opts / nep_iar_result are passed by reference from three places: the header's definition, the Julia `struct`s and
    the ctypes Structures must list the same fields in the same order with the same widths"""
    from nep_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nepmi355.h")).read()
    jl = open(os.path.join(ROOT, "julia", "NEPMI355X.jl")).read()
    jmap = {"Int32": "i32", "Float64": "f64", "ComplexF64": "c128"}
    cmap = {_lib.c_i32: "i32", _lib.c_dbl: "f64", _lib.cdouble: "c128"}
    for cname, jname, ct in (("nep_iar_opts", "IarOpts", _lib.IarOpts), ("nep_iar_result", "IarResult", _lib.IarResult)):
        want = _struct_fields_c(hdr, cname)
        body = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % jname, jl, flags=re.S).group(1)
        body = "\n".join(ln.split("#")[0] for ln in body.splitlines())
        jf = [(a.strip(), jmap[b.strip()]) for a, b in re.findall(r"([A-Za-z_]+)::([A-Za-z0-9]+)", body)]
        assert [t for _, t in jf] == [t for _, t in want], (cname, jf, want)
        assert [n for n, _ in jf] == [n for n, _ in want], (cname, jf, want)
        assert [(n, cmap[t]) for n, t in ct._fields_] == want, cname


def test_julia_wrapper_types_accept_every_reference_spmf_type():
    """Julia's type parameters are invariant: a field `org::AbstractSPMF{T}` with a concrete T cannot hold the reference's
    `PEP <: AbstractSPMF{AbstractMatrix}` or `SPMFSumNEP <: AbstractSPMF{AbstractMatrix}` (round-5 defect: DeviceSPMF{T}).  Checked
    against the reference's own struct declarations when the checkout is present: every AbstractSPMF subtype declared there must
    be a legal value of every field of the glue that is typed AbstractSPMF..., and the glue's own subtypes of a parametric abstract
    type must not carry a parametric field of that abstract type."""
    jl = open(os.path.join(ROOT, "julia", "NEPMI355X.jl")).read()
    jl_nc = "\n".join(ln.split("#")[0] for ln in jl.splitlines())
    fields = re.findall(r"(\w+)::(AbstractSPMF[^\s;,)]*)", jl_nc)
    assert fields, "the glue wraps an AbstractSPMF somewhere"
    for name, typ in fields:
        assert typ == "AbstractSPMF", "field / argument %s::%s is parametric: invariance excludes PEP and SPMFSumNEP" % (name, typ)
    assert re.search(r"mutable struct DeviceSPMF <: AbstractSPMF\{AbstractMatrix\}", jl_nc)
    assert "DeviceSPMF{" not in jl_nc
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present (GPU box)")
    decl = {}
    for fn in os.listdir(ref):
        if fn.endswith(".jl"):
            for m in re.finditer(r"^\s*(?:mutable\s+)?struct\s+(\w+)(\{[^}]*\})?\s*<:\s*AbstractSPMF(\{[^}]*\})?", open(os.path.join(ref, fn)).read(), flags=re.M):
                decl[m.group(1)] = m.group(3)
    # the types the gallery hands out for the BASELINE problems are among them, with BOTH kinds of supertype parameter
    assert decl.get("PEP") == "{AbstractMatrix}" and decl.get("SPMFSumNEP") == "{AbstractMatrix}" and decl.get("SPMF_NEP") == "{T}"
    # the orthogonalisation method is dispatched on an INSTANCE (src/method_iar.jl:50 `orthmethod=DGKS()`, test/iar.jl:13-17)
    assert re.search(r"^struct DeviceDGKS <: IterativeSolvers.OrthogonalizationMethod end", jl, flags=re.M)
    assert "::Type{DeviceDGKS}" not in jl_nc and re.search(r"h::StridedVector\{ComplexF64\}, ::DeviceDGKS\)", jl_nc)
    assert "orthmethod=DGKS()" in open(os.path.join(ref, "method_iar.jl")).read()
    # the probe cache compares contents (Base.hash samples arrays of >= 8192 entries)
    assert "hash(b)" not in jl_nc and "p[3] == b" in jl_nc


def test_julia_iar_tiar_methods_mirror_the_reference_keywords():
    """`iar(::Type{T}, nep::DeviceSPMF; ...)` / `tiar(...)` take exactly the keyword list of the reference's methods (an unchanged
    caller's keywords must all be accepted), and their fallback `invoke`s the reference method with every one of them"""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present (GPU box)")
    jl = open(os.path.join(ROOT, "julia", "NEPMI355X.jl")).read()

    def kwnames(src, head):
        i = src.index(head)
        j = src.index("(", i)
        sig = _balanced(src, j)
        kws = sig.split(";", 1)[1]
        return [re.match(r"\s*([\wσγ]+)", part).group(1) for part in _split_top(kws)]
    for meth, fn in (("iar", "method_iar.jl"), ("tiar", "method_tiar.jl")):
        rsrc = open(os.path.join(ref, fn)).read()
        want = kwnames(rsrc, "function %s(" % meth)
        have = kwnames(jl, "function %s(::Type{T}, nep::DeviceSPMF;" % meth)
        assert have == want, (meth, have, want)
        inv = jl[jl.index("invoke(%s, Tuple{Type{T},NEP}, T, nep;" % meth):]
        inv = _balanced(inv, inv.index("("))
        passed = re.findall(r"([\wσγ]+)\s*=\s*\1\b", inv)
        assert sorted(passed) == sorted(want), (meth, passed, want)


def test_refinement_rule_in_c_equals_the_python_host():
    """UMFPACK's stopping rule replayed on recorded omegas exists twice: linsolvers.py review_recorded (the step-at-a-time host) and
    csrc/iar_run.hip (nep_iar_run; exported as nep_refine_review).  Same verdict, same planned sweeps, same hint on a grid of omega
    sequences around every threshold of the rule (no GPU needed)."""
    import itertools
    from nep_amd import _lib
    from nep_amd.linsolvers import FactorizeLinSolver
    eps = np.finfo(float).eps
    vals = [0.0, 0.5 * eps, 1.9 * eps, 2.1 * eps, 3.9 * eps, 4.1 * eps, 1e-14, 0.49e-14, 0.51e-14, 1e-12, 1e-9, float("nan"), float("inf")]

    class _Nep:
        pass
    n = 0
    for umf in (1, 2, 10):
        for plan in (0, 1, 2, 3):
            for final in (True, False):
                for w in itertools.product(vals, repeat=plan + 1):
                    if final is False and plan == 0:
                        continue
                    s = FactorizeLinSolver.__new__(FactorizeLinSolver)
                    s.umfpack_refinements = umf; s._recorded_plan = None; s.last_omega = None; s.lam = 0.0
                    s.nep = _Nep()
                    w4 = np.zeros(4); w4[:plan + 1] = w
                    ok = s.review_recorded(w4, plan, final_recorded=final)
                    out = (_lib.c_i32 * 4)()
                    _lib.check(_lib.lib.nep_refine_review(umf, plan, 1 if final else 0, _lib.hptr(w4), -1, out))
                    hint = getattr(s.nep, "_refine_hint", None); off = getattr(s.nep, "_refine_hint_off", False)
                    assert bool(out[0]) == bool(ok), (umf, plan, final, w)
                    assert out[1] == (-1 if s._recorded_plan is None else s._recorded_plan), (umf, plan, final, w, out[1], s._recorded_plan)
                    assert out[2] == (-1 if hint is None else hint) and bool(out[3]) == bool(off), (umf, plan, final, w)
                    n += 1
    assert n > 10000
