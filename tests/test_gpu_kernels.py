"""GPU parity tests: every HIP kernel behind the C ABI against the CPU oracle / NumPy on the same
seeded inputs.  Tolerances are stated per test; all arithmetic is complex128."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def na():
    import nep_amd
    assert nep_amd.device_count() >= 1, "no GPU visible"
    return nep_amd


def _rand_spmf(n, mt, dens, seed, cplx_vals=False):
    from oracle import neps as oneps
    rng = np.random.default_rng(seed)
    AA = []
    for i in range(mt):
        A = sp.random(n, n, dens, random_state=seed * 10 + i, format="csc")
        if cplx_vals and i % 2 == 1:
            A = A + 1j * sp.random(n, n, dens, random_state=seed * 10 + i + 100, format="csc")
        AA.append(sp.csc_matrix(A))
    ofv = [oneps.f_one(), oneps.f_id(), oneps.f_isqrt(0.0), oneps.f_isqrt(-108.8774 ** 2), oneps.f_exp(-0.001),
           oneps.f_pow(2)][:mt]
    return AA, ofv, rng


def _pfv(na, mt):
    f = na.funcs
    return [f.one(), f.ident(), f.ISqrt(1.0, 0.0), f.ISqrt(1.0, -108.8774 ** 2), f.Exp(-0.001), f.Monomial(2)][:mt]


@pytest.mark.parametrize("k", [1, 2, 7, 8, 9, 33])      # 1: folded, 2..8: coefficient product fused, >8: k_vc + SpMV
@pytest.mark.parametrize("cplx_vals", [False, True])
def test_mlincomb_vs_oracle(na, k, cplx_vals):
    from oracle import neps as oneps
    n, mt = 203, 4
    AA, ofv, rng = _rand_spmf(n, mt, 0.05, 3, cplx_vals)
    onep = oneps.SPMF_NEP(AA, ofv)
    pnep = na.SPMF_NEP(AA, _pfv(na, mt))
    lam = 120.0 ** 2 + 3j
    V = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    a = rng.standard_normal(k) + 0j
    if k >= 7:
        a[2] = 0; a[5] = 0
    V0 = V.copy()
    z = pnep.compute_Mlincomb(lam, V, a)
    zo = onep.compute_Mlincomb(lam, V, a)
    assert np.array_equal(V, V0)                      # V must not be modified (test/spmf.jl:26-30)
    assert np.linalg.norm(z - zo) <= 1e-10 * np.linalg.norm(zo)
    # a = ones == no a  (test/core.jl:16-32)
    z1 = pnep.compute_Mlincomb(lam, V); z2 = pnep.compute_Mlincomb(lam, V, np.ones(k))
    assert np.array_equal(z1, z2)
    # startder
    if k <= 7:
        zs = pnep.compute_Mlincomb(lam, V, a, 2)
        zso = onep.compute_Mlincomb(lam, V, a, 2)
        assert np.linalg.norm(zs - zso) <= 1e-9 * np.linalg.norm(zso)


def test_mlincomb_dep_pep_sum(na):
    from oracle import gallery as og, neps as oneps
    rng = np.random.default_rng(1)
    # DEP dense n=5 (config C1 plumbing) and n=100
    for n in (5, 100):
        o = og.dep0(n); p = na.nep_gallery("dep0", n)
        V = rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4))
        a = np.array([1.0, -2.0, 0.5, 3.0])
        for lam in (1.0 + 1.0j, -0.3, 0.0):
            zo = o.compute_Mlincomb(lam, V, a); z = p.compute_Mlincomb(lam, V, a)
            assert np.linalg.norm(z - zo) <= 1e-13 * max(1.0, np.linalg.norm(zo))
    # KAT src/Gallery.jl:172-176
    p = na.nep_gallery("dep0", 100)
    assert np.linalg.norm(p.compute_Mlincomb(1.0 + 1.0j, np.ones(100))) == pytest.approx(57.498446538064954, rel=1e-13)
    # PEP + SPMF sum (native gun type) on a reduced twin
    o = og.nlevp_native_gun(655); p = na.nep_gallery("nlevp_native_gun", 655)
    V = rng.standard_normal((655, 3)) + 1j * rng.standard_normal((655, 3))
    lam = 250.0 ** 2 + 10j
    zo = o.compute_Mlincomb(lam, V); z = p.compute_Mlincomb(lam, V)
    assert np.linalg.norm(z - zo) <= 1e-12 * np.linalg.norm(zo)


def test_compute_MM_vs_oracle(na):
    from oracle import neps as oneps
    n, mt = 150, 4
    AA, ofv, rng = _rand_spmf(n, mt, 0.06, 5)
    onep = oneps.SPMF_NEP(AA, ofv); pnep = na.SPMF_NEP(AA, _pfv(na, mt))
    V = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    S = rng.standard_normal((3, 3)) + 120.0 ** 2 * np.eye(3)        # test/spmf.jl:127-156
    Zo = onep.compute_MM(S, V); Z = pnep.compute_MM(S, V)
    assert np.linalg.norm(Z - Zo) <= 1e-10 * np.linalg.norm(Zo)
    D = np.diag([1.0 + 2j, 3.0, -1.0])
    assert np.linalg.norm(pnep.compute_MM(D, V) - onep.compute_MM(D, V)) <= 1e-12 * np.linalg.norm(Zo)


@pytest.mark.parametrize("k", [1, 5, 64, 100, 130])
def test_resid_batch(na, k):
    n, mt = 317, 4
    AA, ofv, rng = _rand_spmf(n, mt, 0.04, 7, cplx_vals=(k == 5))
    pnep = na.SPMF_NEP(AA, _pfv(na, mt))
    Q = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    lams = 120.0 ** 2 + rng.standard_normal(k) * 100 + 1j * rng.standard_normal(k)
    import torch
    QT = torch.from_numpy(np.ascontiguousarray(Q)).to("cuda")
    errm = na.ResidualErrmeasure(pnep)
    e = errm.batch(list(lams), QT)
    fv = _pfv(na, mt)
    for s in range(k):
        r = sum(fv[i](lams[s]) * (AA[i] @ Q[:, s]) for i in range(mt))
        ref = np.linalg.norm(r) / np.linalg.norm(Q[:, s])
        assert e[s] == pytest.approx(ref, rel=1e-11)


@pytest.mark.parametrize("k", [1, 7, 64, 65, 128])
def test_resid_batch_long_and_empty_rows(na, k):
    """K2 on the row-major wave-per-row kernel (k_spmm_rm_g, round 3: gathers issued in groups, entries fetched a row ahead) with rows of
    more than 64 stacked entries (further chunks loaded inside the row), rows without entries and a complex term; k at the 64 / 128
    column boundaries of its two instantiations -- against NumPy"""
    import torch
    n, mt = 403, 4
    AA, ofv, rng = _rand_spmf(n, mt, 0.11, 11, cplx_vals=True)          # ~ 4 x 44 (+ complex part) = 180-270 entries per row
    AA = [sp.lil_matrix(A) for A in AA]
    for A in AA:
        for r in (0, 17, 200, n - 1):
            A[r, :] = 0                                              # rows without entries in every term
    AA = [sp.csc_matrix(A) for A in AA]
    for A in AA:
        A.eliminate_zeros()
    pnep = na.SPMF_NEP(AA, _pfv(na, mt))
    Q = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    lams = 120.0 ** 2 + rng.standard_normal(k) * 100 + 1j * rng.standard_normal(k)
    QT = torch.from_numpy(np.ascontiguousarray(Q)).to("cuda")
    e = na.ResidualErrmeasure(pnep).batch(list(lams), QT)
    fv = _pfv(na, mt)
    for s in range(k):
        r = sum(fv[i](lams[s]) * (AA[i] @ Q[:, s]) for i in range(mt))
        assert e[s] == pytest.approx(np.linalg.norm(r) / np.linalg.norm(Q[:, s]), rel=1e-11)
    # compute_MM runs its SpMM through the same kernel (coefficient 1, one column block per term)
    if k <= 7:
        S = rng.standard_normal((k, k)) + 120.0 ** 2 * np.eye(k)
        from oracle import neps as oneps
        Zo = oneps.SPMF_NEP(AA, ofv).compute_MM(S, Q)
        assert np.linalg.norm(pnep.compute_MM(S, Q) - Zo) <= 1e-10 * np.linalg.norm(Zo)


@pytest.mark.parametrize("rows,k,p", [(16, 4, 8), (100, 3, 5), (1000, 37, 41), (333, 100, 100), (257, 61, 120),
                                      (5, 2, 1), (4096, 16, 60)])
@pytest.mark.parametrize("rowmajor", [False, True])
def test_gemm_ts(na, rows, k, p, rowmajor):
    rng = np.random.default_rng(rows + k + p)
    Z = rng.standard_normal((rows, k)) + 1j * rng.standard_normal((rows, k))
    B = rng.standard_normal((k, p)) + 1j * rng.standard_normal((k, p))   # asymmetric, complex
    Y = na.gemm_ts(na.to_dev(Z), B, rowmajor=rowmajor)
    Yh = Y.cpu().numpy() if rowmajor else na.to_host(Y)
    ref = Z @ B
    assert np.linalg.norm(Yh - ref) <= 1e-13 * np.linalg.norm(ref) * np.sqrt(k)


def test_gemm_ts_ld(na):
    """leading dimension larger than rows (iar: first block row of V)"""
    rng = np.random.default_rng(0)
    ld, rows, k = 500, 123, 9
    Zfull = rng.standard_normal((ld, k)) + 1j * rng.standard_normal((ld, k))
    B = rng.standard_normal((k, k)) + 1j * rng.standard_normal((k, k))
    Y = na.gemm_ts(na.to_dev(Zfull), B, rowmajor=True, k=k, rows=rows, ldz=ld)
    assert np.allclose(Y.cpu().numpy(), Zfull[:rows] @ B, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_orth_vs_oracle(na, method):
    from oracle import solvers as osol
    rng = np.random.default_rng(5)
    rows, k = 5000, 13
    V, _ = np.linalg.qr(rng.standard_normal((rows, k)) + 1j * rng.standard_normal((rows, k)))
    w = rng.standard_normal(rows) + 1j * rng.standard_normal(rows)
    ofun = [osol.dgks, osol.cgs, osol.mgs][method]
    wo = w.copy(); ho = np.zeros(k, dtype=complex)
    bo = ofun(V, wo, ho)
    wd = na.to_dev(w)[0]
    h, beta, npass = na.orthogonalize_and_normalize(na.to_dev(V), wd, k, method=method)
    assert beta == pytest.approx(bo, rel=1e-12)
    assert np.linalg.norm(h - ho) <= 1e-12 * np.linalg.norm(ho)
    assert np.linalg.norm(wd.cpu().numpy() - wo) <= 1e-11
    assert np.linalg.norm(V.conj().T @ wd.cpu().numpy()) < 1e-13


def test_orth_dgks_forced_reorth_and_active(na):
    from oracle import solvers as osol
    rng = np.random.default_rng(6)
    n, k = 700, 6
    rows = n * (k + 1)
    # block-triangular basis like iar's: column j non-zero in the first (j+1)*n rows
    V = np.zeros((rows, k), dtype=complex)
    for j in range(k):
        V[:(j + 1) * n, j] = rng.standard_normal((j + 1) * n) + 1j * rng.standard_normal((j + 1) * n)
    V, _ = np.linalg.qr(V)      # QR of a block upper-triangular-profile matrix keeps the profile
    for j in range(k):
        assert np.all(V[(j + 1) * n:, j] == 0)
    # w almost inside span(V) -> DGKS must re-orthogonalise
    w = V @ (rng.standard_normal(k) + 1j * rng.standard_normal(k)) + 1e-9 * (rng.standard_normal(rows) + 0j)
    wo = w.copy(); ho = np.zeros(k, dtype=complex)
    bo = osol.dgks(V, wo, ho)
    wd = na.to_dev(w)[0]
    active = (np.arange(1, k + 1) * n).astype(np.int64)
    h, beta, npass = na.orthogonalize_and_normalize(na.to_dev(V), wd, k, active_rows=active)
    assert npass >= 2
    assert beta == pytest.approx(bo, rel=1e-6)
    assert np.linalg.norm(h - ho) <= 1e-12 * np.linalg.norm(ho)
    assert np.linalg.norm(V.conj().T @ wd.cpu().numpy()) < 1e-12
    # determinism: bitwise equal across repeats
    wd2 = na.to_dev(w)[0]
    h2, beta2, _ = na.orthogonalize_and_normalize(na.to_dev(V), wd2, k, active_rows=active)
    assert beta2 == beta and np.array_equal(h, h2) and np.array_equal(wd.cpu().numpy(), wd2.cpu().numpy())


@pytest.mark.parametrize("nrhs", [1, 3, 32])
def test_lu_solve_vs_host(na, nrhs):
    from oracle import gallery as og
    nep = og.nlevp_native_gun(1310)
    A = sp.csc_matrix(nep.compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    rng = np.random.default_rng(2)
    n = A.shape[0]
    B = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
    lu = na.DeviceLU(A)
    X = na.to_host(lu.solve(na.to_dev(B)))
    Xo = spla.splu(A).solve(B)
    assert np.linalg.norm(X - Xo) <= 1e-10 * np.linalg.norm(Xo)
    nA = abs(A).sum(axis=0).max()
    assert np.linalg.norm(A @ X - B) / (nA * np.linalg.norm(X)) < 1e-14      # test/linsolver.jl residual criterion
    # scale = -1, in place
    Bd = na.to_dev(B)
    lu.solve(Bd, out=Bd, scale=-1.0)
    assert np.linalg.norm(na.to_host(Bd) + Xo) <= 1e-10 * np.linalg.norm(Xo)


@pytest.mark.parametrize("nrhs", [8, 13, 32])
@pytest.mark.parametrize("bmax", [None, "96"])
def test_lu_solve_blocks_of_rhs_one_workgroup_per_block(na, monkeypatch, nrhs, bmax):
    """csrc/trsv_ml.hip k_ml_level_blk: blocks of right-hand sides (contour_beyn) on a level of many diagonal blocks -- one
    workgroup per block and 4 (or 8) right-hand sides, the block's r staged in LDS once -- against the chunk form
    (NEP_ML_BLK_RHS=0) and the host solve; forced onto every level (NEP_ML_BLK_RHS_MIN=0), with ragged groups of right-hand
    sides, in-place output with scale, and smaller diagonal blocks."""
    from oracle import gallery as og
    if bmax:
        monkeypatch.setenv("NEP_ML_BMAX", bmax)
    nep = og.nlevp_native_gun(2200)
    A = sp.csc_matrix(nep.compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    rng = np.random.default_rng(12)
    n = A.shape[0]
    B = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
    lu = na.DeviceLU(A)
    Xo = spla.splu(A).solve(B)
    res = {}
    for form, rows_min in (("0", "0"), ("4", "0"), ("8", "0"), ("4", "4000")):
        monkeypatch.setenv("NEP_ML_BLK_RHS", form); monkeypatch.setenv("NEP_ML_BLK_RHS_MIN", rows_min)
        X = na.to_host(lu.solve(na.to_dev(B)))
        assert np.linalg.norm(X - Xo) <= 1e-10 * np.linalg.norm(Xo)
        Bd = na.to_dev(B)
        lu.solve(Bd, out=Bd, scale=-0.5)
        assert np.linalg.norm(na.to_host(Bd) + 0.5 * X) <= 1e-13 * np.linalg.norm(X)
        res[(form, rows_min)] = X
    for key in (("4", "0"), ("8", "0")):
        assert np.linalg.norm(res[key] - res[("0", "0")]) <= 1e-12 * np.linalg.norm(Xo)
    # (a row threshold no level of this factor reaches: chunk form again; not compared bit for bit -- the factor's solves switch
    # to the dense apex inverse at the sixth solve)
    assert np.linalg.norm(res[("4", "4000")] - res[("0", "0")]) <= 1e-12 * np.linalg.norm(Xo)


@pytest.mark.parametrize("spec", [dict(), dict(NEP_LU_BLOCK="64", NEP_LU_MID="512", NEP_LU_TAIL="0"),
                                  dict(NEP_LU_BLOCK="128", NEP_LU_MID="384", NEP_LU_TAIL="100"), dict(NEP_LU_MID="0")])
def test_lu_blocked_mid_region(na, spec, monkeypatch):
    """every schedule shape of the hybrid solve (level heads / blocked mid with inverted diagonal blocks / dense tail)
    returns the host solution; nrhs 1 and 5; tolerance 1e-9 relative (explicit block inverses)"""
    monkeypatch.setenv("NEP_LU_SCHED", "old")             # the level schedule stays as the fallback of the block schedule
    for k_, v_ in spec.items():
        monkeypatch.setenv(k_, v_)
    from oracle import gallery as og
    nep = og.nlevp_native_gun(1310)
    A = sp.csc_matrix(nep.compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    lu = na.DeviceLU(A, expected_solves=200)
    assert not lu.block_schedule
    if "NEP_LU_MID" in spec and "NEP_LU_BLOCK" in spec:
        assert lu.mid_block == int(spec["NEP_LU_BLOCK"]) and lu.mid_rows == int(spec["NEP_LU_MID"])
        assert lu.tail == int(spec["NEP_LU_TAIL"])
    if spec.get("NEP_LU_MID") == "0":
        assert lu.mid_rows == 0
    host = spla.splu(A)
    for nrhs in (1, 5):
        B = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
        X = na.to_host(lu.solve(na.to_dev(B)))
        Xo = host.solve(B)
        assert np.linalg.norm(X - Xo) <= 1e-9 * np.linalg.norm(Xo)


@pytest.mark.parametrize("spec", [dict(), dict(NEP_ML_BMAX="64"), dict(NEP_ML_BMAX="32", NEP_ML_CHUNK="32"),
                                  dict(NEP_ML_SPLIT="1"), dict(NEP_ML_SPLIT="0", NEP_ML_CHUNK="4"), dict(NEP_NO_GRAPH="1"),
                                  dict(NEP_ML_NOCACHE="1"), dict(NEP_ML_APEX="0"), dict(NEP_ML_APEX="1", NEP_ML_SPLIT="0"),
                                  dict(NEP_ML_APEX="2", NEP_ML_BMAX="64")])
def test_lu_block_schedule(na, spec, monkeypatch):
    """K5 block schedule (csrc/trsv_ml.hip): every shape of the schedule (block size, chunking, fused / split coupling
    product, graph / eager) returns SuperLU's solution; CSR and CSC input; nrhs 1, 5, 32; in-place, scale, fused update"""
    for k_, v_ in spec.items():
        monkeypatch.setenv(k_, v_)
    import nep_amd_hostlu as hl
    from oracle import gallery as og
    nep = og.nlevp_native_gun(1310)
    A = sp.csc_matrix(nep.compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    host = spla.splu(A)
    for csr in (False, True):
        F = hl.factor(A.data, A.indices, A.indptr, A.shape, csr=csr)
        lu = na.DeviceLU(factors=F, expected_solves=200)
        assert lu.block_schedule and lu.mid_block <= int(spec.get("NEP_ML_BMAX", 256))
        for nrhs in (1, 5, 32):
            B = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
            Xo = host.solve(B)
            for rep in range(2):                       # second call replays the captured graph
                X = na.to_host(lu.solve(na.to_dev(B)))
                assert np.linalg.norm(X - Xo) <= 1e-9 * np.linalg.norm(Xo)
            Bd = na.to_dev(B)
            lu.solve(Bd, out=Bd, scale=-1.0)           # in place
            assert np.linalg.norm(na.to_host(Bd) + Xo) <= 1e-9 * np.linalg.norm(Xo)
            Add = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
            Ad = na.to_dev(Add)
            lu.solve_add(na.to_dev(B), Ad, Ad, 2.0)    # out aliases add
            assert np.linalg.norm(na.to_host(Ad) - 2.0 * (Add + Xo)) <= 1e-9 * np.linalg.norm(Xo)
        assert lu.launches_last_solve() <= 2 * lu.levels + lu.split_levels
        if spec.get("NEP_ML_APEX", "0") != "0":
            assert lu.tail > 0                             # the last levels are one dense inverse


def test_lu_refactor_rowscale_and_pattern_cache(na, monkeypatch):
    """nep_lu_refactor (same pattern, new values: src/method_beyncontour.jl:89-94), nep_lu_set_row_scale (UMFPACK's Rs) and
    the pattern-hash cache of the symbolic analysis; destroying a factorisation right after an asynchronous solve is safe
    (stream-ordered frees)"""
    import nep_amd_hostlu as hl
    from oracle import gallery as og
    nep = og.nlevp_native_gun(1310)
    n = nep.n
    rng = np.random.default_rng(11)
    A1 = sp.csc_matrix(nep.compute_Mder(250.0 ** 2 + 1j), dtype=complex)
    A2 = sp.csc_matrix(nep.compute_Mder(260.0 ** 2 + 5j), dtype=complex)
    F1 = hl.factor(A1.data, A1.indices, A1.indptr, A1.shape)
    F2 = hl.factor(A2.data, A2.indices, A2.indptr, A2.shape)
    same = all(np.array_equal(F1[k], F2[k]) for k in ("Lp", "Li", "Up", "Ui", "perm_r", "perm_c"))
    B = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    lu = na.DeviceLU(factors=F1)
    X1 = na.to_host(lu.solve(na.to_dev(B)))
    assert np.linalg.norm(A1 @ X1 - B) <= 1e-9 * np.linalg.norm(B)
    if same:                                             # diagonal pivoting: the two shifts share the factor pattern
        lu.refactor(F2["Lx"], F2["Ux"])
        X2 = na.to_host(lu.solve(na.to_dev(B)))
        assert np.linalg.norm(A2 @ X2 - B) <= 1e-9 * np.linalg.norm(B)
        lu.refactor(F1["Lx"], F1["Ux"])
        assert np.array_equal(na.to_host(lu.solve(na.to_dev(B))), X1)        # deterministic kernels
    # row scaling: factors of diag(rs) A  ->  solve multiplies b by rs
    rs = 0.5 + rng.random(n)
    As = sp.csc_matrix(sp.diags(rs) @ A1)
    Fs = hl.factor(As.data, As.indices, As.indptr, As.shape)
    lus = na.DeviceLU(factors=Fs)
    lus.set_row_scale(rs)
    Xs = na.to_host(lus.solve(na.to_dev(B)))
    assert np.linalg.norm(A1 @ Xs - B) <= 1e-9 * np.linalg.norm(B)
    lus.set_row_scale(None)
    Xn = na.to_host(lus.solve(na.to_dev(B)))
    assert np.linalg.norm(As @ Xn - B) <= 1e-9 * np.linalg.norm(B)
    # create / solve / drop in a loop: blocks return to the pool while their solve may still be running
    outs = []
    for i in range(6):
        t = na.DeviceLU(factors=F1 if i % 2 == 0 else F2, expected_solves=1)
        outs.append((i, t.solve(na.to_dev(B))))
        del t
    for i, Xd in outs:
        Aref = A1 if i % 2 == 0 else A2
        assert np.linalg.norm(Aref @ na.to_host(Xd) - B) <= 1e-9 * np.linalg.norm(B)
    # singular U is reported at the numeric stage
    Fz = dict(F1); Ux = F1["Ux"].copy()
    Uc = sp.csc_matrix((np.arange(len(Ux)), F1["Ui"], F1["Up"]), shape=(n, n))
    Ux[int(Uc[5, 5])] = 0.0
    Fz["Ux"] = Ux
    with pytest.raises(na.NepError) as ei:
        na.DeviceLU(factors=Fz)
    assert ei.value.status == -3


def test_cw_backward_error_and_refinement(na):
    """nep_cw_backward_error against NumPy (r exact to round-off, omega to 1e-12 relative) and the UMFPACK-style
    refinement loop of FactorizeLinSolver: final componentwise backward error < 10 eps, and refinement never makes the
    residual worse (test/linsolver.jl:77-94)."""
    import ctypes as C
    from nep_amd import _lib
    AA, ofv, rng = _rand_spmf(400, 3, 0.02, 11, cplx_vals=True)
    AA[0] = sp.csc_matrix(AA[0] + 5 * sp.eye(400))
    nep = na.SPMF_NEP(AA, _pfv(na, 3))
    lam = 0.7 + 0.2j
    n = 400
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    cabs = np.array([abs(f.derivs(lam, 1)[0]) for f in nep.get_fv()])
    xd, bd = na.to_dev(x), na.to_dev(b)
    Mx = nep.compute_Mlincomb(lam, xd)
    import torch
    r = torch.empty_like(xd)
    om = np.zeros(1)
    st = __import__("nep_amd").nep.stream_ptr()
    _lib.check(_lib.lib.nep_cw_backward_error(nep.dev.h, _lib.hptr(cabs), None, C.c_void_p(xd.data_ptr()),
                                              C.c_void_p(bd.data_ptr()), C.c_void_p(Mx.data_ptr()), None,
                                              C.c_void_p(r.data_ptr()), _lib.hptr(om), st))
    # fused variant: M x formed inside the kernel from the complex coefficients
    cf = np.array([f.derivs(lam, 1)[0] for f in nep.get_fv()], dtype=complex)
    r2 = torch.empty_like(xd)
    om2 = np.zeros(1)
    _lib.check(_lib.lib.nep_cw_backward_error(nep.dev.h, _lib.hptr(cabs), _lib.hptr(cf), C.c_void_p(xd.data_ptr()),
                                              C.c_void_p(bd.data_ptr()), None, None, C.c_void_p(r2.data_ptr()),
                                              _lib.hptr(om2), st))
    assert np.linalg.norm(na.to_host(r2) - na.to_host(r)) <= 1e-13 * np.linalg.norm(na.to_host(r))
    assert abs(om2[0] - om[0]) <= 1e-12 * om[0]
    M = sum(f.derivs(lam, 1)[0] * A for f, A in zip(nep.get_fv(), AA))
    r_ref = b - M @ x
    den = sum(c * (abs(A) @ abs(x)) for c, A in zip(cabs, AA)) + abs(b)
    assert np.linalg.norm(na.to_host(r)[:, 0] - r_ref) <= 1e-13 * np.linalg.norm(r_ref)
    assert abs(om[0] - np.max(abs(r_ref) / den)) <= 1e-12 * om[0]
    # refinement
    res = []
    for steps in (0, 1, 10):
        ls = na.create_linsolver(na.FactorizeLinSolverCreator(umfpack_refinements=steps), nep, lam)
        xs = na.lin_solve(ls, b)
        res.append(np.linalg.norm(M @ xs - b))
        if steps == 10:
            assert ls.last_omega < 10 * np.finfo(float).eps
    assert res[1] <= res[0] * 1.5 and res[2] <= res[0] * 1.5


def test_lin_solve_interfaces(na):
    """Backslash == Factorize == host solve (test/linsolver.jl:11-69) on a gun twin; dense dep0 too."""
    nep = na.nep_gallery("nlevp_native_gun", 655)
    lam = 250.0 ** 2 + 1j
    b = np.arange(1, 656) + 0j
    x1 = na.lin_solve(na.create_linsolver(na.FactorizeLinSolverCreator(), nep, lam), b)
    x2 = na.lin_solve(na.create_linsolver(na.BackslashLinSolverCreator(), nep, lam), b)
    M = sp.csc_matrix(nep.compute_Mder(lam))
    x3 = spla.spsolve(M, b)
    assert np.linalg.norm(x1 - x3) <= 1e-10 * np.linalg.norm(x3)
    assert np.linalg.norm(x2 - x3) <= 1e-10 * np.linalg.norm(x3)
    B = np.column_stack([b, 2 * b[::-1]])
    X = na.lin_solve(na.create_linsolver(na.FactorizeLinSolverCreator(), nep, lam), B)
    assert X.shape == B.shape and np.linalg.norm(M @ X - B) <= 1e-9 * np.linalg.norm(B)
    d = na.nep_gallery("dep0")
    xs = na.lin_solve(na.create_linsolver(na.DefaultLinSolverCreator(), d, 0.3), np.ones(5))
    assert np.allclose(d.compute_Mder(0.3) @ xs, np.ones(5))
    # factorisation recycling: sizes 0 -> 1 -> 1 -> 2 (test/rk_helper/cached_lin_solver.jl)
    cache = na.LinSolverCache(nep, na.FactorizeLinSolverCreator())
    assert len(cache.solvers) == 0
    cache.solve(lam, b, True); assert len(cache.solvers) == 1
    cache.solve(lam, b, True); assert len(cache.solvers) == 1
    cache.solve(lam + 1, b, False); assert len(cache.solvers) == 1
    cache.solve(lam + 2, b, True); assert len(cache.solvers) == 2


def test_mlincomb_sell_and_fold_paths(na, monkeypatch):
    """large-n SELL-64 SpMV, the k==1 folded and the 2<=k<=16 fused forms agree with the CSR-vector path and with NumPy"""
    from nep_amd import wep
    wd = wep.WaveguideData(61, 57, "JARLEBRING")
    Av = wd.big_matrices()
    n = wd.n
    rng = np.random.default_rng(4)
    fv = [na.funcs.one(), na.funcs.ident(), na.funcs.Monomial(2)]
    lam = -1.3 - 0.31j
    Vs = {k: rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k)) for k in (1, 5, 16, 17)}
    res = {}
    for sell in ("0", "1"):
        monkeypatch.setenv("NEP_SELL", sell)
        nep = na.SPMF_NEP(Av, fv)
        for k, V in Vs.items():
            a = np.arange(1, k + 1) + 0.5j
            z = nep.compute_Mlincomb(lam, V, a)
            ref = sum(a[j] * sum(fv[i].derivs(lam, k)[j] * (Av[i] @ V[:, j]) for i in range(3)) for j in range(k))
            assert np.linalg.norm(z - ref) <= 1e-12 * np.linalg.norm(ref)
            if (k, "0") in res:
                assert np.linalg.norm(z - res[(k, "0")]) <= 1e-13 * np.linalg.norm(ref)
            res[(k, sell)] = z


def test_lu_single_launch_form_opt_in(na, monkeypatch):
    """NEP_ML_FUSE=1: all phases of a single-vector solve as one kernel (tickets + phase counters); same solution as the
    multi-launch schedule.  Kept opt-in because it is 20x slower on this part -- the test keeps it correct."""
    import scipy.sparse as sp
    from oracle import gallery as og
    A = sp.csc_matrix(og.gun_spmf_scaled(1310).compute_Mder(0.1 + 0.02j))
    rng = np.random.default_rng(3)
    b = rng.standard_normal(A.shape[0]) + 1j * rng.standard_normal(A.shape[0])
    import torch
    lu = na.DeviceLU(A)
    bd = torch.from_numpy(b).to("cuda")
    x0 = lu.solve(bd).cpu().numpy()
    assert lu.launches_last_solve() > 1
    monkeypatch.setenv("NEP_ML_FUSE", "1")
    for _ in range(3):                       # counters are monotonic over solves
        x1 = lu.solve(bd).cpu().numpy()
        assert lu.launches_last_solve() == 1
        assert np.linalg.norm(x1 - x0) <= 1e-12 * np.linalg.norm(x0)
    assert np.linalg.norm(A @ x1 - b) <= 1e-9 * np.linalg.norm(b)


def test_error_paths_through_the_c_abi(na):
    """status codes instead of crashes: singular factorisation (SingularException like `lu` in the reference, hence
    LinAlgError), structurally singular U handed to nep_lu_create (NEP_ERR_SINGULAR), non-triangular factors and bad
    leading dimensions (NEP_ERR_ARG), orthogonalisation breakdown on a zero vector (NEP_ERR_BREAKDOWN), and the
    asynchronous DGKS reporting the same breakdown through its status word"""
    import ctypes as C
    import torch
    from nep_amd import _lib
    from nep_amd._lib import lib, hptr, c_vp
    st = __import__("nep_amd").nep.stream_ptr()
    # numerically singular matrix: the host factorisation raises, surfaced as LinAlgError("SingularException...")
    A = sp.csc_matrix(np.array([[1.0, 2.0, 0], [2.0, 4.0, 0], [0, 0, 1.0]]), dtype=complex)
    with pytest.raises(np.linalg.LinAlgError):
        na.DeviceLU(A)
    # zero pivot in U passed directly to the library
    n = 4
    Lp = np.arange(n + 1, dtype=np.int32); Li = np.arange(n, dtype=np.int32); Lx = np.ones(n, dtype=complex)
    Ux = np.ones(n, dtype=complex); Ux[2] = 0.0
    h = C.c_void_p()
    rc = lib.nep_lu_create(n, hptr(Lp), hptr(Li), hptr(Lx), hptr(Lp), hptr(Li), hptr(Ux), None, None, C.byref(h))
    assert rc == _lib.NEP_ERR_SINGULAR and not h.value
    # L with an entry above the diagonal
    Lp2 = np.array([0, 2, 3, 4, 5], dtype=np.int32); Li2 = np.array([0, 3, 1, 2, 3], dtype=np.int32); Lx2 = np.ones(5, dtype=complex)
    rc = lib.nep_lu_create(n, hptr(Lp2), hptr(Li2), hptr(Lx2), hptr(Lp), hptr(Li), hptr(Lx), None, None, C.byref(h))
    assert rc == _lib.NEP_ERR_ARG
    with pytest.raises(na.NepError):
        _lib.check(rc)
    # gemm with a leading dimension smaller than the row count
    Z = torch.zeros((3, 10), dtype=torch.complex128, device="cuda")
    B = np.ones((3, 2), dtype=complex, order="F")
    Y = torch.zeros((2, 10), dtype=torch.complex128, device="cuda")
    rc = lib.nep_gemm_ts(c_vp(Z.data_ptr()), 5, 10, 3, hptr(B), 3, 2, c_vp(Y.data_ptr()), 10, 0, st)
    assert rc == _lib.NEP_ERR_ARG
    # orthogonalising the zero vector: breakdown, synchronous and asynchronous flavour
    V = torch.zeros((2, 64), dtype=torch.complex128, device="cuda"); V[0, 0] = 1; V[1, 1] = 1
    w = torch.zeros(64, dtype=torch.complex128, device="cuda")
    with pytest.raises(na.NepError) as ei:
        na.orthogonalize_and_normalize(V, w, 2)
    assert ei.value.status == _lib.NEP_ERR_BREAKDOWN
    out = torch.zeros(4, dtype=torch.complex128, device="cuda")
    na.dense.orthogonalize_and_normalize_dev(V, w, 2, out)
    flags = na.to_host(out.reshape(1, -1))[:, 0][3]
    assert int(flags.imag) & 2


@pytest.mark.parametrize("rows,k,p", [(16, 3, 5), (1000, 37, 61), (333, 100, 100), (70000, 16, 17), (4097, 129, 33)])
def test_gemm_h_rm(na, rows, k, p):
    """K9 C = W^H Y on the FP64 matrix cores against NumPy (1e-12 relative), incl. sizes that are no multiples of the
    16 x 16 tile / 4-row step and leading dimensions larger than the block"""
    import torch
    rng = np.random.default_rng(rows + k)
    W = rng.standard_normal((rows, k)) + 1j * rng.standard_normal((rows, k))
    Y = rng.standard_normal((rows, p)) + 1j * rng.standard_normal((rows, p))
    ldw, ldy = k + 3, p + 1
    Wp = np.zeros((rows, ldw), dtype=complex); Wp[:, :k] = W
    Yp = np.zeros((rows, ldy), dtype=complex); Yp[:, :p] = Y
    WT = torch.from_numpy(Wp).to("cuda"); YT = torch.from_numpy(Yp).to("cuda")
    Cm = na.dense.gemm_h_rm(WT, YT, rows, k, p, ldw=ldw, ldy=ldy)
    Cref = W.conj().T @ Y
    assert np.linalg.norm(Cm - Cref) <= 1e-12 * np.linalg.norm(Cref)


def test_randomized_sweep_small(na):
    """a short run of scripts/diag/fuzz_kernels.py: K1/K2/K5/K6/K7/K9 against NumPy on ragged and degenerate sizes
    (n in {1,2,3,5,17,...}, empty matrices, zero coefficients, complex terms); 180 cases were run clean when it was written"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag", "fuzz_kernels.py"), "7", "12"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "done, failures: 0" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("rows,cols,dens", [(84, 9956, 0.01), (9956, 84, 0.01), (500, 300, 0.3), (64, 64, 1.0), (3, 1000, 0.9),
                                            (1, 1, 1.0), (40, 7, 0.0)])
def test_csr_mv_vs_scipy(na, rows, cols, dens):
    """nep_csr_mv: y = alpha A x + beta z on rectangular complex CSR operators (the UU^H / [L_1..L_q] factors of the
    low-rank NLEIGS, rk_nep.jl:128-152): every lanes-per-row variant (4 / 16 / 64), empty rows, an empty matrix, aliasing
    of z and y, and beta = 0 with NaN in y (must not be read).  Tolerance 4 eps * nnz-per-row growth -> 1e-14 relative."""
    import torch
    from nep_amd import nep as nn
    rng = np.random.default_rng(rows + cols)
    A = sp.random(rows, cols, density=dens, random_state=1, format="csr") + 1j * sp.random(rows, cols, density=dens, random_state=2, format="csr")
    op = nn.DeviceCSR(A)
    x = rng.standard_normal(cols) + 1j * rng.standard_normal(cols)
    z = rng.standard_normal(rows) + 1j * rng.standard_normal(rows)
    xd = nn.to_dev(x)[0]; zd = nn.to_dev(z)[0]
    yd = torch.full((rows,), float("nan"), dtype=torch.complex128, device="cuda")
    op.mv(0.3 - 0.2j, xd, 0.0, yd, yd)
    ref = (0.3 - 0.2j) * (A @ x)
    assert np.abs(yd.cpu().numpy() - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max())
    op.mv(1.5, xd, -0.5j, zd, zd)
    ref2 = 1.5 * (A @ x) - 0.5j * z
    assert np.abs(zd.cpu().numpy() - ref2).max() <= 1e-14 * np.abs(ref2).max()
    out = torch.empty(rows, dtype=torch.complex128, device="cuda")
    op.mv(1.0, xd.data_ptr(), 2.0, nn.to_dev(z)[0], out)          # raw device addresses are accepted
    assert np.abs(out.cpu().numpy() - (A @ x + 2 * z)).max() <= 1e-14 * max(1.0, np.abs(A @ x + 2 * z).max())


@pytest.mark.parametrize("mt", [33, 83, 128])
def test_many_terms(na, mt):
    """stacked CSR with more than 32 terms (7 term bits, <= 128; the particle example of test/nleigs has 83): K1 for
    k = 1 (fold) and k = 5, K2 norms over 100 vectors -- several column panels, since the mt x panel coefficient block must
    fit the LDS budget -- synchronous and asynchronous, the residual block, and the refinement criterion of a solve"""
    import torch
    n = 257
    rng = np.random.default_rng(mt)
    AA = [sp.random(n, n, 0.02, random_state=mt * 7 + i, format="csc") + (sp.identity(n) * (1.0 + i) if i == 0 else 0 * sp.identity(n))
          for i in range(mt)]
    AA = [sp.csc_matrix(A) for A in AA]
    fv = [na.funcs.Exp(-0.01 * (i + 1)) if i % 2 else na.funcs.Monomial(i % 3) for i in range(mt)]
    nep = na.SPMF_NEP(AA, fv)
    lam = 0.3 + 0.1j
    for k in (1, 5):
        V = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
        a = rng.standard_normal(k)
        z = nep.compute_Mlincomb(lam, V, a)
        ref = sum(sum(a[j] * fv[i].derivs(lam, k)[j] * (AA[i] @ V[:, j]) for j in range(k)) for i in range(mt))
        assert np.linalg.norm(z - ref) <= 1e-12 * np.linalg.norm(ref)
    k = 100
    Q = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    lams = 0.2 + 0.3 * rng.standard_normal(k) + 0.1j * rng.standard_normal(k)
    QT = torch.from_numpy(np.ascontiguousarray(Q)).to("cuda")
    R = np.stack([sum(fv[i](lams[s]) * (AA[i] @ Q[:, s]) for i in range(mt)) for s in range(k)], axis=1)
    refn = np.linalg.norm(R, axis=0) / np.linalg.norm(Q, axis=0)
    e = na.ResidualErrmeasure(nep).batch(list(lams), QT)
    assert np.allclose(e, refn, rtol=1e-11)
    rn, qn, _ = nep.resid_norms_async(lams, QT).get()
    assert np.allclose(rn / qn, refn, rtol=1e-11)
    from nep_amd._lib import lib, check, hptr, c_vp
    F = np.asfortranarray(np.array([[fv[i](lams[s_]) for s_ in range(k)] for i in range(mt)], dtype=np.complex128))
    RT = torch.empty((n, k), dtype=torch.complex128, device="cuda")
    check(lib.nep_resid_block(nep.dev.h, k, hptr(F), c_vp(QT.data_ptr()), k, c_vp(RT.data_ptr()), k, None))
    torch.cuda.synchronize()
    assert np.linalg.norm(RT.cpu().numpy() - R) <= 1e-12 * np.linalg.norm(R)
    solver = na.create_linsolver(na.FactorizeLinSolverCreator(), nep, lam)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x = na.lin_solve(solver, b)
    M = sum(fv[i](lam) * AA[i] for i in range(mt))
    assert np.linalg.norm(M @ x - b) <= 1e-12 * np.linalg.norm(b) and solver.last_omega is not None and solver.last_omega < 1e-14


@pytest.mark.parametrize("ta,tb", [(0, 0), (2, 0), (0, 1), (1, 2)])
def test_zgemm_vs_numpy(na, ta, tb):
    """nep_zgemm (the library's own LDS-tiled GEMM behind the C ABI, csrc/gemm.hip k_gemm_general; no vendor BLAS): C = alpha op(A) op(B) + beta C with none / transpose / conjugate
    transpose, leading dimensions larger than the matrices; 1e-13 relative (k = 77 products per entry)"""
    import torch
    from nep_amd.wep_linsolvers import zgemm
    rng = np.random.default_rng(10 * ta + tb)
    m, n, k = 53, 41, 77
    op = lambda X, t: X if t == 0 else (X.T if t == 1 else X.conj().T)
    A = rng.standard_normal((m, k) if ta == 0 else (k, m)) + 1j * rng.standard_normal((m, k) if ta == 0 else (k, m))
    B = rng.standard_normal((k, n) if tb == 0 else (n, k)) + 1j * rng.standard_normal((k, n) if tb == 0 else (n, k))
    C0 = rng.standard_normal((m + 3, n)) + 1j * rng.standard_normal((m + 3, n))
    Ad, Bd, Cd = na.to_dev(A), na.to_dev(B), na.to_dev(C0)
    alpha, beta = 0.7 - 0.2j, -0.3 + 1.1j
    zgemm(ta, tb, m, n, k, alpha, Ad, A.shape[0], Bd, B.shape[0], beta, Cd, m + 3)
    ref = C0.copy(); ref[:m] = alpha * (op(A, ta) @ op(B, tb)) + beta * C0[:m]
    assert np.linalg.norm(na.to_host(Cd) - ref) <= 1e-13 * np.linalg.norm(ref)
    with pytest.raises(na.NepError):
        zgemm(0, 0, m, n, k, 1.0, Ad, m - 1 if ta == 0 else 1, Bd, B.shape[0], 0.0, Cd, m + 3)      # lda too small


@pytest.mark.parametrize("rows,k", [(999, 999), (7, 7), (1517, 1517), (65, 3), (1, 1)])
def test_gemv_hd_vs_numpy(na, rows, k):
    """nep_gemv_hd: y = d .* (A^H x) with the result on the device (boundary operator P^{-1} of the waveguide problem and
    the SMW coefficient solve), with and without the diagonal factor, lda > rows; 1e-13 relative"""
    import torch
    from nep_amd._lib import lib, check, c_vp
    rng = np.random.default_rng(rows * 7 + k)
    lda = rows + 5
    A = rng.standard_normal((lda, k)) + 1j * rng.standard_normal((lda, k))
    x = rng.standard_normal(rows) + 1j * rng.standard_normal(rows)
    d = rng.standard_normal(k) + 1j * rng.standard_normal(k)
    Ad, xd, dd = na.to_dev(A), na.to_dev(x)[0], na.to_dev(d)[0]
    y = torch.full((k,), float("nan"), dtype=torch.complex128, device="cuda")
    ref = A[:rows].conj().T @ x
    check(lib.nep_gemv_hd(c_vp(Ad.data_ptr()), lda, rows, k, c_vp(xd.data_ptr()), None, c_vp(y.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.linalg.norm(y.cpu().numpy() - ref) <= 1e-13 * np.linalg.norm(ref)
    check(lib.nep_gemv_hd(c_vp(Ad.data_ptr()), lda, rows, k, c_vp(xd.data_ptr()), c_vp(dd.data_ptr()), c_vp(y.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.linalg.norm(y.cpu().numpy() - d * ref) <= 1e-13 * np.linalg.norm(d * ref)


def test_plain_c_smoke_program():
    """examples/smoke_c.c: the C ABI driven from a plain C process (no Python, no torch): nep_spmf_create -> nep_mlincomb ->
    nep_lu_create_csc / nep_lu_solve -> nep_orth -> nep_gemm_ts, each checked against host arithmetic inside the program"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "smoke_c")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "smoke_c.c"), "-o", exe,
                               "-L", os.path.join(root, "nonlineareigenproblems.jl_amd"), "-lnepmi355", "-lm",
                               "-Wl,-rpath,$ORIGIN/../nonlineareigenproblems.jl_amd"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "smoke_c ok" in p.stdout


def test_concurrent_lu_create_solve_stress(na):
    """the path Beyn uses: several host threads create, solve with and drop factorisations at the same time (shared pattern
    -> shared symbolic analysis from the cache, different values; per-build streams; stream-ordered pool frees) while the
    main thread keeps solving with a long-lived factorisation.  Every result is checked."""
    import threading
    import nep_amd_hostlu as hl
    from oracle import gallery as og
    nep = og.nlevp_native_gun(1310)
    n = nep.n
    rng = np.random.default_rng(17)
    shifts = [250.0 ** 2 + 1j, 255.0 ** 2 + 3j, 245.0 ** 2 + 0.5j, 262.0 ** 2 + 2j]
    mats = [sp.csc_matrix(nep.compute_Mder(s), dtype=complex) for s in shifts]
    facs = [hl.factor(A.data, A.indices, A.indptr, A.shape) for A in mats]
    B = rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4))
    Bd = na.to_dev(B)
    errors = []

    def worker(tid):
        try:
            import torch
            torch.cuda.set_device(0)
            for it in range(12):
                i = (tid + it) % len(facs)
                lu = na.DeviceLU(factors=facs[i], expected_solves=1 if it % 2 else 200)
                X = lu.solve(Bd)
                if it % 3 == 0:
                    lu.refactor(facs[i]["Lx"], facs[i]["Ux"])
                    X = lu.solve(Bd)
                Xh = na.to_host(X)
                del lu
                r = np.linalg.norm(mats[i] @ Xh - B) / np.linalg.norm(B)
                if not r < 1e-9:
                    errors.append((tid, it, r))
        except Exception as e:                      # noqa: BLE001
            errors.append((tid, repr(e)))

    main_lu = na.DeviceLU(factors=facs[0], expected_solves=200)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in th:
        t.start()
    for _ in range(200):
        Xm = main_lu.solve(Bd)
    for t in th:
        t.join()
    assert not errors, errors[:3]
    assert np.linalg.norm(mats[0] @ na.to_host(Xm) - B) <= 1e-9 * np.linalg.norm(B)


@pytest.mark.parametrize("n", [1310, 9956])
def test_device_plan_enumeration_equals_host_enumeration(na, n):
    """csrc/lufac.hip, round 3: the products of the refactorisation plan are enumerated, classified and placed ON THE DEVICE
    (k_lu_enum_*, one scan + one radix sort).  The plan arrays downloaded from the device hash to the value of the host-only
    enumeration of the same inputs (nep_lu_refac_analyze), i.e. they are bit-identical; gun pattern at test size and at full size."""
    import ctypes as C
    import scipy.sparse as sp
    from oracle import gallery as og
    from nep_amd._lib import lib, check, hptr, c_vp
    import nep_amd_hostlu as hl
    onep = og.gun_spmf_scaled(n)
    A0 = sp.csc_matrix(onep.compute_Mder(0.0)).astype(np.complex128)
    F = hl.factor(A0.data, A0.indices, A0.indptr, A0.shape)
    ref = na.DeviceLU(factors=F)
    arrs = [np.ascontiguousarray(F[k], dtype=np.int32) for k in ("Lp", "Li", "Up", "Ui", "perm_r", "perm_c")]
    Ap = np.ascontiguousarray(A0.indptr, dtype=np.int32); Ai = np.ascontiguousarray(A0.indices, dtype=np.int32)
    ana = (C.c_int64 * 8)()
    check(lib.nep_lu_refac_analyze(n, *[hptr(a) for a in arrs], hptr(Ap), hptr(Ai), ana))
    h = c_vp()
    check(lib.nep_lu_refac_create(ref.h, n, *[hptr(a) for a in arrs], hptr(Ap), hptr(Ai), C.byref(h)))
    info = (C.c_int64 * 6)(); check(lib.nep_lu_refac_info(h, info))
    hh = (C.c_int64 * 2)(); check(lib.nep_lu_refac_hash(h, hh))
    lib.nep_lu_refac_destroy(h)
    assert hh[1] == 1, "the plan was not enumerated on the device"
    assert (info[1], info[2], info[3], info[4]) == (ana[0], ana[1], ana[2], ana[3])
    assert hh[0] == ana[7]


def test_device_numeric_factorization(na, monkeypatch):
    """csrc/lufac.hip: the second matrix of a sparsity pattern is factorised on the GPU with the pivot sequence of the first
    (host) factorisation: L and U equal the host factor's to round-off, the solves agree, a growth limit that is not met
    sends the matrix back to the host path"""
    import ctypes as C
    import scipy.sparse as sp
    import torch
    from oracle import gallery as og
    from nep_amd import linsolvers as ls
    from nep_amd._lib import lib, check, hptr, c_vp
    from nep_amd.nep import stream_ptr
    import nep_amd_hostlu as hl
    ls._DeviceRefactor.clear()
    onep = og.gun_spmf_scaled(1310)
    A0 = sp.csc_matrix(onep.compute_Mder(0.0)).astype(np.complex128)
    A1 = sp.csc_matrix(onep.compute_Mder(0.15 + 0.05j)).astype(np.complex128)
    assert np.array_equal(A0.indices, A1.indices)
    # (i) raw API: values of L and U against the host factor of the same matrix
    F = hl.factor(A0.data, A0.indices, A0.indptr, A0.shape)
    ref = na.DeviceLU(factors=F)
    h = c_vp()
    check(lib.nep_lu_refac_create(ref.h, A0.shape[0], hptr(F["Lp"]), hptr(F["Li"]), hptr(F["Up"]), hptr(F["Ui"]), hptr(F["perm_r"]),
                                  hptr(F["perm_c"]), hptr(np.ascontiguousarray(A0.indptr, dtype=np.int32)),
                                  hptr(np.ascontiguousarray(A0.indices, dtype=np.int32)), C.byref(h)))
    info = (C.c_int64 * 6)(); check(lib.nep_lu_refac_info(h, info))
    assert info[1] == info[2] + info[3] + (info[1] - info[2] - info[3]) and info[1] > 0
    n = A0.shape[0]; nL = len(F["Lx"])
    for A in (A0, A1):
        Fh = hl.factor(A.data, A.indices, A.indptr, A.shape)
        assert np.array_equal(Fh["perm_r"], F["perm_r"])
        LU = np.empty(nL + len(F["Ux"]), dtype=np.complex128); health = np.zeros(3); out = c_vp()
        check(lib.nep_lu_factor_dev(h, hptr(np.ascontiguousarray(A.data)), 10, 1e8, hptr(health), hptr(LU), C.byref(out), stream_ptr()))
        Ld = sp.csc_matrix((LU[:nL], F["Li"], F["Lp"]), shape=(n, n)); Ud = sp.csc_matrix((LU[nL:], F["Ui"], F["Up"]), shape=(n, n))
        Lh = sp.csc_matrix((Fh["Lx"], Fh["Li"], Fh["Lp"]), shape=(n, n)); Uh = sp.csc_matrix((Fh["Ux"], Fh["Ui"], Fh["Up"]), shape=(n, n))
        assert abs(Ld - Lh).max() <= 1e-10 * abs(Lh).max() and abs(Ud - Uh).max() <= 1e-10 * abs(Uh).max()
        assert health[0] == 0 and 0 < health[1] < 1e4
        # health[2] = the element growth max|U| / max|A| (|Re| + |Im| norm) of THIS factorisation, against the host factor's
        a1 = lambda z: (abs(z.real) + abs(z.imag)).max()
        assert health[2] == pytest.approx(a1(Fh["Ux"]) / a1(A.data), rel=1e-8)
        b = np.random.default_rng(1).standard_normal(n) + 0j
        bd = torch.from_numpy(b).to("cuda"); x = torch.empty_like(bd)
        check(lib.nep_lu_solve(out, 1, c_vp(bd.data_ptr()), n, c_vp(x.data_ptr()), n, 1.0, stream_ptr()))
        assert np.linalg.norm(A @ x.cpu().numpy() - b) <= 1e-9 * np.linalg.norm(b)
        lib.nep_lu_destroy(out)
    # a growth limit below the factor's growth: refused with NEP_ERR_SINGULAR, nothing returned
    out = c_vp()
    assert lib.nep_lu_factor_dev(h, hptr(np.ascontiguousarray(A0.data)), 10, 1e-3, None, None, C.byref(out), stream_ptr()) == -3
    assert not out.value
    # ... also when only the growth in U exceeds it (limit between max |L| and max|U| / max|A|, whichever order they come in)
    hh = np.zeros(3); out = c_vp()
    check(lib.nep_lu_factor_dev(h, hptr(np.ascontiguousarray(A0.data)), 10, 1e8, hptr(hh), None, C.byref(out), stream_ptr()))
    lib.nep_lu_destroy(out)
    if abs(np.log(hh[1] / hh[2])) > 0.1:
        mid = float(np.sqrt(hh[1] * hh[2])); out = c_vp()
        assert lib.nep_lu_factor_dev(h, hptr(np.ascontiguousarray(A0.data)), 10, mid, None, None, C.byref(out), stream_ptr()) == -3
        assert not out.value
    lib.nep_lu_refac_destroy(h)
    # (ii) through DeviceLU: first matrix on the host (plan built in the background), second on the device
    lu0 = na.DeviceLU(A0)
    assert not lu0.device_factorized
    ls._DeviceRefactor.wait()
    lu1 = na.DeviceLU(A1)
    assert lu1.device_factorized and lu1.block_schedule and lu1.growth > 0
    b = np.random.default_rng(2).standard_normal(n) + 1j
    x1 = lu1.solve(torch.from_numpy(b).to("cuda")).cpu().numpy()
    assert np.linalg.norm(A1 @ x1 - b) <= 1e-9 * np.linalg.norm(b)
    monkeypatch.setattr(ls._DeviceRefactor, "GROWTH", 1e-3)
    lu2 = na.DeviceLU(A1)                                   # falls back to the host factorisation
    assert not lu2.device_factorized
    x2 = lu2.solve(torch.from_numpy(b).to("cuda")).cpu().numpy()
    assert np.linalg.norm(x2 - x1) <= 1e-9 * np.linalg.norm(x1)
    monkeypatch.setenv("NEP_LU_DEV", "0")
    assert not na.DeviceLU(A1).device_factorized
    ls._DeviceRefactor.clear()


@pytest.mark.parametrize("case", ["gun1310", "gun2600", "grid_random"])
def test_device_factorization_wide_levels_in_panels(na, monkeypatch, case):
    """csrc/lufac.hip, k_lu_widep: the chain of pivot steps at the top of the elimination tree taken P pivots per launch (every
    thread eliminates the bordered (P+1) x (P+1) matrix of its destination, the updates into the panel's own later rows and
    columns are deferred to the end of the level).  Same factor as the step-by-step form (P = 1) and as the host factorisation
    for P = 2, 3, 4 -- also on a pattern whose top is NOT dense (operands that reach a destination only through the panel's
    chain) and for a batch of matrices in one pass -- with a quarter of the launches at P = 4."""
    import ctypes as C
    import torch
    from oracle import gallery as og
    from nep_amd._lib import lib, check, hptr, c_vp
    from nep_amd.nep import stream_ptr
    import nep_amd_hostlu as hl
    rng = np.random.default_rng(5)
    if case.startswith("gun"):
        onep = og.gun_spmf_scaled(int(case[3:]))
        mats = [sp.csc_matrix(onep.compute_Mder(z)).astype(np.complex128) for z in (0.1, 0.15 + 0.05j)]
    else:
        nx = 24; n_ = nx * nx
        T = sp.diags([-1, 2.2, -1], [-1, 0, 1], shape=(nx, nx))
        K = sp.kron(sp.identity(nx), T) + sp.kron(T, sp.identity(nx))
        R = sp.random(n_, n_, density=1.0 / n_, random_state=7, format="csc")
        patt = (abs(K) + abs(R) + abs(R.T)).tocsc(); patt.sort_indices()
        mats = []
        for _ in range(2):
            A = patt.astype(np.complex128).copy()
            A.data = 0.05 * (rng.standard_normal(A.nnz) + 1j * rng.standard_normal(A.nnz))
            mats.append((A + sp.identity(n_) * (4.0 + 0.5j)).tocsc())
    for A in mats:
        A.sort_indices()
    A0, A1 = mats
    assert np.array_equal(A0.indices, A1.indices) and np.array_equal(A0.indptr, A1.indptr)
    n = A0.shape[0]
    F = hl.factor(A0.data, A0.indices, A0.indptr, A0.shape)
    F1 = hl.factor(A1.data, A1.indices, A1.indptr, A1.shape)
    ref = na.DeviceLU(factors=F)
    nL = len(F["Lx"]); nU = len(F["Ux"])
    LUs = {}; launches = {}
    for P in (1, 2, 3, 4):
        monkeypatch.setenv("NEP_LU_WIDE_P", str(P))
        h = c_vp()
        check(lib.nep_lu_refac_create(ref.h, n, hptr(F["Lp"]), hptr(F["Li"]), hptr(F["Up"]), hptr(F["Ui"]), hptr(F["perm_r"]),
                                      hptr(F["perm_c"]), hptr(np.ascontiguousarray(A0.indptr, dtype=np.int32)),
                                      hptr(np.ascontiguousarray(A0.indices, dtype=np.int32)), C.byref(h)))
        wi = (C.c_int64 * 5)(); check(lib.nep_lu_refac_wide_info(h, wi))
        assert wi[0] == P and wi[1] > 0
        launches[P] = wi[2]
        if P > 1:
            assert wi[3] > 0 and wi[2] <= (wi[1] + P - 1) // P + 16 * (P - 1)
        # both matrices in one pass (grid.y = matrix), values read back
        Ax = np.ascontiguousarray(np.stack([A0.data, A1.data]))
        LU = np.empty((2, nL + nU), dtype=np.complex128); health = np.zeros((2, 3)); outs = (c_vp * 2)()
        check(lib.nep_lu_factor_dev_batch(h, 2, hptr(Ax), 10, 1e8, hptr(health), hptr(LU), outs, stream_ptr()))
        assert outs[0] and outs[1] and health[0, 0] == 0 and health[1, 0] == 0
        LUs[P] = LU.copy()
        for b_, (A, Fh) in enumerate(((A0, F), (A1, F1))):
            if np.array_equal(Fh["perm_r"], F["perm_r"]) and np.array_equal(Fh["perm_c"], F["perm_c"]) and np.array_equal(Fh["Lp"], F["Lp"]):
                # (copies: scipy sorts the index arrays it was given IN PLACE on the first subtraction, and the plan of the next
                # P is created from F's arrays)
                mk = lambda x, i_, p_: sp.csc_matrix((np.array(x), np.array(i_), np.array(p_)), shape=(n, n))
                Ld = mk(LU[b_, :nL], F["Li"], F["Lp"]); Ud = mk(LU[b_, nL:], F["Ui"], F["Up"])
                Lh = mk(Fh["Lx"], Fh["Li"], Fh["Lp"]); Uh = mk(Fh["Ux"], Fh["Ui"], Fh["Up"])
                assert abs(Ld - Lh).max() <= 1e-10 * abs(Lh).max() and abs(Ud - Uh).max() <= 1e-10 * abs(Uh).max()
            rhs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            bd = torch.from_numpy(rhs).to("cuda"); x = torch.empty_like(bd)
            check(lib.nep_lu_solve(outs[b_], 1, c_vp(bd.data_ptr()), n, c_vp(x.data_ptr()), n, 1.0, stream_ptr()))
            assert np.linalg.norm(A @ x.cpu().numpy() - rhs) <= 1e-9 * np.linalg.norm(rhs)
            lib.nep_lu_destroy(outs[b_])
        # the same plan, the same matrix: the same bits (record order comes from an atomic counter; the values must not)
        LU2 = np.empty(nL + nU, dtype=np.complex128); out = c_vp()
        check(lib.nep_lu_factor_dev(h, hptr(np.ascontiguousarray(A0.data)), 10, 1e8, None, hptr(LU2), C.byref(out), stream_ptr()))
        assert np.array_equal(LU2.view(np.float64), LUs[P][0].view(np.float64))
        lib.nep_lu_destroy(out)
        lib.nep_lu_refac_destroy(h)
    for P in (2, 3, 4):
        assert np.abs(LUs[P] - LUs[1]).max() <= 1e-10 * np.abs(LUs[1]).max()
    assert launches[4] < launches[2] < launches[1] and launches[4] <= launches[1] // 3 + 8


def test_linear_solver_from_terms_and_pattern_digest_memo(na, monkeypatch):
    """linsolvers.FactorizeLinSolver._lu_from_terms: with a ready plan the matrix of a pure SPMF NEP is assembled and factorised on
    the device from the m_t coefficients (no compute_Mder, no value upload) -- same solves as the general route, which
    NEP_LU_TERMS=0 restores; _DeviceRefactor.key does not hash index arrays it has hashed before (same memory, kept alive)
    and still tells patterns apart."""
    import torch
    from nep_amd import linsolvers as ls
    ls._DeviceRefactor.clear()
    nep = na.nep_gallery("gun_spmf_scaled", 1310)
    lam0, lam1 = 0.0, 0.12 + 0.03j
    s0 = ls.FactorizeLinSolver(nep, lam0)                       # host factorisation; starts the plan
    assert not s0.lu.device_factorized
    ls._DeviceRefactor.wait()
    calls = {"n": 0}
    orig = type(nep).compute_Mder

    def counting(self, lam, i=0):
        calls["n"] += 1
        return orig(self, lam, i)
    monkeypatch.setattr(type(nep), "compute_Mder", counting)
    # (patched: the terms route must refuse a NEP whose compute_Mder is not the SPMF one)
    s_gen = ls.FactorizeLinSolver(nep, lam1)
    assert calls["n"] == 1 and s_gen.lu.device_factorized
    monkeypatch.undo()
    s_t = ls.FactorizeLinSolver(nep, lam1)
    assert s_t.lu.device_factorized and s_t.lu.normA == pytest.approx(s_gen.lu.normA, rel=1e-12)
    b = np.random.default_rng(3).standard_normal(nep.n) + 1j * np.random.default_rng(4).standard_normal(nep.n)
    bd = torch.from_numpy(b).to("cuda")
    x_t = ls.lin_solve(s_t, bd).cpu().numpy().ravel(); x_g = ls.lin_solve(s_gen, bd).cpu().numpy().ravel()
    A1 = nep.compute_Mder(lam1)
    assert np.linalg.norm(A1 @ x_t - b) <= 1e-10 * np.linalg.norm(b)
    assert np.linalg.norm(x_t - x_g) <= 1e-9 * np.linalg.norm(x_g)
    monkeypatch.setenv("NEP_LU_TERMS", "0")
    assert ls.FactorizeLinSolver._lu_from_terms(nep, lam1, None, {"expected_solves": 200}) is None
    monkeypatch.delenv("NEP_LU_TERMS")
    assert ls.FactorizeLinSolver._lu_from_terms(nep, lam1, "COLAMD", {"expected_solves": 200}) is None
    # digest memo: two matrices of the NEP share the index arrays -> one hash; equal to a fresh hash; a different pattern differs
    Aa = sp.csc_matrix(nep.compute_Mder(lam0), dtype=np.complex128); Ab = sp.csc_matrix(nep.compute_Mder(lam1), dtype=np.complex128)
    ka = ls._DeviceRefactor.key(Aa, (None, None, None)); kb = ls._DeviceRefactor.key(Ab, (None, None, None))
    del ls._DeviceRefactor._digests[:]
    Ac = Aa.copy()
    assert ka == kb == ls._DeviceRefactor.key(Ac, (None, None, None))
    Ad = Ac.copy(); Ad.indices = Ad.indices.copy(); Ad.indices[0], Ad.indices[1] = Ad.indices[1], Ad.indices[0]
    assert ls._DeviceRefactor.key(Ad, (None, None, None)) != ka
    assert ls._DeviceRefactor.key(Ac, (None, None, None)) == ka and len(ls._DeviceRefactor._digests) <= 8
    ls._DeviceRefactor.clear()


def test_compute_types(na):
    """test/compute_types.jl, the precisions NumPy and the device share: host results of a REAL NEP are float64 for real
    lambda / V / S and complex128 otherwise; a complex NEP (or one with a function that leaves the reals, gun's i*sqrt)
    always returns complex128.  Values are those of the complex computation."""
    from oracle import neps as on
    rng = np.random.default_rng(0)
    n = 5
    Ar = [rng.standard_normal((n, n)) for _ in range(3)]
    cases = [(na.PEP(Ar), True), (na.PEP([A + 1j * np.eye(n) for A in Ar]), False), (na.DEP([Ar[0], Ar[1]], [0.0, 0.7]), True),
             (na.SPMF_NEP([Ar[0], Ar[1]], [na.funcs.one(), na.funcs.ISqrt(1.0, 0.0)]), False)]
    for nep, real in cases:
        assert nep.is_real() == real
        for lam in (1.0, 1.0 + 1.0j):
            M = nep.compute_Mder(lam)
            dt = M.dtype if hasattr(M, "dtype") else np.asarray(M).dtype
            assert dt == on.result_type(real, lam)
            for V in (np.ones((n, 3)), np.ones((n, 3)) + 0j):
                y = nep.compute_Mlincomb(lam, V)
                assert y.dtype == on.result_type(real, lam, V)
                yc = nep.compute_Mlincomb(complex(lam), V + 0j)
                assert np.linalg.norm(y - yc) <= 1e-13 * max(1.0, np.linalg.norm(yc))
                assert nep.compute_Mlincomb(lam, V[:, 0]).dtype == on.result_type(real, lam, V)
        for S in (np.eye(2), np.eye(2) + 0j):
            for V in (np.ones((n, 2)), np.ones((n, 2)) + 0j):
                assert nep.compute_MM(S, V).dtype == on.result_type(real, S, V)


@pytest.mark.parametrize("case", ["gun", "wep", "qdep0", "random_complex"])
def test_mlincomb_tiled_one_launch_kernel(na, case):
    """csrc/spmv_tile.hip: compute_Mlincomb as ONE launch on footprint tiles (grid patches for the gun / waveguide stencils,
    consecutive-row blocks otherwise) against the two-launch / folded kernels (nep_k1_set_mode 2) and host NumPy; every k
    regime of the kernel: k = 1, small k (one thread group per footprint column), large k (column groups + LDS reduction);
    V not modified"""
    import torch
    from nep_amd._lib import lib, check
    from nep_amd import gallery, wep
    rng = np.random.default_rng(11)
    if case == "gun":
        K, M, W1, W2 = gallery.gun_matrices(); Av = [K, -M, W1, W2]
    elif case == "wep":
        Av = wep.WaveguideData(303, 299, "JARLEBRING").big_matrices()          # n = 91 195: SELL path, non-temporal entry loads
    elif case == "qdep0":
        Av = na.nep_gallery("qdep0").get_Av()
    else:
        n0 = 3001
        Av = [sp.random(n0, n0, density=0.003, random_state=1, format="csr") + sp.identity(n0, format="csr"),
              sp.random(n0, n0, density=0.002, random_state=2, format="csr") * (1 + 2j),
              sp.random(n0, n0, density=0.001, random_state=3, format="csr")]
    dev = na.SPMFDevice(Av)
    ti = dev.tile_info()
    assert ti["blocks"] > 0 and ti["max_footprint"] > 0
    if case in ("gun", "wep"):
        assert ti["stride"] == (131 if case == "gun" else 299)               # the grid line length, found from the pattern
    n, mt = dev.n, dev.mt
    try:
        for k in (1, 2, 7, 8, 33, 100):
            V = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
            Cm = rng.standard_normal((k, mt)) + 1j * rng.standard_normal((k, mt))
            Vd = torch.from_numpy(V).to("cuda"); V0 = Vd.clone()
            ref = sum(Av[t] @ (V.T @ Cm[:, t]) for t in range(mt))
            zs = {}
            for mode in (1, 2):
                check(lib.nep_k1_set_mode(mode))
                zs[mode] = dev.mlincomb(Cm, Vd).cpu().numpy()
                zd = torch.empty(n, dtype=torch.complex128, device="cuda")
                dev.mlincomb_dev(na.to_dev(Cm), k, k, Vd, n, zd)
                assert np.array_equal(zd.cpu().numpy(), zs[mode])                # host- and device-coefficient entry points agree
            assert torch.equal(Vd, V0)
            scale = np.linalg.norm(ref)
            assert np.linalg.norm(zs[1] - ref) <= 1e-12 * scale and np.linalg.norm(zs[2] - ref) <= 1e-12 * scale
    finally:
        check(lib.nep_k1_set_mode(0))


def test_iar_same_result_with_tiled_k1(na):
    """iar's native step folds the block shift of the basis column into the K1 kernel: with the tiled kernel forced on for
    every k (mode 1) and forced off (mode 2) the gun twin returns the same eigenpairs and error history"""
    from nep_amd._lib import lib, check
    nep = na.nep_gallery("gun_spmf_scaled", 1310); n = nep.n
    out = {}
    try:
        for mode in (1, 2):
            check(lib.nep_k1_set_mode(mode))
            h = []
            lam, Q, _ = na.iar(nep, sigma=0.0, gamma=1.0, maxit=40, neigs=np.inf, v=np.ones(n), tol=1e-10, errhist=h)
            out[mode] = (lam, h)
    finally:
        check(lib.nep_k1_set_mode(0))
    (l1, h1), (l2, h2) = out[1], out[2]
    assert len(l1) == len(l2) >= 1 and len(h1) == len(h2)
    assert max(np.min(abs(l2 - x)) / max(1.0, abs(x)) for x in l1) < 1e-11


@pytest.mark.parametrize("case", ["wep", "gun", "random_complex"])
def test_resid_batch_tiled_kernel(na, case):
    """K2 on the footprint tiles (k_tile_resid: Q rows of a block's footprint staged in LDS per column panel, thread per row)
    against the wave-per-row kernel (mode 2) and NumPy: column norms of the residual block and of Q (nep_resid_batch_dev), and
    the residual block itself (nep_resid_block); k below, at and above the panel widths, ldq > k"""
    import ctypes as C
    import torch
    from nep_amd._lib import lib, check, hptr, c_vp
    from nep_amd import gallery, wep
    rng = np.random.default_rng(12)
    if case == "gun":
        K, M, W1, W2 = gallery.gun_matrices(); Av = [K, -M, W1, W2]
    elif case == "wep":
        Av = wep.WaveguideData(303, 299, "JARLEBRING").big_matrices()
    else:
        n0 = 3001
        Av = [sp.random(n0, n0, density=0.003, random_state=1, format="csr") + sp.identity(n0, format="csr"),
              sp.random(n0, n0, density=0.002, random_state=2, format="csr") * (1 + 2j)]
    dev = na.SPMFDevice(Av)
    n, mt = dev.n, dev.mt
    try:
        for k, ldq in ((1, 1), (3, 5), (8, 8), (13, 16), (60, 60), (130, 130)):
            Q = rng.standard_normal((n, ldq)) + 1j * rng.standard_normal((n, ldq))
            F = np.asfortranarray(rng.standard_normal((mt, k)) + 1j * rng.standard_normal((mt, k)))
            R = np.column_stack([sum(F[t, s] * (Av[t] @ Q[:, s]) for t in range(mt)) for s in range(k)])
            Qd = torch.from_numpy(Q).to("cuda")
            res = {}
            for mode in (1, 2):
                check(lib.nep_k1_set_mode(mode))
                o = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                check(lib.nep_resid_batch_dev(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, c_vp(o.data_ptr()), None))
                RT = torch.zeros((n, k), dtype=torch.complex128, device="cuda")
                check(lib.nep_resid_block(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, c_vp(RT.data_ptr()), k, None))
                res[mode] = (o.cpu().numpy(), RT.cpu().numpy())
            for mode in (1, 2):
                o, RT = res[mode]
                assert np.allclose(o[:k], np.sum(abs(R) ** 2, axis=0), rtol=1e-12)
                assert np.allclose(o[k:], np.sum(abs(Q[:, :k]) ** 2, axis=0), rtol=1e-12)
                assert np.linalg.norm(RT - R) <= 1e-13 * np.linalg.norm(R)
            # deterministic: a second launch returns the same bits
            check(lib.nep_k1_set_mode(1))
            o2 = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
            check(lib.nep_resid_batch_dev(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, c_vp(o2.data_ptr()), None))
            assert np.array_equal(o2.cpu().numpy(), res[1][0])
    finally:
        check(lib.nep_k1_set_mode(0))


def test_lu_fused_last_launch_and_apex_gemv(na, monkeypatch):
    """K5 single-vector solve: the last launch with U level 0's coupling product inside (k_ml_u0_fused: paired rows of the packed
    inverse, two workgroups per block) and the two-rows-per-wave apex product (k_apex_gemv1) against the separate launches
    (NEP_ML_U0FUSE=0, NEP_ML_GEMV1=0) and SciPy; with and without the fused refinement update x + A^{-1} r and UMFPACK's row
    scaling; one launch fewer per solve"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    from oracle import gallery as og
    A = sp.csc_matrix(og.gun_spmf_scaled(9956).compute_Mder(0.0)).astype(np.complex128)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xs = spla.splu(A).solve(b)
    lu = na.DeviceLU(A, expected_solves=100)
    bd = torch.from_numpy(b).to("cuda")
    torch.cuda.synchronize()
    import time
    time.sleep(0.05)                                   # the dense apex is built behind the first solves
    res = {}
    for fuse, gemv in (("0", "0"), ("1", "1")):
        monkeypatch.setenv("NEP_ML_U0FUSE", fuse); monkeypatch.setenv("NEP_ML_GEMV1", gemv)
        for _ in range(3):
            x = lu.solve(bd).cpu().numpy()
        res[fuse] = (x, lu.launches_last_solve())
        assert np.linalg.norm(x - xs) <= 1e-9 * np.linalg.norm(xs)
    assert np.linalg.norm(res["1"][0] - res["0"][0]) <= 1e-12 * np.linalg.norm(xs)
    assert res["1"][1] == res["0"][1] - 1                # (the fused form is opt-in: measured no faster)
    # through FactorizeLinSolver: refinement update fused into the last launch (nep_lu_solve_add), omega at round-off
    nep = na.nep_gallery("gun_spmf_scaled")
    ls = na.create_linsolver(na.FactorizeLinSolverCreator(), nep, 0.0)
    xr = na.lin_solve(ls, b)
    assert ls.last_omega < 10 * np.finfo(float).eps
    assert np.linalg.norm(A @ xr - b) <= 1e-13 * np.linalg.norm(b) * np.sqrt(n)


@pytest.mark.parametrize("m,n,k", [(130, 70, 33), (64, 64, 16), (1, 1, 1), (200, 3, 129)])
def test_own_gemm_tile_edges_and_real(na, m, n, k):
    """k_gemm_general at tile edges (sizes that are, straddle, or fall short of the 64 x 64 x 16 tiles), beta = 0 with NaN in
    C (C must not be read), and the float64 instantiation behind nep_dgemm"""
    import torch
    from nep_amd._lib import lib, check, c_vp, cd
    from nep_amd.nep import stream_ptr
    rng = np.random.default_rng(m + 7 * n + 13 * k)
    A = rng.standard_normal((k, m)) + 1j * rng.standard_normal((k, m))           # used as A^H
    B = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
    Ad, Bd = na.to_dev(A), na.to_dev(B)                                          # column-major k x m, k x n
    Cd = torch.full((n, m), float("nan"), dtype=torch.complex128, device="cuda")
    check(lib.nep_zgemm(2, 0, m, n, k, cd(1.0), c_vp(Ad.data_ptr()), k, c_vp(Bd.data_ptr()), k, cd(0.0), c_vp(Cd.data_ptr()), m, stream_ptr()))
    ref = A.conj().T @ B
    assert np.linalg.norm(Cd.cpu().numpy().T - ref) <= 1e-13 * np.linalg.norm(ref)
    Ar = rng.standard_normal((m, k)); Br = rng.standard_normal((n, k)); C0 = rng.standard_normal((m, n))
    Ard = torch.from_numpy(np.asfortranarray(Ar).T.copy()).to("cuda")            # column-major m x k
    Brd = torch.from_numpy(np.asfortranarray(Br).T.copy()).to("cuda")            # column-major n x k, used transposed
    Crd = torch.from_numpy(np.asfortranarray(C0).T.copy()).to("cuda")
    check(lib.nep_dgemm(0, 1, m, n, k, 0.5, c_vp(Ard.data_ptr()), m, c_vp(Brd.data_ptr()), n, -2.0, c_vp(Crd.data_ptr()), m, stream_ptr()))
    refr = 0.5 * Ar @ Br.T - 2.0 * C0
    assert np.linalg.norm(Crd.cpu().numpy().T - refr) <= 1e-13 * np.linalg.norm(refr)


@pytest.mark.parametrize("case", ["wep", "gun"])
def test_resid_batch_column_major_tiled(na, case):
    """nep_resid_batch_cm_dev (k_tile_resid_cm: K2 on the footprint tiles with a COLUMN-major Ritz block, entries of a thread's
    row held in registers across the column panels): squared norms against NumPy for k below / at / above the panel width, and
    the split form (rows below row0 in the norms, the tail rows written column-major)"""
    import torch
    from nep_amd._lib import lib, check, hptr, c_vp
    from nep_amd import gallery, wep
    rng = np.random.default_rng(21)
    if case == "gun":
        K, M, W1, W2 = gallery.gun_matrices(); Av = [K, -M, W1, W2]
    else:
        Av = wep.WaveguideData(303, 299, "JARLEBRING").big_matrices()
    dev = na.SPMFDevice(Av)
    n, mt = dev.n, dev.mt
    for k in (1, 3, 4, 9, 60):
        Q = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
        F = np.asfortranarray(rng.standard_normal((mt, k)) + 1j * rng.standard_normal((mt, k)))
        R = np.column_stack([sum(F[t, s] * (Av[t] @ Q[:, s]) for t in range(mt)) for s in range(k)])
        Qc = torch.from_numpy(np.ascontiguousarray(Q.T)).to("cuda")              # (k, n) = column-major n x k
        o = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
        check(lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, -1, c_vp(o.data_ptr()), None, 0, None))
        oh = o.cpu().numpy()
        assert np.allclose(oh[:k], np.sum(abs(R) ** 2, axis=0), rtol=1e-12)
        assert np.allclose(oh[k:], np.sum(abs(Q) ** 2, axis=0), rtol=1e-12)
        row0 = n - 37
        o2 = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
        tail = torch.full((k, n - row0), float("nan"), dtype=torch.complex128, device="cuda")
        check(lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, row0, c_vp(o2.data_ptr()), c_vp(tail.data_ptr()),
                                         n - row0, None))
        o2h = o2.cpu().numpy()
        assert np.allclose(o2h[:k], np.sum(abs(R[:row0]) ** 2, axis=0), rtol=1e-12)
        assert np.allclose(o2h[k:], np.sum(abs(Q) ** 2, axis=0), rtol=1e-12)
        assert np.linalg.norm(tail.cpu().numpy().T - R[row0:]) <= 1e-13 * np.linalg.norm(R[row0:])
        check(lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, -1, c_vp(o2.data_ptr()), None, 0, None))
        assert np.array_equal(o2.cpu().numpy(), oh)                              # deterministic


@pytest.mark.parametrize("n", [1, 5, 64, 257, 1000])
def test_dense_inverse_gauss_jordan_on_the_device(na, n):
    """nep_zinv_h_dev: inv(M + I)^H by in-place Gauss-Jordan with partial pivoting (the Sylvester-SMW matrix of the waveguide
    preconditioner) against numpy.linalg.inv: a matrix that NEEDS the pivoting (tiny diagonal), leading dimensions larger than n, the
    singular case reported through info, bitwise repeatable"""
    import ctypes as C
    import torch
    from nep_amd._lib import lib, check, c_vp
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    M[np.arange(n), np.arange(n)] = 1e-9 - 1.0              # M + I has a tiny diagonal: unpivoted elimination would lose everything
    ld = n + 3
    Md = torch.zeros((n, ld), dtype=torch.complex128, device="cuda")          # column-major, ld > n
    Md[:, :n] = torch.from_numpy(np.ascontiguousarray(M.T)).to("cuda")
    outs = []
    for _ in range(2):
        out = torch.zeros((n, ld), dtype=torch.complex128, device="cuda")
        work = torch.empty(2 * n + 2, dtype=torch.complex128, device="cuda")
        info = C.c_int32(-1)
        check(lib.nep_zinv_h_dev(n, c_vp(Md.data_ptr()), ld, 1.0, c_vp(out.data_ptr()), ld, c_vp(work.data_ptr()), C.byref(info), None))
        assert info.value == 0
        outs.append(out.cpu().numpy()[:, :n].T)             # out[c, r] = X[r, c]
    ref = np.linalg.inv(M + np.eye(n)).conj().T
    assert np.linalg.norm(outs[0] - ref) <= 1e-10 * np.linalg.norm(ref) * max(1.0, np.linalg.cond(M + np.eye(n)) * 1e-3)
    assert np.array_equal(outs[0], outs[1])
    if n >= 5:                                               # singular: M + I with two equal rows, and with a zero column
        for kind in ("rows", "column"):
            A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            if kind == "rows":
                A[3] = A[1]
            else:
                A[:, 2] = 0.0
            S = A - np.eye(n)
            Sd = torch.from_numpy(np.ascontiguousarray(S.T)).to("cuda").contiguous()
            out = torch.zeros((n, n), dtype=torch.complex128, device="cuda")
            work = torch.empty(2 * n + 2, dtype=torch.complex128, device="cuda")
            info = C.c_int32(0)
            check(lib.nep_zinv_h_dev(n, c_vp(Sd.data_ptr()), n, 1.0, c_vp(out.data_ptr()), n, c_vp(work.data_ptr()), C.byref(info), None))
            oh = out.cpu().numpy()
            if kind == "column":                                 # (a zero column of M + I = a zero ROW of the adjoint that is inverted:
                assert info.value == n                           #  the search avoids it until the last step)
            else:                                                # rounding leaves a pivot of ~1e-16 instead of 0: reported or visible
                assert info.value != 0 or not np.isfinite(oh).all() or np.abs(oh).max() > 1e8


@pytest.mark.parametrize("case", ["wep", "gun", "wep_small_patch", "wep_tall_patch"])
def test_resid_batch_super_panel_kernel(na, case, monkeypatch):
    """K2 in super-panels (k_tile_resid_sp: one workgroup per block, the row's entries in registers, 4-column footprint tiles filled by
    LDS-DMA, double-buffered) against NumPy and against the older kernels, for BOTH layouts of the Ritz block: row-major
    (nep_resid_batch_dev, nep_resid_block, nep_resid_split_dev) and column-major (nep_resid_batch_cm_dev, with and without a tail
    block); k below / at / above the panel width and odd; ldq > k; bitwise repeatable"""
    import torch
    from nep_amd._lib import lib, check, hptr, c_vp
    from nep_amd import gallery, wep
    rng = np.random.default_rng(33)
    if case == "gun":
        K, M, W1, W2 = gallery.gun_matrices(); Av = [K, -M, W1, W2]
    elif case == "wep":
        Av = wep.WaveguideData(303, 299, "JARLEBRING").big_matrices()
    elif case == "wep_tall_patch":
        # a 14 x 64 patch: footprint 16 x 66 = 1056 slots, inside the tile pitch of a 1024-thread workgroup (1152) but beyond what
        # the 16-bit byte offset of the ROW-major tile can address (slot >= 1024): row-major blocks must take the older kernels,
        # column-major ones stay on the super-panel kernel (silent wrong residuals before)
        monkeypatch.setenv("NEP_K1_TILE_XP", "14"); monkeypatch.setenv("NEP_K1_TILE_ZP", "64")
        Av = wep.WaveguideData(303, 299, "JARLEBRING").big_matrices()
    else:
        Av = wep.WaveguideData(61, 37, "JARLEBRING").big_matrices()          # blocks cut by the grid's edge, short footprints
    dev = na.SPMFDevice(Av)
    n, mt = dev.n, dev.mt
    if case == "wep_tall_patch":
        ti = dev.tile_info()
        assert ti["blocks"] > 0 and 1024 < ti["max_footprint"] <= 1152, ti
    try:
        for k, ldq in ((1, 1), (3, 5), (4, 4), (7, 8), (8, 8), (13, 16), (60, 60), (61, 64)):
            Q = rng.standard_normal((n, ldq)) + 1j * rng.standard_normal((n, ldq))
            F = np.asfortranarray(rng.standard_normal((mt, k)) + 1j * rng.standard_normal((mt, k)))
            R = np.column_stack([sum(F[t, s] * (Av[t] @ Q[:, s]) for t in range(mt)) for s in range(k)])
            r2 = np.sum(abs(R) ** 2, axis=0); q2 = np.sum(abs(Q[:, :k]) ** 2, axis=0)
            Qd = torch.from_numpy(Q).to("cuda")
            Qc = torch.from_numpy(np.ascontiguousarray(Q[:, :k].T)).to("cuda")          # (k, n) = column-major n x k
            row0 = n - 37
            got = {}
            for mode in (0, 2):
                check(lib.nep_k2_set_sp_mode(mode))
                o = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                check(lib.nep_resid_batch_dev(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, c_vp(o.data_ptr()), None))
                RT = torch.zeros((n, k), dtype=torch.complex128, device="cuda")
                check(lib.nep_resid_block(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, c_vp(RT.data_ptr()), k, None))
                os_ = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                tl = torch.full((n - row0, k), float("nan"), dtype=torch.complex128, device="cuda")
                check(lib.nep_resid_split_dev(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, row0, c_vp(os_.data_ptr()), c_vp(tl.data_ptr()), k, None))
                oc = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                rc = lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, -1, c_vp(oc.data_ptr()), None, 0, None)
                oc2 = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                tc = torch.full((k, n - row0), float("nan"), dtype=torch.complex128, device="cuda")
                rc2 = lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, row0, c_vp(oc2.data_ptr()), c_vp(tc.data_ptr()), n - row0, None)
                assert rc == 0 and rc2 == 0
                got[mode] = [x.cpu().numpy() for x in (o, RT, os_, tl, oc, oc2, tc)]
            for mode in (0, 2):
                o, RT, os_, tl, oc, oc2, tc = got[mode]
                assert np.allclose(o[:k], r2, rtol=1e-12) and np.allclose(o[k:], q2, rtol=1e-12), (mode, k)
                assert np.linalg.norm(RT - R) <= 1e-13 * np.linalg.norm(R)
                assert np.allclose(os_[:k], np.sum(abs(R[:row0]) ** 2, axis=0), rtol=1e-12) and np.allclose(os_[k:], q2, rtol=1e-12)
                assert np.linalg.norm(tl - R[row0:]) <= 1e-13 * np.linalg.norm(R[row0:])
                assert np.allclose(oc[:k], r2, rtol=1e-12) and np.allclose(oc[k:], q2, rtol=1e-12)
                assert np.allclose(oc2[:k], np.sum(abs(R[:row0]) ** 2, axis=0), rtol=1e-12) and np.allclose(oc2[k:], q2, rtol=1e-12)
                assert np.linalg.norm(tc.T - R[row0:]) <= 1e-13 * np.linalg.norm(R[row0:])
            # the ring-of-four-half-tiles variant of the column-major kernel (opt-in): the same bits as the two-tile kernel
            if k >= 9:
                monkeypatch.setenv("NEP_K2_SP_RING", "4")
                check(lib.nep_k2_set_sp_mode(2))
                oc4 = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                check(lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, -1, c_vp(oc4.data_ptr()), None, 0, None))
                oc5 = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
                tc5 = torch.full((k, n - row0), float("nan"), dtype=torch.complex128, device="cuda")
                check(lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, row0, c_vp(oc5.data_ptr()), c_vp(tc5.data_ptr()), n - row0, None))
                monkeypatch.delenv("NEP_K2_SP_RING")
                assert np.array_equal(oc4.cpu().numpy(), got[2][4]) and np.array_equal(oc5.cpu().numpy(), got[2][5])
                assert np.array_equal(tc5.cpu().numpy(), got[2][6])
            # the super-panel form again: the same bits
            check(lib.nep_k2_set_sp_mode(2))
            o = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
            check(lib.nep_resid_batch_dev(dev.h, k, hptr(F), c_vp(Qd.data_ptr()), ldq, c_vp(o.data_ptr()), None))
            assert np.array_equal(o.cpu().numpy(), got[2][0])
            oc = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
            check(lib.nep_resid_batch_cm_dev(dev.h, k, hptr(F), c_vp(Qc.data_ptr()), n, -1, c_vp(oc.data_ptr()), None, 0, None))
            assert np.array_equal(oc.cpu().numpy(), got[2][4])
    finally:
        check(lib.nep_k2_set_sp_mode(-1))


@pytest.mark.parametrize("k", [3, 16, 17, 40, 70, 104, 128, 130])
@pytest.mark.parametrize("reorth", [False, True, "borderline"])
def test_orth_dev_decision_published_by_the_update(na, k, reorth):
    """the asynchronous DGKS (nep_orth_dev): the decision of a pass is published by that pass' own update kernel before the
    update runs, from ||w||^2 and ||c||^2 (Pythagoras; csrc/orth.hip k_orth_update) -- against the oracle's DGKS, which takes the
    norm of the updated vector (IterativeSolvers 0.9.2): h, beta, w, pass count and flags, with and without a forced second
    pass, a vector placed 2 % on either side of the criterion's threshold, iar's block-triangular basis (`active`), rows that are
    no multiple of the 64-row tile"""
    import torch
    from oracle import solvers as osol
    rng = np.random.default_rng(100 + k)
    n = 37
    rows = n * (k + 1) + 5
    V = np.zeros((rows, k), dtype=complex)
    for j in range(k):
        V[:(j + 1) * n, j] = rng.standard_normal((j + 1) * n) + 1j * rng.standard_normal((j + 1) * n)
    V, _ = np.linalg.qr(V)
    active = (np.arange(1, k + 1) * n).astype(np.int64)
    if reorth == "borderline":
        # ||w_perp|| / ||c|| = 0.98 / sqrt(2) (re-orthogonalise) resp. 1.02 / sqrt(2) (do not): both sides of the threshold
        for fac, want in ((0.98, 2), (1.02, 1)):
            c = rng.standard_normal(k) + 1j * rng.standard_normal(k)
            z = rng.standard_normal(rows) + 1j * rng.standard_normal(rows)
            z -= V @ (V.conj().T @ z)
            wb = V @ c + z * (fac * np.linalg.norm(c) / np.sqrt(2) / np.linalg.norm(z))
            wo = wb.copy(); ho = np.zeros(k, dtype=complex)
            osol.dgks(V, wo, ho)
            outb = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
            wdb = na.to_dev(wb)[0]
            na.dense.orthogonalize_and_normalize_dev(na.to_dev(V), wdb, k, outb, rows=rows, ldv=rows, active_dev=torch.from_numpy(active).to("cuda"))
            ob = outb.cpu().numpy()
            assert int(ob[k + 1].real) == want and int(ob[k + 1].imag) == 0
            assert np.linalg.norm(ob[:k] - ho) <= 1e-12 * np.linalg.norm(ho)
        return
    if reorth:
        w = V @ (rng.standard_normal(k) + 1j * rng.standard_normal(k)) + 1e-9 * (rng.standard_normal(rows) + 1j * rng.standard_normal(rows))
    else:
        w = rng.standard_normal(rows) + 1j * rng.standard_normal(rows)
    wo = w.copy(); ho = np.zeros(k, dtype=complex)
    bo = osol.dgks(V, wo, ho)
    Vd = na.to_dev(V); act_d = torch.from_numpy(active).to("cuda")
    out = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
    wd = na.to_dev(w)[0]
    na.dense.orthogonalize_and_normalize_dev(Vd, wd, k, out, rows=rows, ldv=rows, active_dev=act_d)
    o = out.cpu().numpy()
    h, beta, passes, flags = o[:k], o[k].real, int(o[k + 1].real), int(o[k + 1].imag)
    assert passes == (2 if reorth else 1) and flags == 0
    assert beta == pytest.approx(bo, rel=1e-6 if reorth else 1e-12)
    assert np.linalg.norm(h - ho) <= 1e-12 * np.linalg.norm(ho)
    wh = wd.cpu().numpy()
    assert np.linalg.norm(V.conj().T @ wh) < 1e-12 and abs(np.linalg.norm(wh) - 1.0) < 1e-12
    if not reorth:
        assert np.linalg.norm(wh - wo) <= 1e-11
    # deterministic
    out2 = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
    wd2 = na.to_dev(w)[0]
    na.dense.orthogonalize_and_normalize_dev(Vd, wd2, k, out2, rows=rows, ldv=rows, active_dev=act_d)
    assert torch.equal(out, out2) and torch.equal(wd, wd2)


def _hess_eig_check(na, H, eig_tol, res_tol):
    import torch
    from nep_amd import dense
    k = H.shape[0]
    Hd = torch.from_numpy(np.ascontiguousarray(H.T)).to("cuda")            # (k, k) tensor = column-major H
    w, Z = dense.hess_eig_dev(Hd, k)
    wh = w.cpu().numpy(); Zh = Z.cpu().numpy().T
    assert wh[k].real == 0 and wh[k + 1].real == 0, (wh[k], wh[k + 1])       # QR converged, every inverse iteration grew
    lam = wh[:k]
    ref = np.linalg.eigvals(H)
    used = np.zeros(k, bool)
    for x in lam:                                                            # eigenvalues as multisets
        d = np.abs(ref - x); d[used] = np.inf; j = int(np.argmin(d)); used[j] = True
        assert d[j] <= eig_tol * max(np.abs(ref).max(), 1e-300), (x, ref[j])
    nH = max(np.linalg.norm(H), 1e-300)
    res = np.linalg.norm(H @ Zh - Zh * lam[None, :], axis=0) / nH
    assert res.max() <= res_tol, res.max()
    assert np.abs(np.linalg.norm(Zh, axis=0) - 1).max() < 1e-14             # zgeev's normalisation: unit 2-norm,
    big = Zh[np.argmax(np.abs(Zh), axis=0), np.arange(k)]                    # largest component real positive
    assert np.abs(big.imag).max() < 1e-14 and big.real.min() > 0
    return lam, Zh


@pytest.mark.parametrize("k", [1, 2, 3, 17, 50, 64, 65, 100])
def test_hess_eig_dev_gun_arnoldi_matrix(na, k):
    """nep_hess_eigvals_dev / nep_hess_eigvecs_dev (csrc/hesseig.hip) replace `D,Z = eigen(H[1:k,1:k])` of
    src/method_iar.jl:112: on the Hessenberg matrix of the headline gun run (fixture written by the oracle's iar, ||H|| = 3.5e6)
    the eigenvalues equal LAPACK's as multisets to 1e-12 of the spectral radius and every pair has ||H z - w z|| <= 1e-12 ||H||
    (measured 1e-13 / 3e-14; the host route zhseqr + zhsein this replaces: 8e-10 on the same matrix)."""
    import os
    H = np.load(os.path.join(os.path.dirname(__file__), "golden", "gun_iar_H100.npy"))[:k, :k]
    _hess_eig_check(na, H, 1e-12, 1e-12)


@pytest.mark.parametrize("case", ["random", "real", "triangular", "blocks", "defective", "graded", "zero", "k128"])
def test_hess_eig_dev_edge_matrices(na, case):
    """shapes the QR iteration and the inverse iteration have to survive: random complex, real entries (conjugate pairs), an
    already triangular matrix (no sweep at all), zero subdiagonal entries (decoupled blocks), a Jordan-like block (coincident
    eigenvalues: zhsein's perturbation), entries from 1e-12 to 1e12, the zero matrix"""
    rng = np.random.default_rng(7)
    k = 40
    A = np.triu(rng.standard_normal((k, k)) + 1j * rng.standard_normal((k, k)), -1)
    eig_tol, res_tol = 1e-12, 1e-13
    if case == "real":
        A = np.triu(rng.standard_normal((k, k)), -1).astype(complex)
    elif case == "triangular":
        A = np.triu(A)
    elif case == "blocks":
        A[10, 9] = 0; A[25, 24] = 0
    elif case == "defective":
        A = np.triu(A); A[np.arange(k), np.arange(k)] = 2.0 + 1j            # one eigenvalue of multiplicity k
        A[np.arange(1, k), np.arange(k - 1)] = 0
        eig_tol = 1e-12
    elif case == "graded":
        d = np.logspace(-6, 6, k); A = (d[:, None] * A) / d[None, :] * 1.0
        A = np.triu(A, -1); eig_tol = 1e-9; res_tol = 1e-12
    elif case == "zero":
        A = np.zeros((k, k), dtype=complex)
    elif case == "k128":                                                      # the largest size the packed LDS layout takes
        k = 128
        A = np.triu(rng.standard_normal((k, k)) + 1j * rng.standard_normal((k, k)), -1)
    if case == "defective":
        import torch
        from nep_amd import dense
        Hd = torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda")
        w, Z = dense.hess_eig_dev(Hd, k)
        wh = w.cpu().numpy(); Zh = Z.cpu().numpy().T
        assert wh[k].real == 0 and np.abs(wh[:k] - (2.0 + 1j)).max() < 1e-12
        assert np.all(np.isfinite(Zh))                                       # vectors of a defective matrix: finite, unit norm or flagged
        return
    _hess_eig_check(na, A, eig_tol, res_tol)


def test_hess_eig_dev_reads_the_iar_row_layout(na):
    """the matrix is read in place from nep_iar_step's device H block: row j of the (m, m + 4) block = column j of H (h[0..j],
    beta at j + 1, then flags / recorded omegas that lie BELOW the first subdiagonal and must be ignored)"""
    import os
    import torch
    from nep_amd import dense
    m = 30; k = 22
    H = np.load(os.path.join(os.path.dirname(__file__), "golden", "gun_iar_H100.npy"))[:k, :k]
    blk = np.full((m, m + 4), 7e300 + 3e300j)                                # poison everywhere the kernel must not look
    for j in range(k):
        blk[j, :min(j + 2, k)] = H[:min(j + 2, k), j]
    Hd = torch.from_numpy(blk).to("cuda")
    w, Z = dense.hess_eig_dev(Hd, k, ldh=m + 4)
    wh = w.cpu().numpy()
    assert wh[k].real == 0 and wh[k + 1].real == 0
    ref = np.linalg.eigvals(H)
    assert np.abs(np.sort_complex(wh[:k]) - np.sort_complex(ref)).max() <= 1e-11 * np.abs(ref).max()
    with pytest.raises(na.NepError):
        dense.hess_eig_worksize(129)                                         # LDS-resident limit: the caller keeps LAPACK there


@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100003, 3000017])
def test_own_scan_and_radix_sort(na, n):
    """csrc/devprims.h (the library's own exclusive scan and stable radix sort, which replaced hipCUB in the plan enumeration of the
    device LU): against NumPy, sizes around the tile boundaries, 64-bit sums that carry into the high word (the enumeration packs two
    counters into one item), keys with many duplicates (stability: values of equal keys keep their input order), bitwise repeatable"""
    import torch
    from nep_amd._lib import lib, check, c_vp
    rng = np.random.default_rng(n)
    x = rng.integers(0, 3, n).astype(np.uint64) + (rng.integers(0, 2, n).astype(np.uint64) << np.uint64(32))
    xd = torch.from_numpy(x.view(np.int64)).to("cuda"); od = torch.empty_like(xd)
    check(lib.nep_devprim_exclusive_sum(c_vp(xd.data_ptr()), c_vp(od.data_ptr()), n, None))
    want = np.concatenate([[0], np.cumsum(x)[:-1]]).astype(np.uint64)
    assert np.array_equal(od.cpu().numpy().view(np.uint64), want)
    for nbits, hi in ((9, 1 << 9), (40, 1 << 40), (47, 37)):            # few distinct keys in the last case
        keys = rng.integers(0, hi, n, dtype=np.int64).astype(np.uint64)
        vals = np.arange(n, dtype=np.uint64) * np.uint64(3)
        out = []
        for rep in range(2):
            kd = torch.from_numpy(keys.view(np.int64).copy()).to("cuda"); vd = torch.from_numpy(vals.view(np.int64).copy()).to("cuda")
            check(lib.nep_devprim_sort_pairs(c_vp(kd.data_ptr()), c_vp(vd.data_ptr()), n, nbits, None))
            out.append((kd.cpu().numpy().view(np.uint64), vd.cpu().numpy().view(np.uint64)))
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(out[0][0], keys[order]) and np.array_equal(out[0][1], vals[order])
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
