import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, torch, nep_amd as na
# randomized parity sweep of the C-ABI kernels against NumPy on ragged / degenerate sizes
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
f = na.funcs
bad = 0
def chk(name, err, tol, info):
    global bad
    if not (err <= tol):
        bad += 1; print("FAIL", name, err, info, flush=True)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 65, 257, 1000, 4099]))
    mt = int(rng.integers(1, 6))
    dens = float(rng.choice([0.0, 0.01, 0.2, 1.0])) if n < 300 else float(rng.choice([0.0, 0.002, 0.02]))
    cp = bool(rng.integers(0, 2))
    AA = []
    for i in range(mt):
        A = sp.random(n, n, dens, random_state=int(rng.integers(1 << 30)), format="csc")
        if cp and i % 2:
            A = A + 1j * sp.random(n, n, dens, random_state=int(rng.integers(1 << 30)), format="csc")
        AA.append(sp.csc_matrix(A))
    fv = [f.one(), f.ident(), f.Exp(-0.3), f.Monomial(2), f.ISqrt(1.0, 2.0)][:mt]
    nep = na.SPMF_NEP(AA, fv)
    k = int(rng.choice([1, 2, 7, 33, 100]))
    V = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    a = rng.standard_normal(k); a[rng.random(k) < 0.2] = 0
    lam = 0.3 + 0.2j
    z = nep.compute_Mlincomb(lam, V, a)
    ref = sum(sum(a[j] * fv[i].derivs(lam, k)[j] * (AA[i] @ V[:, j]) for j in range(k)) for i in range(mt))
    chk("K1", np.linalg.norm(z - ref), 1e-11 * max(1.0, np.linalg.norm(ref)), (n, mt, k, dens, cp))
    # K2
    kk = int(rng.choice([1, 3, 64, 130]))
    Q = rng.standard_normal((n, kk)) + 1j * rng.standard_normal((n, kk))
    lams = rng.standard_normal(kk) * 0.3 + 0.1j
    E = na.ResidualErrmeasure(nep)
    e = E.batch(list(lams), torch.from_numpy(np.ascontiguousarray(Q)).to("cuda"))
    er = np.array([np.linalg.norm(sum(fv[i].derivs(lams[s], 1)[0] * (AA[i] @ Q[:, s]) for i in range(mt))) / np.linalg.norm(Q[:, s]) for s in range(kk)])
    chk("K2", np.max(abs(e - er)), 1e-11 * max(1.0, er.max()), (n, mt, kk))
    # K6
    kq = min(k, n)
    Vq, _ = np.linalg.qr(rng.standard_normal((n, kq)) + 1j * rng.standard_normal((n, kq)))
    w = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Vd = na.to_dev(Vq); wd = na.to_dev(w)[0]
    h, beta, _ = na.orthogonalize_and_normalize(Vd, wd, kq) if n > kq else (None, None, None)
    if h is not None:
        wr = w - Vq @ (Vq.conj().T @ w); wr = wr - Vq @ (Vq.conj().T @ wr)
        chk("K6", abs(beta - np.linalg.norm(wr)), 1e-10 * max(1.0, np.linalg.norm(w)), (n, kq))
    # K7 / K9
    p = int(rng.choice([1, 2, 16, 17, 100]))
    B = rng.standard_normal((k, p)) + 1j * rng.standard_normal((k, p))
    Y = na.to_host(na.gemm_ts(na.to_dev(V), B))
    chk("K7", np.linalg.norm(Y - V @ B), 1e-11 * max(1.0, np.linalg.norm(V @ B)), (n, k, p))
    WT = torch.from_numpy(np.ascontiguousarray(V)).to("cuda"); YT = torch.from_numpy(np.ascontiguousarray(V @ B)).to("cuda")
    Cm = na.dense.gemm_h_rm(WT, YT, n, k, p)
    Cr = V.conj().T @ (V @ B)
    chk("K9", np.linalg.norm(Cm - Cr), 1e-11 * max(1.0, np.linalg.norm(Cr)), (n, k, p))
    # K5
    if n >= 2:
        A = sp.csc_matrix(sum(AA) + sp.identity(n) * (3.0 + mt), dtype=complex)
        nr = int(rng.choice([1, 3, 32]))
        Bm = rng.standard_normal((n, nr)) + 1j * rng.standard_normal((n, nr))
        try:
            lu = na.DeviceLU(A)
            X = na.to_host(lu.solve(na.to_dev(Bm)))
            chk("K5", np.linalg.norm(A @ X - Bm), 1e-9 * np.linalg.norm(Bm), (n, nr, lu.tail, lu.mid_rows))
        except np.linalg.LinAlgError:
            pass
print("done, failures:", bad)
