"""NumPy prototype of the device Hessenberg eigen-solver (csrc/hesseig.hip): explicit single-shift QR (EISPACK comqr's
structure, LAPACK zlahqr's shift / deflation tests) for the eigenvalues, zlaein-style inverse iteration for the vectors.
Development aid only (not imported by the product or the tests)."""
import sys
import numpy as np

ULP = np.finfo(float).eps
SAFMIN = np.finfo(float).tiny


def cabs1(z):
    return abs(z.real) + abs(z.imag)


def lartg(f, g):
    """c real, s complex, r with [[c, s], [-conj(s), c]] [f; g] = [r; 0]"""
    if g == 0:
        return 1.0, 0j, f
    if f == 0:
        ag = abs(g)
        return 0.0, np.conj(g) / ag, ag
    f2 = f.real ** 2 + f.imag ** 2; g2 = g.real ** 2 + g.imag ** 2; h2 = f2 + g2
    d = np.sqrt(f2 * h2)
    c = f2 / d
    s = np.conj(g) * (f / d)
    r = f * (h2 / d)
    return c, s, r


def hess_eigvals(H, stats=None):
    H = np.array(H, dtype=complex)
    n = H.shape[0]
    w = np.zeros(n, dtype=complex)
    smlnum = SAFMIN * (n / ULP)
    i = n - 1
    itmax = 30 * max(10, n)
    sweeps = 0; steps = 0
    kdefl = 0
    while i >= 0:
        l = 0
        conv = False
        for its in range(itmax + 1):
            # look for a small subdiagonal
            k = i
            while k > l:
                if cabs1(H[k, k - 1]) <= smlnum:
                    break
                tst = cabs1(H[k - 1, k - 1]) + cabs1(H[k, k])
                if tst == 0:
                    if k - 2 >= 0:
                        tst += cabs1(H[k - 1, k - 2])
                    if k + 1 <= n - 1:
                        tst += cabs1(H[k + 1, k])
                if cabs1(H[k, k - 1]) <= ULP * tst:
                    ab = max(cabs1(H[k, k - 1]), cabs1(H[k - 1, k])); ba = min(cabs1(H[k, k - 1]), cabs1(H[k - 1, k]))
                    aa = max(cabs1(H[k, k]), cabs1(H[k - 1, k - 1] - H[k, k])); bb = min(cabs1(H[k, k]), cabs1(H[k - 1, k - 1] - H[k, k]))
                    s = aa + ab
                    if ba * (ab / s) <= max(smlnum, ULP * (bb * (aa / s))):
                        break
                k -= 1
            l = k
            if l > 0:
                H[l, l - 1] = 0
            if l >= i:
                conv = True
                break
            kdefl += 1
            # shift
            if kdefl % 20 == 0:
                t = 0.75 * cabs1(H[i, i - 1]) + H[i, i]
            elif kdefl % 10 == 0:
                t = 0.75 * cabs1(H[l + 1, l]) + H[l, l]
            else:
                t = H[i, i]
                u = np.sqrt(H[i - 1, i]) * np.sqrt(H[i, i - 1])
                s = cabs1(u)
                if s != 0:
                    x = 0.5 * (H[i - 1, i - 1] - t)
                    sx = cabs1(x)
                    s = max(s, sx)
                    y = s * np.sqrt((x / s) ** 2 + (u / s) ** 2)
                    if sx > 0 and (x / sx).real * y.real + (x / sx).imag * y.imag < 0:
                        y = -y
                    t = t - u * (u / (x + y))
            # explicit QR step on the window [l, i]
            sweeps += 1; steps += i - l
            cs = np.zeros(i + 1); sn = np.zeros(i + 1, dtype=complex)
            up = H[l, l:i + 1].copy(); up[0] -= t                   # running upper row, columns l..i
            for j in range(l + 1, i + 1):
                lo = H[j, l:i + 1].copy(); lo[:j - 1 - l] = 0; lo[j - l] -= t
                c, s, r = lartg(up[j - 1 - l], lo[j - 1 - l])
                cs[j] = c; sn[j] = s
                newup = c * up + s * lo
                newlo = -np.conj(s) * up + c * lo
                newup[j - 1 - l] = r; newlo[j - 1 - l] = 0
                newup[:j - 1 - l] = 0
                H[j - 1, l:i + 1] = np.where(np.arange(l, i + 1) >= j - 1, newup, H[j - 1, l:i + 1])
                up = newlo
            H[i, i] = up[i - l]
            # (rows l..i of the window now hold R; entries left of the diagonal inside the window are zero)
            for j in range(l + 1, i + 1):
                H[j, j - 1] = 0
            # RQ: columns j-1, j  <-  [c y + conj(s) z, -s y + c z]
            for j in range(l + 1, i + 1):
                c = cs[j]; s = sn[j]
                y = H[l:j + 1, j - 1].copy(); z = H[l:j + 1, j].copy()
                H[l:j + 1, j - 1] = c * y + np.conj(s) * z
                H[l:j + 1, j] = -s * y + c * z
            for j in range(l, i + 1):
                H[j, j] += t
        if not conv:
            return None
        w[i] = H[i, i]
        kdefl = 0
        i = l - 1
    if stats is not None:
        stats["sweeps"] = sweeps; stats["steps"] = steps
    return w


def hess_eigvecs(H, w):
    """zhsein / zlaein (right vectors, no initial vector): LU of H - w I with row interchanges, only U is used"""
    H = np.asarray(H, dtype=complex)
    n = H.shape[0]
    hnorm = np.max(np.sum(np.abs(np.triu(H, -1)), axis=1))
    eps3 = hnorm * ULP if hnorm > 0 else SAFMIN * (n / ULP)
    smlnum = SAFMIN * (n / ULP)
    w = np.array(w, dtype=complex)
    wk = w.copy()
    for k in range(n):                       # perturb close eigenvalues (zhsein)
        again = True
        while again:
            again = False
            for i in range(k - 1, -1, -1):
                if cabs1(wk[i] - wk[k]) < eps3:
                    wk[k] += eps3; again = True; break
    Z = np.zeros((n, n), dtype=complex); fail = 0
    rootn = np.sqrt(n); growto = 0.1 / rootn
    for e in range(n):
        B = np.triu(H, -1).copy() - wk[e] * np.eye(n)
        for i in range(n - 1):
            ei = H[i + 1, i]
            if cabs1(B[i, i]) < cabs1(ei):
                x = B[i, i] / ei
                B[i, i] = ei
                tmp = B[i + 1, i + 1:].copy()
                B[i + 1, i + 1:] = B[i, i + 1:] - x * tmp
                B[i, i + 1:] = tmp
            else:
                if B[i, i] == 0:
                    B[i, i] = eps3
                x = ei / B[i, i]
                if x != 0:
                    B[i + 1, i + 1:] -= x * B[i, i + 1:]
        if B[n - 1, n - 1] == 0:
            B[n - 1, n - 1] = eps3
        U = np.triu(B)
        v = np.full(n, eps3, dtype=complex)
        ok = False
        for its in range(1, n + 1):
            x = np.zeros(n, dtype=complex)
            b = v.copy()
            for i in range(n - 1, -1, -1):
                x[i] = b[i] / U[i, i]
                b[:i] -= U[:i, i] * x[i]
            v = x
            if np.sum(np.abs(v.real) + np.abs(v.imag)) >= growto:
                ok = True; break
            rtemp = eps3 / (rootn + 1)
            v = np.full(n, rtemp, dtype=complex); v[0] = eps3; v[n - its] -= eps3 * rootn
        if not ok:
            fail += 1
        big = v[np.argmax(np.abs(v))]
        Z[:, e] = v * (abs(big) / big) / np.linalg.norm(v)
    return Z, fail


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    mats = []
    for f in sys.argv[1:]:
        mats.append((f, np.load(f)))
    for n in (5, 30, 64, 100):
        A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        mats.append(("rand%d" % n, np.triu(A, -1)))
    for name, H in mats:
        for k in sorted({H.shape[0], max(1, H.shape[0] // 2), max(1, H.shape[0] // 4)}):
            Hk = H[:k, :k]
            st = {}
            w = hess_eigvals(Hk, st)
            ref = np.linalg.eigvals(Hk)
            # match
            used = np.zeros(k, bool); worst = 0
            for x in w:
                d = np.abs(ref - x); d[used] = np.inf; j = np.argmin(d); used[j] = True; worst = max(worst, d[j] / max(abs(x), 1e-300))
            Z, fail = hess_eigvecs(Hk, w)
            res = np.linalg.norm(Hk @ Z - Z * w[None, :], axis=0) / np.linalg.norm(Hk)
            print(name, k, "sweeps", st["sweeps"], "steps", st["steps"], "max rel eig diff %.2e" % worst, "invit fail", fail,
                  "resid max %.2e med %.2e" % (res.max(), np.median(res)))
