"""NLEIGS (fully rational Krylov) on the device backend -- keyword surface of src/method_nleigs.jl:60-81.

Supported: SPMF-type NEPs (PEP, SPMF_NEP, PEP+SPMF SumNEP, DEP), dynamic and static variants, return_details
(NleigsSolutionDetails), divided differences by matrix functions (isfunm=true) or by differencing (isfunm=false),
leja in {0,1,2}, reusefact in {0,1,2}, non-SPMF NEP types through matrix-valued divided differences (`Mder_NEP`: the D_j become the
terms of an SPMF in the rational Newton basis), and the low-rank compression of PEP + LowRankFactorizedNEP problems
(rk_nep.jl:128-152; method_nleigs.jl:206-211,406-414,424-430,464-471,480,510: the blocks of the Krylov vectors beyond the
polynomial degree p hold r = sum rank(C_i) rows instead of n -- gun: 84 instead of 9956).

Device realisation of `backslash` (method_nleigs.jl:399-518).  The reference runs O(N) stacked SpMVs per step
(`sum(reshape(BBCC*z_block,n,:) .* transpose(sgdd[:,ii+1]),dims=2)`, :462).  The block recurrence for z does not
depend on z[1:n], so here
   Bw          one pass            (nep_rk_bw)
   z blocks    one pass, sequential over the N blocks inside each thread   (nep_block_recur)
   z0          ONE K1 call with k = N columns:  sum_j A_j (Z_blocks * sgdd[j,2:N+1]^T)   (nep_mlincomb)
   w0          K5 with the cached factorisation of the shift (LinSolverCache), scaled by -1/beta_1
   w blocks    one pass            (nep_block_recur)
With low-rank structure the same steps run over p blocks of n rows and N-p+1 blocks of r rows; the three seams
(Bw_p, z_p, w_p) apply UU^H and the tail of z0 applies [L_1 ... L_q] through the rectangular CSR operator (nep_csr_mv),
the weighted block sum in between is one nep_rowdot.
followed by K6 DGKS on the (N+1) n-row basis with per-column active row counts, and -- every check_error_every
steps -- host `eig(K,H)`, one K7 GEMM for the Ritz block and K2 for all residuals.
"""
import os
import numpy as np
import scipy.linalg as sla
import torch

from . import dense, rk_helper as rk
from ._lib import lib, check, hptr, c_vp
from .errmeasure import ResidualErrmeasure, estimate_errors
from .linsolvers import DefaultLinSolverCreator, LinSolverCache
from .nep import CDT, to_dev, to_host, stream_ptr

EPS = np.finfo(float).eps


def _c128(x):
    return np.ascontiguousarray(x, dtype=np.complex128)


class NleigsSolutionDetails:
    """src/method_nleigs.jl:538-561: Ritz values / residuals per iteration, nodes, poles, scaling, divided-difference
    norms and the iteration at which the linearisation converged"""

    def __init__(self, Lam, Res, sigma, xi, beta, nrmD, kconv):
        self.Lam, self.Res, self.sigma, self.xi, self.beta, self.nrmD, self.kconv = Lam, Res, sigma, xi, beta, nrmD, kconv


def nleigs(nep, *args, **kw):
    """src/method_nleigs.jl (see _nleigs).  The host side of a call is a few hundred small dense operations (divided differences
    of matrix functions: 102 x 102 solves, the pencil's generalised eigenproblem, Leja-Bagby points): with a threaded BLAS each of
    them pays the wake-up of the pool -- 6 ms per 102 x 102 `solve` instead of 0.3 (gun R1: 23 of 86 ms) -- so BLAS runs on one
    thread for the duration of the call (NEP_NLEIGS_BLAS_GUARD=0: left alone), as in iar's loop."""
    import nep_amd_hostlu as _nep_hostlu
    ctl = _nep_hostlu.blas_controller() if os.environ.get("NEP_NLEIGS_BLAS_GUARD", "1") != "0" else None

    def run():
        miss = False
        held = []                      # the run's LinSolverCache: closed on EVERY way out of it (its factors and prefetch pool are device memory)
        try:
            return _nleigs(nep, *args, _cache_holder=held, **kw)
        except _OrthPassMiss:          # "twice is enough" failed for a step of the asynchronous Gram-Schmidt: exact DGKS through the synchronous calls
            miss = True                # (the re-run starts AFTER this block: a live exception keeps the aborted run's frame -- basis, H, the
        finally:                       # cached factorisations -- alive, which would double the peak device memory of the fallback)
            for c_ in held:
                c_.close()
        assert miss
        nleigs.orth_misses += 1
        return _nleigs(nep, *args, _sync_orth=True, **kw)
    if ctl is None:
        return run()
    with ctl.limit(limits=1, user_api="blas"):
        return run()


class _OrthPassMiss(Exception):
    pass


nleigs.orth_misses = 0


def _nleigs(nep, Sigma=(-1.0 - 1j, -1 + 1j, 1 + 1j, 1 - 1j), Xi=(np.inf,), logger=0, maxdgr=100, minit=20, maxit=200,
            linsolvercreator=None, tol=1e-10, tollin=None, v=None, errmeasure=None, isfunm=True, static=False, leja=1,
            nodes=(), reusefact=1, blksize=20, return_details=False, check_error_every=5, info=None, _sync_orth=False,
            _cache_holder=None):
    import warnings
    if tollin is None:
        tollin = max(tol / 10, 100 * EPS)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = ResidualErrmeasure(nep)
    Sigma = np.asarray(Sigma, dtype=complex); Xi = np.asarray(Xi, dtype=float)
    nodes = np.asarray(nodes, dtype=complex)
    from .nep import require_pure_spmf, AbstractSPMF, SPMFDevice
    n = nep.size(1)
    newton_basis = not isinstance(nep, AbstractSPMF)
    if newton_basis:
        # non-SPMF NEP type (method_nleigs.jl:149-153: `D = ratnewtoncoeffs(lam -> compute_Mder(nep, lam), ...)`): the
        # matrix-valued divided differences D_0..D_{maxdgr+1} ARE an SPMF in the rational Newton basis, M ~ sum_j b_j(lam) D_j,
        # whose scalar divided differences are the identity -- so the stacked-CSR kernels below run unchanged on the D_j
        if not hasattr(nep, "compute_Mder"):
            raise TypeError("nleigs needs an SPMF-type NEP or one that provides compute_Mder")
        if maxdgr + 2 > 128:
            raise ValueError("nleigs on a non-SPMF NEP keeps maxdgr+2 divided-difference matrices on the device: maxdgr <= 126")
        p, q, mt, LR = 0, 0, maxdgr + 2, None
    else:
        require_pure_spmf(nep, "nleigs")
        p, q = rk.rk_structure(nep)
        mt = len(nep.get_Av())
        LR = rk.low_rank_structure(nep)
    if LR is not None and not 1 <= p <= 2:
        # the reference reads block p-1 of a two-block vector in its first step (method_nleigs.jl:408-414)
        raise ValueError("nleigs with low-rank structure needs a polynomial part of degree 1 or 2")
    r = LR.r if LR is not None else n
    blk = lambda j: n if (LR is None or j < p) else r                    # rows of block j (method_nleigs.jl:206-211)
    off = lambda j: j * n if (LR is None or j <= p) else p * n + (j - p) * r
    if n == 1:
        maxdgr = maxit + 1
    if v is None:
        v = np.random.randn(n) + 0j
    cache = LinSolverCache(nep, linsolvercreator)
    if _cache_holder is not None:
        _cache_holder.append(cache)

    # ---- interpolation nodes, poles, scaling (method_nleigs.jl:121-146)
    if leja == 0:
        if len(nodes) == 0:
            raise ValueError("Interpolation nodes must be provided via 'nodes' when no Leja-Bagby points ('leja' == 0) are used.")
        gamma, _ = rk.discretizepolygon(Sigma)
        max_count = (maxit + maxdgr + 2) if static else max(maxit, maxdgr) + 2
        sigma = np.tile(nodes, int(np.ceil(max_count / len(nodes))))
        _, xi, beta = rk.lejabagby(sigma[:maxdgr + 2], Xi, gamma, maxdgr + 2, True, p)
    elif leja == 1:
        if len(nodes) == 0:
            gamma, nodes = rk.discretizepolygon(Sigma, True)
        else:
            gamma, _ = rk.discretizepolygon(Sigma)
        nodes = np.tile(nodes, int(np.ceil((maxit + 1) / len(nodes))))
        sigma, xi, beta = rk.lejabagby(gamma, Xi, gamma, maxdgr + 2, False, p)
    else:
        gamma, _ = rk.discretizepolygon(Sigma)
        max_count = (maxit + maxdgr + 2) if static else max(maxit, maxdgr) + 2
        sigma, xi, beta = rk.lejabagby(gamma, Xi, gamma, max_count, False, p)
    sigma = np.array(sigma, dtype=complex); xi = np.array(xi, dtype=float); beta = np.array(beta, dtype=float)
    xi[maxdgr + 1] = np.nan
    rng_ = slice(0, maxdgr + 2)
    if not isfunm and len(np.unique(sigma)) != len(sigma):                          # method_nleigs.jl:142-145
        raise ValueError("All interpolation nodes must be distinct when no matrix functions are used for computing "
                         "the generalized divided differences.")
    if newton_basis:
        if len(np.unique(sigma[rng_])) != len(sigma[rng_]):
            raise ValueError("All interpolation nodes must be distinct when no matrix functions are used for computing "
                             "the generalized divided differences.")
        import scipy.sparse as sp
        Dall = rk.ratnewtoncoeffs(lambda lam_: sp.csr_matrix(nep.compute_Mder(lam_), dtype=np.complex128), sigma[rng_], xi[rng_], beta[rng_])
        nrmD_all = [float(np.sqrt(abs(Dj.multiply(Dj.conj()).sum()))) for Dj in Dall]        # Frobenius norms (:152,225)
        newton_dev = SPMFDevice(Dall)
        sgdd = np.eye(mt, dtype=np.complex128)
        nrmD = [nrmD_all[0]]
    else:
        sgdd = rk.scgendivdiffs(sigma[rng_], xi[rng_], beta[rng_], nep.get_fv(), isfunm)   # mt x (maxdgr+2)
        nrmD = [float(np.max(abs(sgdd[:, 0])))]
    kdev = newton_dev if newton_basis else nep.dev                                          # the operator of the z0 sum
    if not np.isfinite(nrmD[0]):
        raise ValueError("The generalized divided differences must be finite.")

    # ---- device state: V ((kmax+2) n x (kmax+2)), work vectors of (kmax+2) n entries
    # static variant (method_nleigs.jl:101-102,176,250-257): the linearisation is built first (no Krylov steps), then
    # maxit steps run on vectors of the frozen length (N+1) n; the start vector is zero-padded, which the zero-initialised V
    # provides for free.  At most maxdgr+1 blocks.
    kmax = maxit + maxdgr if static else maxit
    ldv = off(min(kmax, maxdgr + 1) + 2) if static else off(kmax + 2)
    ncol = maxit + 2
    V = torch.zeros((ncol, ldv), dtype=CDT, device="cuda")
    Bw = torch.empty(ldv, dtype=CDT, device="cuda")
    zb = torch.empty(ldv, dtype=CDT, device="cuda")
    tmp = torch.empty(n, dtype=CDT, device="cuda")
    if LR is not None:
        ylr = torch.empty(r, dtype=CDT, device="cuda")
        Wlr = to_dev(sgdd[p + 1 + LR.iL, :])          # (maxdgr+2, r): column ii holds dd[iL] of method_nleigs.jl:464
    H = np.zeros((ncol, ncol - 1), dtype=complex); K = np.zeros((ncol, ncol - 1), dtype=complex)
    # Asynchronous Gram-Schmidt (default): the rational Krylov recurrence itself never reads H -- the continuation vector is the last
    # basis vector (method_nleigs.jl:287-288) and shifts, poles and block sizes are host data -- so the orthogonalisation of a step
    # is enqueued with the DGKS decision on the device (nep_orth_dev: h, beta and the flags stay in row l - 1 of Hdev) and the rows
    # are fetched in ONE copy when the pencil (K, H) is needed: at a convergence check and at the end.  The step-synchronous
    # nep_orth (h and beta read back in every step: the host waited for the device and the device for the host, 100 times per call)
    # remains for NEP_NLEIGS_SYNC=1 and as the fallback when a step still wanted a third pass.
    async_orth = not _sync_orth and os.environ.get("NEP_NLEIGS_SYNC", "0") == "0"
    if async_orth:
        Hdev = torch.zeros((ncol, ncol + 2), dtype=CDT, device="cuda")
        active_d = torch.zeros(ncol, dtype=torch.int64, device="cuda")
    pending = []                                       # (l, k) of the steps whose row of Hdev has not been read yet

    def flush_H():
        if not pending:
            return
        l0 = pending[0][0]; l1 = pending[-1][0]
        rows_ = Hdev[l0 - 1:l1].cpu().numpy()          # one device-to-host copy (synchronises)
        for (l_, k_) in pending:
            row = rows_[l_ - l0]
            flags = int(row[l_ + 1].imag)
            if flags & 2:
                raise ArithmeticError("orthogonalisation breakdown in nleigs step %d" % k_)
            if flags & 1:
                raise _OrthPassMiss(k_)
            H[:l_, l_ - 1] = row[:l_]; H[l_, l_ - 1] = row[l_].real
            K[:l_, l_ - 1] = H[:l_, l_ - 1] * sigma[k_]
            K[l_ - 1, l_ - 1] += 1.0
            K[l_, l_ - 1] = H[l_, l_ - 1] * sigma[k_]
        pending.clear()
    Lam = np.zeros((ncol - 1, ncol - 1), dtype=complex); Res = np.zeros((ncol - 1, ncol - 1))
    active = np.zeros(ncol, dtype=np.int64)
    st = stream_ptr

    v0 = _c128(v) / np.linalg.norm(v)
    # host factorisations of the next shifts run ahead on a worker thread; with reusefact = 2 (every factorisation is kept) and a
    # device-factorisation plan for the pattern, ALL distinct shifts are factorised on the GPU in one batched pass instead
    cache.prefetch(sigma[:3], keep_all=list(dict.fromkeys(complex(s_) for s_ in sigma)) if reusefact == 2 else None)
    x0 = cache.solve(sigma[0], to_dev(v0)[0], reusefact == 2)
    nx0 = dense.nrm2(x0)
    dense.copy(x0, V[0], n); dense.scal(V[0], 1.0 / nx0, n)
    active[0] = n

    expand = True
    kconv = np.iinfo(np.int64).max // 2
    kn = n; l = 0; N = 0; nbconv = 0; nblamin = 0
    res_state = {"lam": np.zeros(0, dtype=complex), "QT": None, "ilam": np.zeros(0, dtype=int), "res": np.zeros(0),
                 "conv": np.zeros(0, dtype=bool)}

    def backslash(k, l):
        """w = continuation solve for column l-1 of V; result in V[l][:kn]"""
        shift = sigma[k]
        wc = V[l - 1]
        with np.errstate(all="ignore"):
            cB = _c128(beta[1:N + 1] / xi[:N])                                    # beta[ii+1]/xi[ii]
            check(lib.nep_rk_bw(n, N, c_vp(wc.data_ptr()), hptr(cB), c_vp(Bw.data_ptr()), st()))
            # z blocks: z_1 = Bw_1/nu_1 ; z_i = Bw_i/nu_i + (mu_i/nu_i) z_{i-1}
            nu = beta[1:N + 1] * (1 - shift / xi[:N])
            a = _c128(1.0 / nu)
            b = np.zeros(N, dtype=np.complex128)
            if N > 1:
                b[1:] = (shift - sigma[1:N]) / nu[1:]
            dense.copy(Bw, zb, (N + 1) * n)
            check(lib.nep_block_recur(n, N, hptr(a), hptr(b), c_vp(zb.data_ptr()), c_vp(zb.data_ptr()), st()))
            # z0 = -sum_j A_j (Zblocks sgdd[j, 1:N+1]^T)   (Bw[0:n] = 0 without low-rank structure)
            Cm = np.asfortranarray(sgdd[:, 1:N + 1].T)                             # N x mt
            kdev.mlincomb(Cm, zb.data_ptr() + 16 * n, tmp, k=N, ldv=n)
            add_to_cache = ((not expand or k > kconv) and reusefact == 1) or reusefact == 2
            w = V[l]
            w0 = cache.solve_dev(shift, tmp, add_to_cache, out=w[:n], scale=-1.0 / beta[0])
            # w_i = (mu_i/nu_i) w_{i-1} + Bw_i/nu_i
            mu = shift - sigma[:N]
            bw = _c128(mu / nu)
            check(lib.nep_block_recur(n, N, hptr(a), hptr(bw), c_vp(Bw.data_ptr()), c_vp(w.data_ptr()), st()))
        return w

    def backslash_lowrank(k, l):
        """the same solve over p blocks of n rows and N-p+1 blocks of r rows (low-rank branches of method_nleigs.jl:399-518);
        needs N >= p-1, which p <= 2 guarantees"""
        shift = sigma[k]
        wc = V[l - 1]; w = V[l]
        at = lambda T, j: T.data_ptr() + 16 * off(j)
        with np.errstate(all="ignore"):
            cB = _c128(beta[1:N + 1] / xi[:N])
            nu = beta[1:N + 1] * (1 - shift / xi[:N])
            a = _c128(1.0 / nu)
            b = np.zeros(N, dtype=np.complex128)
            if N > 1:
                b[1:] = (shift - sigma[1:N]) / nu[1:]
            bw = _c128((shift - sigma[:N]) / nu)
            nh = min(N, p - 1)                 # n-row blocks 1..p-1 follow the plain recurrences
            nr = N - p                         # r-row blocks p+1..N
            # ---- Bw (:417-436); its first block (:407-415) goes straight into the K1 call below
            if nh > 0:
                check(lib.nep_rk_bw(n, nh, c_vp(wc.data_ptr()), hptr(_c128(cB[:nh])), c_vp(Bw.data_ptr()), st()))
            if nr > 0:
                check(lib.nep_rk_bw(r, nr, c_vp(at(wc, p)), hptr(_c128(cB[p:])), c_vp(at(Bw, p)), st()))
            if nr >= 0:                        # seam: Bw_p = UU^H wc_{p-1} + beta_p/xi_{p-1} wc_p
                LR.UUH.mv(1.0, at(wc, p - 1), cB[p - 1], at(wc, p), at(Bw, p))
            # ---- z blocks (:438-491); block 0 of zb carries wc_{p-1} for the K1 call
            check(lib.nep_dev_copy(c_vp(zb.data_ptr()), c_vp(at(wc, p - 1)), 16 * n, st()))
            check(lib.nep_dev_copy(c_vp(at(zb, 1)), c_vp(at(Bw, 1)), 16 * (off(N + 1) - n), st()))
            if nh > 0:
                check(lib.nep_block_recur(n, nh, hptr(_c128(a[:nh])), hptr(_c128(b[:nh])), c_vp(zb.data_ptr()),
                                          c_vp(zb.data_ptr()), st()))
            if nr >= 0:                        # seam: z_p = Bw_p/nu_p + mu_p/nu_p UU^H z_{p-1}   (mu_1/nu_1 := 0)
                LR.UUH.mv(b[p - 1], at(zb, p - 1), a[p - 1], at(zb, p), at(zb, p))
            if nr > 0:
                check(lib.nep_block_recur(r, nr, hptr(_c128(a[p:])), hptr(_c128(b[p:])), c_vp(at(zb, p)),
                                          c_vp(at(zb, p)), st()))
            # ---- -z0 = D_p wc_{p-1}/beta_p + sum_{j<p} D_j z_j + [L_1..L_q] (sum_{j>p} dd_j o z_j)   (:407-415,455-471)
            Cm = np.zeros((p, mt), dtype=np.complex128, order="F")
            Cm[0, :] = sgdd[:, p] / beta[p]
            for j in range(1, p):
                Cm[j, :] = sgdd[:, j]
            if p >= 2 and N >= p:
                # (shift - sigma_{p-1})/beta_p D_p z_{p-1}: the elimination of block p-1 of the first block row leaves this
                # term; the reference omits it (its low-rank tests all have p = 1, where z_0 = 0) -- see oracle/nleigs.py
                Cm[p - 1, :] += b[p - 1] * sgdd[:, p]
            nep.dev.mlincomb(Cm, zb.data_ptr(), tmp, k=p, ldv=n)
            if nr > 0:
                check(lib.nep_rowdot(r, nr, c_vp(at(zb, p + 1)), r, c_vp(Wlr.data_ptr() + 16 * r * (p + 1)), r,
                                     c_vp(ylr.data_ptr()), st()))
                LR.Lall.mv(1.0, ylr, 1.0, tmp, tmp)
            add_to_cache = ((not expand or k > kconv) and reusefact == 1) or reusefact == 2
            cache.solve_dev(shift, tmp, add_to_cache, out=w[:n], scale=-1.0 / beta[0])
            # ---- w blocks (:496-515)
            if nh > 0:
                check(lib.nep_block_recur(n, nh, hptr(_c128(a[:nh])), hptr(_c128(bw[:nh])), c_vp(Bw.data_ptr()),
                                          c_vp(w.data_ptr()), st()))
            if nr >= 0:                        # seam: w_p = mu_p/nu_p UU^H w_{p-1} + Bw_p/nu_p
                LR.UUH.mv(bw[p - 1], at(w, p - 1), a[p - 1], at(Bw, p), at(w, p))
            if nr > 0:
                check(lib.nep_block_recur(r, nr, hptr(_c128(a[p:])), hptr(_c128(bw[p:])), c_vp(at(Bw, p)),
                                          c_vp(at(w, p)), st()))
        return w

    if LR is not None:
        backslash = backslash_lowrank

    def check_convergence(k, l, all_=False):
        flush_H()
        lambda_, S = sla.eig(K[:l, :l], H[:l, :l])
        if not all_:
            lamin = rk.in_Sigma(lambda_, Sigma, tol)
            ilam = np.nonzero(lamin)[0]
            lam = lambda_[ilam]
        else:                                                   # method_nleigs.jl:309-313: every finite Ritz value
            ilam = np.nonzero(np.isfinite(lambda_))[0]
            lam = lambda_[ilam]
            lamin = rk.in_Sigma(lam, Sigma, tol)
        S = S.copy()
        for i in ilam:
            S[:, i] /= np.linalg.norm(H[:l + 1, :l] @ S[:, i])
        if len(ilam):
            QT = dense.gemm_ts(V, H[:l + 1, :l] @ S[:, ilam], rowmajor=True, k=l + 1, rows=n, ldz=ldv)
            res = estimate_errors(errmeasure, lam, QT)
        else:
            QT = None; res = np.zeros(0)
        conv = np.abs(res) < tol
        if all_:                                                # :328-336 history of all Ritz values / residuals
            resall = np.full(l, np.nan)
            resall[ilam] = res
            si = sorted(range(l), key=lambda i: (abs(lambda_[i]), np.angle(lambda_[i])))
            Res[:l, l - 1] = resall[si]
            Lam[:l, l - 1] = lambda_[si]
            conv = conv & lamin
        res_state.update(lam=lam, QT=QT, ilam=ilam, res=res, conv=conv)
        return int(np.sum(lamin)), int(np.sum(conv))

    k = 1
    while k <= kmax:
        if expand:
            kn += blk(k)
            N += 1
            nrmD.append(nrmD_all[k] if newton_basis else float(np.max(abs(sgdd[:, k]))))
            if not np.isfinite(nrmD[k]):
                raise ValueError("The generalized divided differences must be finite.")
            if n > 1 and k >= 5 and k < kconv:
                frozen = False
                if sum(nrmD[k - 4:k + 1]) < 5 * tollin:
                    kconv = k - 1
                    frozen = True
                    xi = xi[:k]; beta = beta[:k]; nrmD = nrmD[:k]
                    if static:
                        kmax = maxit + kconv
                        kn -= blk(k)
                elif k == maxdgr + 1:
                    kconv = k
                    frozen = True
                    warnings.warn("NLEIGS: Linearization not converged after %d iterations" % maxdgr)
                if frozen:
                    expand = False
                    if leja == 1:
                        if len(sigma) < kmax + 1:
                            sigma = np.concatenate([sigma, np.zeros(kmax + 1 - len(sigma), dtype=complex)])
                        sigma[k:kmax + 1] = nodes[:kmax - k + 1]
                    N -= 1
        l = k - N if static else k
        if not static or not expand:
            cache.prefetch(sigma[k:k + 3])
            w = backslash(k, l)
            active[l] = kn
            if async_orth:
                if not pending or pending[-1][0] != l - 1 or l == 1:
                    active_d[:l + 1].copy_(torch.from_numpy(active[:l + 1]))        # (first step, or after a gap: the whole prefix)
                else:
                    active_d[l:l + 1].fill_(int(kn))
                dense.orthogonalize_and_normalize_dev(V, w, l, Hdev[l - 1], rows=kn, ldv=ldv, active_dev=active_d, method=dense.DGKS)
                pending.append((l, k))
            else:
                h, hb, _ = dense.orthogonalize_and_normalize(V, w, l, rows=kn, ldv=ldv, active_rows=active, method=dense.DGKS)
                H[:l, l - 1] = h; H[l, l - 1] = hb
                K[:l, l - 1] = H[:l, l - 1] * sigma[k]
                K[l - 1, l - 1] += 1.0
                K[l, l - 1] = hb * sigma[k]
        if not return_details and (
                (not expand and k >= N + minit and (k - (N + minit)) % check_error_every == 0) or
                (k >= kconv + minit and (k - (kconv + minit)) % check_error_every == 0) or k == kmax):
            nblamin, nbconv = check_convergence(k, l)
        elif return_details and (not static or not expand):
            nblamin, nbconv = check_convergence(k, l, True)
        if ((not expand and k >= N + minit) or k >= kconv + minit) and nblamin == nbconv:
            break
        k += 1
    flush_H()
    lam = res_state["lam"]; conv = res_state["conv"]; res = res_state["res"]
    cache.close()
    if info is not None:
        info.update(kconv=kconv, N=N, k=min(k, kmax), nfact=len(cache.solvers), nrmD=nrmD, nblamin=nblamin, vrows=ldv,
                    lowrank_r=(r if LR is not None else 0))
    if res_state["QT"] is None or not np.any(conv):
        X = np.zeros((n, 0), dtype=complex)
    else:
        X = to_host(dense.rowmajor_to_cols(res_state["QT"], np.nonzero(conv)[0]))
        X = X / np.linalg.norm(X, axis=0)[None, :]
    if return_details:                                          # method_nleigs.jl:363-374
        kk = min(k, kmax)
        if expand:
            xi = xi[:kk]; beta = beta[:kk]; nrmD = nrmD[:kk]
            warnings.warn("NLEIGS: Linearization not converged after %d iterations" % maxdgr)
        return lam[conv], X, res[conv], NleigsSolutionDetails(Lam[:l, :l], Res[:l, :l], sigma[:kk], xi, beta, nrmD, kconv)
    return lam[conv], X, res[conv]
