"""Host sparse LU (SuperLU via SciPy) as a plain function of arrays, importable WITHOUT torch/HIP so that it can run
in worker processes (Beyn: one new matrix per quadrature node, src/method_beyncontour.jl:89-94; SciPy's splu releases
the GIL for other Python threads but two splu calls of one process do not overlap -- measured: 4 threads take as long
as 4 sequential calls -- so concurrent factorisations need processes, while ONE background thread is enough to overlap a
factorisation with device work, see linsolvers.LinSolverCache.prefetch).  Returns the factors in the compressed-column form
SuperLU produces (nep_lu_create_csc; csr=True converts to the CSR form of nep_lu_create)."""
import threading
import os
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


_CTL = [False]


def blas_controller():
    """one cached threadpoolctl controller (constructing one scans every loaded shared library: ~50-100 ms in a
    process that has torch loaded, so never do it per call)"""
    if _CTL[0] is False:
        try:
            from threadpoolctl import ThreadpoolController
            _CTL[0] = ThreadpoolController()
        except Exception:
            _CTL[0] = None
    return _CTL[0]


_CAP = [None]


def cap_blas_threads(nmax=8):
    """Permanently caps the BLAS thread pools of this process at `nmax` threads (NEP_BLAS_THREADS overrides, 0 = leave
    alone).  Measured on a 2 x 64-core host: with the default 64-128 thread OpenBLAS pools, every change of the
    thread count (threadpoolctl around SuperLU / the k x k LAPACK calls) and every large allocation stalls the
    launching thread for 50-100 ms at random places (splu 22 -> 99 ms, CSR conversion 5 -> 70 ms); with <= 8 threads
    the same code is stable at 26-28 ms.  All host BLAS in this backend is small (k <= a few hundred)."""
    import os
    if _CAP[0] is not None:
        return
    try:
        nmax = int(os.environ.get("NEP_BLAS_THREADS", nmax))
    except ValueError:
        pass
    _CAP[0] = False
    if nmax <= 0:
        return
    ctl = blas_controller()
    if ctl is None:
        return
    try:
        cur = max([lib.num_threads for lib in ctl.lib_controllers if lib.user_api == "blas"] + [0])
        if cur > nmax:
            _CAP[0] = ctl.limit(limits=nmax, user_api="blas")     # kept alive: never restored
    except Exception:
        pass


def physical_cores(allowed=None):
    """one logical CPU per physical core among `allowed` (default: this process' affinity mask), in increasing order"""
    import os
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    seen, out = set(), []
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib); out.append(c)
    return out


def pin_worker(counter, cpus):
    """pool initializer: worker number i runs on cpus[i % len(cpus)] only.  Two SuperLU workers that the scheduler puts on
    the SMT siblings of one core each run at half speed (measured: splu 16 ms alone, 34 ms with 16 unpinned workers on a
    64-core / 128-thread socket)"""
    import os
    try:
        with counter.get_lock():
            i = counter.value
            counter.value += 1
        if cpus:
            os.sched_setaffinity(0, {cpus[i % len(cpus)]})
    except Exception:
        pass


def ping(i):
    time.sleep(0.02)      # keeps the first tasks from all landing on one worker while the others still start
    return i


def pattern_symmetric(Ac, threshold=0.5):
    """UMFPACK's strategy test: the symmetric strategy is chosen when the non-zero pattern is (nearly) symmetric and
    the diagonal is zero-free.  Symmetry = fraction of off-diagonal entries whose transposed position is also stored
    (1.0 for gun, 0.997 for the WEP whose C1 / C2^T coupling blocks differ by one stencil point)."""
    P = sp.csc_matrix((np.ones(Ac.nnz, dtype=np.int8), Ac.indices, Ac.indptr), shape=Ac.shape)
    if not bool(np.all(Ac.diagonal() != 0)):
        return False
    both = P.multiply(P.T)
    nd = P.nnz - Ac.shape[0]
    sym = 1.0 if nd <= 0 else (both.nnz - Ac.shape[0]) / nd
    return sym >= threshold


class _blas_limit:
    """`with _blas_limit(n)`: process-wide BLAS thread limit for the duration of a factorisation.  Re-entrant across
    threads (several factorisations may run on worker threads at once): the first one in sets the limit, the last one
    out restores it, so an interleaved exit cannot leave the process at the inner value."""
    _lock = threading.Lock()
    _depth = 0
    _ctx = None

    def __init__(self, nthreads):
        self.nthreads = nthreads

    def __enter__(self):
        ctl = blas_controller()
        if ctl is None:
            return self
        cls = _blas_limit
        with cls._lock:
            if cls._depth == 0:
                cls._ctx = ctl.limit(limits=self.nthreads, user_api="blas")
            cls._depth += 1
        return self

    def __exit__(self, *exc):
        if blas_controller() is None:
            return False
        cls = _blas_limit
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._ctx is not None:
                cls._ctx.restore_original_limits()
                cls._ctx = None
        return False


_SYMMETRY = {}        # pattern digest -> UMFPACK-style symmetry verdict
_ORDERS = {}          # (pattern digest, permc_spec, symmetric) -> (ip, indices, indptr, data map) of the permuted matrix


def _pattern_key(Ac, permc_spec, symmetric):
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    h.update(np.ascontiguousarray(Ac.indptr)); h.update(np.ascontiguousarray(Ac.indices))
    return (h.digest(), Ac.shape, permc_spec, symmetric)


def _cached_order(Ac, permc_spec, symmetric):
    if permc_spec == "NATURAL":
        return None
    return _ORDERS.get(_pattern_key(Ac, permc_spec, symmetric))


def _store_order(Ac, permc_spec, symmetric, perm_c):
    """perm_c: SuperLU's column permutation (column j of A sits at position perm_c[j] of the factored matrix).  Symmetric
    strategy: rows are permuted alike, so that the diagonal stays the diagonal (SymmetricMode prefers diagonal pivots)."""
    if permc_spec == "NATURAL":
        return
    ip = np.argsort(perm_c).astype(np.int64)
    T = sp.csc_matrix((np.arange(1, Ac.nnz + 1, dtype=np.float64), Ac.indices, Ac.indptr), shape=Ac.shape)
    T = (T[ip][:, ip] if symmetric else T[:, ip]).tocsc()
    T.sort_indices()
    if len(_ORDERS) >= 8:
        _ORDERS.pop(next(iter(_ORDERS)))
    _ORDERS[_pattern_key(Ac, permc_spec, symmetric)] = (ip, T.indices.copy(), T.indptr.copy(), (T.data - 1).astype(np.int64))


def factor(data, indices, indptr, shape, permc_spec=None, diag_pivot_thresh=None, symmetric_mode=None, panel_size=8,
           relax=4, csr=False):
    """UMFPACK-like strategy selection (see linsolvers.DeviceLU) + SuperLU factorisation.
    panel_size / relax: SuperLU's panel width and relaxed-supernode size.  With single-threaded BLAS on these complex
    sparse matrices the defaults (20 / 10) are slower: measured on the bench host gun (n=9956) 20.4 -> 15.5 ms,
    waveguide n=91k 411 -> 348 ms, n=1e6 13.7 -> 13.5 s (scripts/diag/superlu_params.py); same fill, same pivots."""
    t0 = time.perf_counter()
    Ac = sp.csc_matrix((np.asarray(data, dtype=np.complex128), indices, indptr), shape=shape)
    if permc_spec is None or symmetric_mode is None:
        dkey = _pattern_key(Ac, None, None)
        sym = _SYMMETRY.get(dkey)
        if sym is None:
            sym = pattern_symmetric(Ac)               # 1 ms at gun size; a property of the pattern (and a zero-free diagonal)
            if len(_SYMMETRY) >= 64:
                _SYMMETRY.clear()
            _SYMMETRY[dkey] = sym
        elif sym and not bool(np.all(Ac.diagonal() != 0)):
            sym = False
        if permc_spec is None:
            permc_spec = "MMD_AT_PLUS_A" if sym else "COLAMD"
        if symmetric_mode is None:
            symmetric_mode = sym and permc_spec == "MMD_AT_PLUS_A"
    if diag_pivot_thresh is None and symmetric_mode:
        diag_pivot_thresh = 0.001
    kw = dict(permc_spec=permc_spec)
    if panel_size:
        kw["panel_size"] = int(panel_size)
    if relax:
        kw["relax"] = int(relax)
    if diag_pivot_thresh is not None:
        kw["diag_pivot_thresh"] = diag_pivot_thresh
    if symmetric_mode:
        kw["options"] = dict(SymmetricMode=True)
    # SuperLU is sequential and calls small BLAS-2/3 kernels: a many-thread OpenBLAS only adds spinning threads
    # (measured 40 -> 230 ms jitter on a 128-core host), so BLAS is pinned to one thread for the call -- except for
    # large problems whose supernodes are big enough for threaded zgemm/ztrsm to pay (measured: n = 91k 0.36 -> 0.40 s,
    # n = 251k 1.90 -> 1.71 s, n = 1e6 13.5 -> 10.4 s with 8 threads; scripts/diag/superlu_threads.py)
    nthreads = 1 if shape[0] < 200000 else 8
    # The fill-reducing column ordering (MMD / COLAMD + etree postorder) is a function of the sparsity pattern alone.  It is
    # kept per pattern (the 64 nodes of a contour, the factorisations of nleigs, repeated solves of one problem) and the
    # next matrix with that pattern is handed to SuperLU already permuted with permc_spec = "NATURAL": identical pivots and
    # factors, without the ordering pass (gun: 16 -> 13 ms per factorisation).
    order = _cached_order(Ac, permc_spec, bool(symmetric_mode)) if not os.environ.get("NEP_NO_ORDER_CACHE") else None
    with _blas_limit(nthreads):
        if order is None:
            lu = spla.splu(Ac, **kw)  # RuntimeError("Factor is exactly singular") propagates to the caller
            perm_r, perm_c = lu.perm_r, lu.perm_c
            _store_order(Ac, permc_spec, bool(symmetric_mode), perm_c)
        else:
            ip, Bi, Bp, dmap = order
            B = sp.csc_matrix((Ac.data[dmap], Bi, Bp), shape=shape)
            kw["permc_spec"] = "NATURAL"
            lu = spla.splu(B, **kw)
            perm_c = np.empty(shape[0], dtype=np.int32); perm_c[ip] = lu.perm_c
            if symmetric_mode:
                perm_r = np.empty(shape[0], dtype=np.int32); perm_r[ip] = lu.perm_r
            else:
                perm_r = lu.perm_r
    t_factor = time.perf_counter() - t0
    # SuperLU hands L and U out in compressed columns; nep_lu_create_csc takes them as they are (the former CSC -> CSR
    # conversion + index sort cost 3-6 ms per gun factorisation on the host)
    L = lu.L; U = lu.U
    if csr:
        L = sp.csr_matrix(L); U = sp.csr_matrix(U)
        L.sort_indices(); U.sort_indices()
    return dict(
        n=shape[0], fmt="csr" if csr else "csc",
        Lp=np.ascontiguousarray(L.indptr, dtype=np.int32), Li=np.ascontiguousarray(L.indices, dtype=np.int32),
        Lx=np.ascontiguousarray(L.data, dtype=np.complex128),
        Up=np.ascontiguousarray(U.indptr, dtype=np.int32), Ui=np.ascontiguousarray(U.indices, dtype=np.int32),
        Ux=np.ascontiguousarray(U.data, dtype=np.complex128),
        perm_r=np.ascontiguousarray(perm_r, dtype=np.int32), perm_c=np.ascontiguousarray(perm_c, dtype=np.int32),
        normA=float(np.linalg.norm(Ac.data)), t_factor=t_factor, t_total=time.perf_counter() - t0,
        strategy=dict(permc_spec=permc_spec, diag_pivot_thresh=diag_pivot_thresh, symmetric_mode=bool(symmetric_mode)))


_ARRAYS = ("Lp", "Li", "Lx", "Up", "Ui", "Ux", "perm_r", "perm_c")


def factor_shm(data, indices, indptr, shape, **kw):
    """`factor` for worker processes: the arrays travel through ONE POSIX shared-memory block instead of the result
    pipe (pickling + unpickling 26 MB of factors per gun node serialised Beyn's 64 factorisations in the parent).
    Returns the small metadata dict; the parent maps the block with `attach_shm` and unlinks it with `release_shm`."""
    from multiprocessing import shared_memory, resource_tracker
    t_in = time.perf_counter()
    F = factor(data, indices, indptr, shape, **kw)
    t_f = time.perf_counter()
    layout = []
    off = 0
    for key in _ARRAYS:
        a = F[key]
        off = (off + 63) & ~63
        layout.append((key, off, a.dtype.str, a.shape[0]))
        off += a.nbytes
    shm = shared_memory.SharedMemory(create=True, size=max(off, 64))
    for (key, o, dt, cnt) in layout:
        np.frombuffer(shm.buf, dtype=dt, count=cnt, offset=o)[:] = F[key]
    meta = {k: v for k, v in F.items() if k not in _ARRAYS}
    meta.update(shm_name=shm.name, shm_layout=layout, t_worker=time.perf_counter() - t_in, t_worker_factor_call=t_f - t_in)
    try:
        resource_tracker.unregister(shm._name, "shared_memory")      # ownership passes to the parent
    except Exception:
        pass
    shm.close()
    return meta


def attach_shm(meta):
    """parent side: factor dict whose arrays are views of the worker's shared-memory block (keep F['_shm'] alive)"""
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=meta["shm_name"])
    F = {k: v for k, v in meta.items() if k not in ("shm_name", "shm_layout")}
    for (key, o, dt, cnt) in meta["shm_layout"]:
        F[key] = np.frombuffer(shm.buf, dtype=dt, count=cnt, offset=o)
    F["_shm"] = shm
    return F


def release_shm(F):
    shm = F.pop("_shm", None)
    if shm is not None:
        for key in _ARRAYS:
            F.pop(key, None)            # drop the views before closing the mapping
        try:
            shm.close()
        finally:
            try:
                shm.unlink()
            except FileNotFoundError:
                pass
