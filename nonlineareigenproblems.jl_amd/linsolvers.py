"""Linear solvers for M(lam) x = b: host factorisation (one-off per shift), device triangular solves.

Mirrors src/LinSolvers.jl:100-159 and src/LinSolverCreators.jl:11-196:
`create_linsolver(creator, nep, lam)`, `lin_solve(solver, b; tol)`, `FactorizeLinSolver`,
`BackslashLinSolver`, `FactorizeLinSolverCreator(umfpack_refinements, max_factorizations, nep,
precomp_values)`, `BackslashLinSolverCreator`, `DefaultLinSolverCreator`; plus `LinSolverCache`
(src/rk_helper/linsolvercache.jl:7-26).

The reference factorises with UMFPACK on the host (`factorize(A)`, LinSolvers.jl:116).  Here the
host step is SuperLU (SciPy's bundled copy -- the only sparse LU in this image) and the factors
are uploaded once; every lin_solve is the HIP kernel k_lu_solve (csrc/trsv.hip).
"""
import ctypes as C
import os
import threading
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from ._lib import lib, check, hptr, c_vp, c_i64
from .nep import CDT, to_dev, to_host, is_dev, stream_ptr


class LinSolver:
    pass


# the torch-free worker module lives alone in _workers/ under a unique top-level name: only that directory goes on
# sys.path (spawned workers import it by name), so no module of this package can shadow a user's `build`, `nep`, ...
_WORKERS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_workers")
if _WORKERS_DIR not in sys.path:
    sys.path.insert(0, _WORKERS_DIR)
import nep_amd_hostlu as _nep_hostlu  # noqa: E402


class HostLUPool:
    """process pool for concurrent host factorisations (SciPy's splu holds the GIL).  Workers are spawned (never forked
    from a process with a live HIP context) and import only NumPy/SciPy."""
    _pool = None
    _workers = 0

    @classmethod
    def get(cls, workers=None):
        if workers is None and cls._pool is not None:
            return cls._pool                    # whatever size it was started with
        if workers is None:
            from ._affinity import cpu_budget
            workers = int(os.environ.get("NEP_HOSTLU_WORKERS", min(16, max(1, cpu_budget() - 2))))
        if cls._pool is None or cls._workers != workers:
            cls.shutdown()
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            # SuperLU is sequential: one BLAS thread per worker (the children read these variables when they load
            # NumPy; 16 workers x 64 spinning OpenBLAS threads made a factorisation 10x slower)
            keys = ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")
            saved = {k: os.environ.get(k) for k in keys}
            for k in keys:
                os.environ[k] = "1"
            # spawn re-imports the parent's __main__ in every child (multiprocessing.spawn.get_preparation_data); a user
            # script without an `if __name__ == "__main__":` guard would run again inside each worker.  The workers only
            # need _nep_hostlu (importable by name), so __main__ is hidden from the preparation data while they start.
            main = sys.modules.get("__main__")
            hidden = {}
            for attr in ("__file__", "__spec__"):
                if main is not None and getattr(main, attr, None) is not None:
                    hidden[attr] = getattr(main, attr)
                    try:
                        setattr(main, attr, None)
                    except Exception:
                        hidden.pop(attr)
            try:
                ctx = mp.get_context("spawn")
                # one physical core per worker, taken from the END of the allowed set (the launching thread, the eig and
                # builder threads of this process stay on the first cores); NEP_HOSTLU_PIN=0 leaves placement to the scheduler
                cores = _nep_hostlu.physical_cores() if os.environ.get("NEP_HOSTLU_PIN", "1") != "0" else []
                cores = cores[::-1][:max(workers, 1)] if len(cores) >= 2 * workers else []
                cls._pool = ProcessPoolExecutor(max_workers=workers, mp_context=ctx, initializer=_nep_hostlu.pin_worker,
                                                initargs=(ctx.Value("i", 0), cores))
                cls._workers = workers
                list(cls._pool.map(_nep_hostlu.ping, range(4 * workers)))     # forces all workers to start now
            finally:
                for attr, val in hidden.items():
                    setattr(main, attr, val)
                for k_, v_ in saved.items():
                    if v_ is None:
                        os.environ.pop(k_, None)
                    else:
                        os.environ[k_] = v_
        return cls._pool

    @classmethod
    def shutdown(cls):
        if cls._pool is not None:
            cls._pool.shutdown(wait=False, cancel_futures=True)
            cls._pool = None

    @classmethod
    def warm(cls, workers=None):
        """start the workers and wait until each has imported SciPy (keeps the start-up out of timed regions)"""
        cls.get(workers)

    @classmethod
    def submit(cls, A, workers=None, **kw):
        """future of the factor METADATA; the arrays sit in a shared-memory block: DeviceLU(factors=meta) maps, uploads
        and unlinks it; a result that is never consumed must be released by its holder (contour._NodeSolve.close does)"""
        Ac = sp.csc_matrix(A, dtype=np.complex128)
        # `workers` (BackslashLinSolverCreator(workers=N)) sizes the pool when it has to be started; a running pool is kept
        pool = cls._pool if cls._pool is not None else cls.get(workers)
        return pool.submit(_nep_hostlu.factor_shm, Ac.data, Ac.indices, Ac.indptr, Ac.shape, **kw)


import atexit  # noqa: E402
atexit.register(HostLUPool.shutdown)


class _DeviceRefactor:
    """Per sparsity pattern: the plan of the device-side numeric factorisation (csrc/lufac.hip).  The first matrix of a pattern
    is factorised on the host; when that factorisation used the symmetric strategy with diagonal pivots only (perm_r ==
    perm_c) and landed on the block schedule, a background thread enumerates the updates of the right-looking LU for that
    pivot sequence (0.3 s for the gun pattern, once per pattern and process).  Later matrices with the same pattern -- the next
    iar call, the next shift of nleigs -- are factorised on the GPU with the stored pivot sequence (static pivoting); pivot
    breakdown or element growth above `GROWTH` sends that matrix back to the host path."""
    plans = {}           # key -> dict(state="building"|"ready"|"off", handle, thread, strategy, fails)
    lock = threading.Lock()
    GROWTH = float(os.environ.get("NEP_LU_DEV_GROWTH", "1e6"))
    # factors whose solves are NOT followed by iterative refinement (contour_beyn's node solves; the reference pivots afresh at
    # every node): accepted only with element growth max|U| / max|A| and max|L| up to 1e3 -- the growth a threshold-pivoted
    # factorisation with diag_pivot_thresh = 1e-3 tolerates per step -- otherwise that node goes to the host
    GROWTH_UNREFINED = float(os.environ.get("NEP_LU_DEV_GROWTH_UNREFINED", "1e3"))
    MAX = 4

    @classmethod
    def enabled(cls):
        return os.environ.get("NEP_LU_DEV", "1") != "0" and os.environ.get("NEP_LU_SCHED") != "old"

    _digests = []          # (indptr array, indices array, digest) of the last patterns hashed: see key()

    @staticmethod
    def _fingerprint(ip, ix):
        return (ip[::max(1, ip.shape[0] // 256)].tobytes(), ix[::max(1, ix.shape[0] // 768)].tobytes(),
                int(ip[-1]) if ip.shape[0] else 0)

    @classmethod
    def key(cls, Ac, opts):
        import hashlib
        # the matrices of one SPMF NEP share the index arrays of its union pattern (compute_Mder builds them around the NEP's
        # aligned terms without a copy): a pattern whose two arrays ARE arrays hashed before -- same memory, same length, and
        # the remembered arrays are kept alive here, so the address cannot have been recycled -- is not hashed again (0.54 ms
        # per linear solver on the gun pattern)
        ip, ix = Ac.indptr, Ac.indices
        dig = None
        fp = None
        if isinstance(ip, np.ndarray) and isinstance(ix, np.ndarray):
            a_ip = ip.__array_interface__["data"][0]; a_ix = ix.__array_interface__["data"][0]
            # ... and whose CONTENTS still look the same: keeping the arrays alive rules out a recycled address, not an in-place
            # edit (A.sort_indices(), a rewritten A.indices of equal length).  A strided sample of both arrays (about 1 k entries,
            # a few microseconds) travels with the digest; an edit that leaves every sampled entry alone would have to be aimed
            fp = cls._fingerprint(ip, ix)
            for rp, rx, d, f in cls._digests:
                if (rp.__array_interface__["data"][0] == a_ip and rx.__array_interface__["data"][0] == a_ix and rp.shape == ip.shape
                        and rx.shape == ix.shape and rp.dtype == ip.dtype and rx.dtype == ix.dtype
                        and rp.strides == ip.strides and rx.strides == ix.strides and f == fp):
                    dig = d
                    break
        if dig is None:
            h = hashlib.blake2b(digest_size=16)
            # (index dtype normalised: the same pattern arrives as int32 from scipy and as int64 from the NEP's aligned terms)
            h.update(np.ascontiguousarray(ip, dtype=np.int64)); h.update(np.ascontiguousarray(ix, dtype=np.int64))
            dig = h.digest()
            if isinstance(ip, np.ndarray) and isinstance(ix, np.ndarray):
                with cls.lock:
                    cls._digests.append((ip, ix, dig, fp))
                    del cls._digests[:-8]
        knobs = tuple(os.environ.get(k) for k in ("NEP_ML_BMAX", "NEP_ML_SPLIT", "NEP_ML_CHUNK"))   # they change the partition
        return (dig, Ac.shape, opts, knobs)

    @classmethod
    def lookup(cls, key):
        with cls.lock:
            p = cls.plans.get(key)
            return p if (p is not None and p["state"] == "ready") else None

    @classmethod
    def maybe_start(cls, key, lu, F, Ac):
        """called after a host factorisation: start the plan of this pattern if the factor qualifies"""
        if not cls.enabled() or not lu.block_schedule or F.get("fmt") != "csc":
            return
        if not F["strategy"].get("symmetric_mode") or not np.array_equal(F["perm_r"], F["perm_c"]):
            return
        if int(F["Lp"][-1]) + int(F["Up"][-1]) > int(os.environ.get("NEP_LU_DEV_MAXNNZ", "4000000")):
            return        # factors of this size mean far more products than a plan may hold (the library would refuse it anyway)
        with cls.lock:
            if key in cls.plans:
                return
            while len(cls.plans) >= cls.MAX:
                old = next(iter(cls.plans))
                po = cls.plans[old]
                if po["state"] == "building":
                    return
                cls.plans.pop(old)
                if po.get("handle"):
                    lib.nep_lu_refac_destroy(po["handle"])
            plan = dict(state="building", handle=None, strategy=dict(F["strategy"]), fails=0, uses=0)
            cls.plans[key] = plan
        arrs = [np.array(F[k], dtype=np.int32, copy=True) for k in ("Lp", "Li", "Up", "Ui", "perm_r", "perm_c")]
        Ap = np.array(Ac.indptr, dtype=np.int32, copy=True); Ai = np.array(Ac.indices, dtype=np.int32, copy=True)
        n = int(F["n"])

        def build():
            h = c_vp()
            rc = lib.nep_lu_refac_create(lu.h, n, *[hptr(a) for a in arrs], hptr(Ap), hptr(Ai), C.byref(h))   # GIL released
            with cls.lock:
                if rc == 0:
                    plan["handle"] = h; plan["state"] = "ready"
                else:
                    plan["state"] = "off"
            plan.pop("keep", None)
        if "NEP_LU_PLAN_THREADS" not in os.environ:
            # enumeration threads from the CPU budget of this rank (8 ranks on a 16-CPU quota must not start 48 threads); through a
            # setter: writing the environment from here would race with getenv calls of plan threads already running
            from ._affinity import cpu_budget
            check(lib.nep_lu_set_plan_threads(max(1, min(6, cpu_budget() - 1))))
        plan["keep"] = lu                  # the reference factor must outlive the build
        t = threading.Thread(target=build, name="nep-lu-refac-plan", daemon=True)
        plan["thread"] = t
        t.start()

    @classmethod
    def factor_batch(cls, plan, n, vals, expected_solves=1, growth=None):
        """vals: (B, nnz) values of B matrices of the plan's pattern.  Returns a list of DeviceLU (None where the stored pivot
        sequence was refused for that matrix -- the caller factorises those on the host)."""
        B = int(vals.shape[0])
        vals = np.ascontiguousarray(vals, dtype=np.complex128)
        health = np.zeros((B, 3))
        outs = (c_vp * B)()
        check(lib.nep_lu_set_expected_solves(int(expected_solves)))
        check(lib.nep_lu_factor_dev_batch(plan["handle"], B, hptr(vals), int(expected_solves),
                                          cls.GROWTH if growth is None else float(growth), hptr(health), None, outs,
                                          stream_ptr()))
        res = []
        for b in range(B):
            if not outs[b]:
                plan["fails"] += 1
                res.append(None)
                continue
            plan["uses"] += 1
            res.append(DeviceLU._from_handle(c_vp(outs[b]), n, float(np.linalg.norm(vals[b])),
                                             dict(plan["strategy"], numeric="device (stored pivot sequence, batched)"),
                                             float(max(health[b, 1], health[b, 2]))))
        return res

    @classmethod
    def factor_batch_terms(cls, plan, n, D_dev, Cf, normA, expected_solves=1, growth=None):
        """the same for B matrices A_b = sum_t Cf[b, t] A_t whose term values sit on the device (D_dev: nnz x m_t, the union
        pattern of the plan): nothing of size B x nnz is formed on the host or uploaded.  normA: the B Frobenius norms."""
        Cf = np.ascontiguousarray(Cf, dtype=np.complex128)
        B, mt = Cf.shape
        assert D_dev.is_contiguous() and D_dev.shape[1] == mt
        health = np.zeros((B, 3))
        outs = (c_vp * B)()
        check(lib.nep_lu_set_expected_solves(int(expected_solves)))
        check(lib.nep_lu_factor_dev_batch_terms(plan["handle"], B, c_vp(D_dev.data_ptr()), mt, hptr(Cf), int(expected_solves),
                                                cls.GROWTH if growth is None else float(growth), hptr(health), outs, stream_ptr()))
        res = []
        for b in range(B):
            if not outs[b]:
                plan["fails"] += 1
                res.append(None)
                continue
            plan["uses"] += 1
            res.append(DeviceLU._from_handle(c_vp(outs[b]), n, float(normA[b]),
                                             dict(plan["strategy"], numeric="device (stored pivot sequence, batched)"),
                                             float(max(health[b, 1], health[b, 2]))))
        return res

    @classmethod
    def wait(cls):
        """block until every plan under construction is finished (tests, benchmarks)"""
        for p in list(cls.plans.values()):
            t = p.get("thread")
            if t is not None:
                t.join()

    @classmethod
    def clear(cls):
        cls.wait()
        with cls.lock:
            for p in cls.plans.values():
                if p.get("handle"):
                    lib.nep_lu_refac_destroy(p["handle"])
            cls.plans.clear()


atexit.register(_DeviceRefactor.wait)      # a plan thread inside the HIP runtime must not be cut off by interpreter shutdown


class DeviceLU:
    """nep_lu handle built from a host sparse LU of A (CSC/CSR/dense).

    Pivoting strategy mirrors what UMFPACK (the reference's `factorize`, LinSolvers.jl:116) selects
    automatically: for a structurally symmetric matrix with a zero-free diagonal the *symmetric strategy*
    (fill-reducing ordering of A+A', diagonal pivots preferred with tolerance 0.001) -> SuperLU with
    permc_spec=MMD_AT_PLUS_A, SymmetricMode, diag_pivot_thresh=0.001; otherwise the unsymmetric strategy
    (column ordering, threshold partial pivoting) -> COLAMD with SuperLU's default threshold.  Explicit
    arguments override the choice.  `factors` (the dict returned by _nep_hostlu.factor, e.g. from a HostLUPool
    worker) skips the host factorisation."""

    def __init__(self, A=None, permc_spec=None, diag_pivot_thresh=None, symmetric_mode=None, expected_solves=50,
                 factors=None, plan_pattern=None):
        """plan_pattern (with `factors`): the csc pattern these host factors belong to -- lets a factorisation that was
        computed elsewhere (contour_beyn's worker processes) seed the pattern's device-factorisation plan"""
        _lib.require_gpu()
        t0 = time.perf_counter()
        self.device_factorized = False
        rkey = None; Ac = None
        if factors is not None and plan_pattern is not None and _DeviceRefactor.enabled():
            k0 = _DeviceRefactor.key(plan_pattern, (permc_spec, diag_pivot_thresh, symmetric_mode))
            with _DeviceRefactor.lock:
                known = k0 in _DeviceRefactor.plans
            if not known:
                rkey = k0; Ac = plan_pattern
        if factors is None:
            Ac = sp.csc_matrix(A, dtype=np.complex128)
            if _DeviceRefactor.enabled():
                rkey = _DeviceRefactor.key(Ac, (permc_spec, diag_pivot_thresh, symmetric_mode))
                plan = _DeviceRefactor.lookup(rkey)
                if plan is not None and self._factor_on_device(plan, Ac, expected_solves, t0):
                    return
            try:
                factors = _nep_hostlu.factor(Ac.data, Ac.indices, Ac.indptr, Ac.shape, permc_spec=permc_spec,
                                             diag_pivot_thresh=diag_pivot_thresh, symmetric_mode=symmetric_mode)
            except RuntimeError as e:  # "Factor is exactly singular"
                raise np.linalg.LinAlgError("SingularException: " + str(e))
        F = factors
        shm_backed = "shm_name" in F
        if shm_backed:
            F = _nep_hostlu.attach_shm(F)
        n = int(F["n"])
        self.n = n
        self.normA = F["normA"]          # ||A||_F, used by the refinement stopping test
        self.strategy = F["strategy"]
        self.t_factor = F["t_factor"]
        t_a = time.perf_counter()
        Lp, Li, Lx, Up, Ui, Ux, pr, pc = (F[k] for k in ("Lp", "Li", "Lx", "Up", "Ui", "Ux", "perm_r", "perm_c"))
        self.t_convert = time.perf_counter() - t_a
        t_b = time.perf_counter()
        check(lib.nep_lu_set_expected_solves(int(expected_solves)))
        h = c_vp()
        try:
            create = lib.nep_lu_create_csc if F.get("fmt", "csr") == "csc" else lib.nep_lu_create
            check(create(n, hptr(Lp), hptr(Li), hptr(Lx), hptr(Up), hptr(Ui), hptr(Ux), hptr(pr), hptr(pc), C.byref(h)))
        finally:
            Fplan = F
            if shm_backed:
                if rkey is not None:        # the plan builder reads the index arrays after the shared block is gone
                    Fplan = {k: (np.array(F[k], copy=True) if k in ("Lp", "Li", "Up", "Ui", "perm_r", "perm_c") else F[k])
                             for k in ("Lp", "Li", "Up", "Ui", "perm_r", "perm_c", "fmt", "strategy", "n") if k in F}
                del Lp, Li, Lx, Up, Ui, Ux, pr, pc
                _nep_hostlu.release_shm(F)
        self.h = h
        self.t_create = time.perf_counter() - t_b
        self._describe()
        self.t_setup = time.perf_counter() - t0
        if rkey is not None:
            _DeviceRefactor.maybe_start(rkey, self, Fplan, Ac)

    @classmethod
    def _from_handle(cls, h, n, normA, strategy, growth):
        """wrap an nep_lu handle produced by the device-side numeric factorisation"""
        self = cls.__new__(cls)
        self.device_factorized = True
        self.growth = growth
        self.n = int(n); self.normA = normA; self.strategy = strategy
        self.t_factor = 0.0; self.t_convert = 0.0; self.t_create = 0.0; self.t_setup = 0.0
        self.h = h
        self._describe()
        return self

    def _factor_on_device(self, plan, Ac, expected_solves, t0):
        """numeric factorisation on the GPU with the pattern's stored pivot sequence; False -> the caller takes the host path"""
        health = np.zeros(3)
        h = c_vp()
        Ax = np.ascontiguousarray(Ac.data, dtype=np.complex128)
        check(lib.nep_lu_set_expected_solves(int(expected_solves)))
        rc = lib.nep_lu_factor_dev(plan["handle"], hptr(Ax), int(expected_solves), _DeviceRefactor.GROWTH, hptr(health), None,
                                   C.byref(h), stream_ptr())
        if rc != 0:
            plan["fails"] += 1
            if rc != _lib.NEP_ERR_SINGULAR or plan["fails"] >= 3:       # repeated breakdowns: the pivot sequence does not suit
                plan["state"] = "off"
            return False
        plan["uses"] += 1
        self.device_factorized = True
        self.growth = float(max(health[1], health[2]))       # max |L| and the element growth max|U| / max|A|
        self.n = int(Ac.shape[0])
        self.normA = float(np.linalg.norm(Ax))
        self.strategy = dict(plan["strategy"], numeric="device (stored pivot sequence)")
        self.t_factor = time.perf_counter() - t0
        self.t_convert = 0.0; self.t_create = 0.0
        self.h = h
        self._describe()
        self.t_setup = time.perf_counter() - t0
        return True

    def _describe(self):
        n = self.n
        info = (c_i64 * 6)()
        check(lib.nep_lu_info(self.h, info))
        self.nnzL, self.nnzU, self.levL, self.levU, self.solve_bytes = (int(info[1]), int(info[2]), int(info[3]),
                                                                        int(info[4]), int(info[5]))
        sch = (c_i64 * 8)()
        check(lib.nep_lu_schedule(self.h, sch))
        self.tail, self.levL_full, self.levU_full = int(sch[0]), int(sch[2]), int(sch[3])
        self.wide_segments, self.narrow_segments = int(sch[4]), int(sch[5])
        self.mid_rows, self.mid_block = int(sch[6]), int(sch[7])
        ib = C.c_int32(0)
        check(lib.nep_lu_is_block_schedule(self.h, C.byref(ib)))
        self.block_schedule = bool(ib.value)          # elimination-tree block schedule (trsv_ml.hip) vs level schedule
        if self.block_schedule:
            self.levels, self.split_levels, self.blocks = int(sch[2]), int(sch[4]), int(sch[5])
        # (nnz L + nnz U)(16 + 4) + 8(n + 1) + 3*16 n for one right-hand side
        self.algorithmic_bytes = (self.nnzL + self.nnzU) * 20 + 8 * (n + 1) + 48 * n

    def refactor(self, Lx, Ux):
        """same-pattern refactorisation (nep_lu_refactor): new values in the entry order of the factors this handle
        was created with"""
        Lx = np.ascontiguousarray(Lx, dtype=np.complex128); Ux = np.ascontiguousarray(Ux, dtype=np.complex128)
        assert Lx.shape[0] == self.nnzL and Ux.shape[0] == self.nnzU
        check(lib.nep_lu_refactor(self.h, hptr(Lx), hptr(Ux)))

    def set_row_scale(self, rs):
        rs = None if rs is None else np.ascontiguousarray(rs, dtype=np.float64)
        check(lib.nep_lu_set_row_scale(self.h, hptr(rs) if rs is not None else None))

    def launches_last_solve(self):
        sch = (c_i64 * 8)()
        check(lib.nep_lu_schedule(self.h, sch))
        return int(sch[1])

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.nep_lu_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self, B, out=None, scale=1.0):
        """B: device tensor (nrhs, n) (column-major n x nrhs block). Returns same-shaped tensor."""
        Bd = B if B.dim() == 2 else B.reshape(1, -1)
        nrhs, n = Bd.shape
        assert n == self.n
        X = torch.empty_like(Bd) if out is None else out
        check(lib.nep_lu_solve(self.h, nrhs, c_vp(Bd.data_ptr()), n, c_vp(X.data_ptr()), n, float(scale), stream_ptr()))
        return X.reshape(B.shape)

    def solve_add(self, B, add, out, scale=1.0):
        """out = scale * (add + A^{-1} B)  (refinement update; out may alias add)"""
        Bd = B if B.dim() == 2 else B.reshape(1, -1)
        nrhs, n = Bd.shape
        assert n == self.n
        check(lib.nep_lu_solve_add(self.h, nrhs, c_vp(Bd.data_ptr()), n, c_vp(add.data_ptr()), n, c_vp(out.data_ptr()), n,
                                   float(scale), stream_ptr()))
        return out


EPS = np.finfo(float).eps


def seed_plan_from_rank0(nep, lam, group=None, permc_spec=None, wait=True):
    """Collective over the ranks of a node (torch.distributed): the FIRST factorisation of a sparsity pattern -- the host SuperLU
    run that fixes ordering, pivot sequence and fill, and from which the pattern's device-factorisation plan is built -- is done
    by rank 0 alone and its factor arrays are broadcast; every rank uploads them, starts its plan (enumerated on its own GPU,
    21 ms for gun) and, with wait=True, returns when the plan is ready, so that no rank ever runs SuperLU for this pattern (8
    ranks on a shared CPU quota would otherwise start 8 concurrent host factorisations in their first call).  Returns the
    DeviceLU of M(lam) built from the broadcast factors.  Without an initialised process group: the plain local path."""
    import torch.distributed as dist
    A = sp.csc_matrix(nep.compute_Mder(lam), dtype=np.complex128)
    A.sort_indices()
    distd = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distd else 0
    F = None
    if rank == 0:
        F = _nep_hostlu.factor(A.data, A.indices, A.indptr, A.shape, permc_spec=permc_spec, diag_pivot_thresh=None, symmetric_mode=None)
    if distd:
        box = [F]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        F = box[0]
    seed_plan_from_rank0.host_factorisations += 1 if rank == 0 else 0
    lu = DeviceLU(factors=F, plan_pattern=A, permc_spec=permc_spec)
    if wait:
        _DeviceRefactor.wait()
    return lu


seed_plan_from_rank0.host_factorisations = 0


class FactorizeLinSolver(LinSolver):
    """src/LinSolvers.jl:109-137: factor M(lam) once, solve many right-hand sides.  Like UMFPACK's solve
    (control[8] = umfpack_refinements, LinSolvers.jl:118-120) each single-vector solve is followed by iterative
    refinement with UMFPACK's stopping rule: r = b - M(lam) x (kernel K1 through the NEP), componentwise backward error
    omega = max_i |r_i| / (|M||x| + |b|)_i (nep_cw_backward_error; |M| is bounded by sum_i |f_i(lam)| |A_i|); stop when
    omega < eps, when omega stops halving (reverting a step that made it worse) or after umfpack_refinements steps.
    The device factors use explicitly inverted diagonal blocks (trsv.hip), whose raw solves are a little less accurate
    than a plain substitution; the refinement makes the result independent of that."""

    def __init__(self, nep, lam, umfpack_refinements=10, permc_spec=None, _lu=None, **lu_kw):
        self.nep = nep
        self.lam = lam
        self.umfpack_refinements = umfpack_refinements
        lu_kw.setdefault("expected_solves", 200)      # a FactorizeLinSolver exists to be reused (iar/tiar: maxit solves)
        if _lu is None:
            _lu = self._lu_from_terms(nep, lam, permc_spec, lu_kw)
        self.lu = _lu if _lu is not None else DeviceLU(nep.compute_Mder(lam), permc_spec=permc_spec, **lu_kw)
        self.refine_steps_taken = 0
        self.refine_checks = 0
        self.last_omega = None
        self._plan = 0                  # refinement steps the last checked solve needed
        self._stable = 0                # consecutive checked solves that needed exactly _plan steps
        self._recorded_plan = None      # sweeps the stopping rule asked for on the last reviewed record (nep_iar_step)
        self.solves = 0
        self._W = None
        self._cabs = None
        self._cf = None

    @staticmethod
    def _lu_from_terms(nep, lam, permc_spec, lu_kw):
        """M(lam) = sum_t f_t(lam) A_t of a pure SPMF NEP whose pattern has a device-factorisation plan: the m_t coefficients go
        to the device, the values are assembled there (k_lu_init_terms) and factorised with the stored pivot sequence -- no
        compute_Mder on the host, no value upload, no pattern hash (1 ms of the 4.4 ms a gun linear solver took).  None: the
        caller takes the general route (no plan yet, other options, a refusal)."""
        if (permc_spec is not None or set(lu_kw) - {"expected_solves"} or not _DeviceRefactor.enabled()
                or os.environ.get("NEP_LU_TERMS", "1") == "0"):
            return None
        from .nep import AbstractSPMF
        if not (isinstance(nep, AbstractSPMF) and type(nep).compute_Mder is AbstractSPMF.compute_Mder
                and hasattr(nep, "aligned_terms_dev")):
            return None
        if not _DeviceRefactor.plans:                      # nothing planned yet: do not build the device term block for it
            return None
        al = nep.aligned_terms_dev()
        if al is None:
            return None
        indptr, indices, D_dev, G = al

        class _Pattern:
            pass
        A0 = _Pattern(); A0.indptr = indptr; A0.indices = indices; A0.shape = (nep.n, nep.n)
        plan = _DeviceRefactor.lookup(_DeviceRefactor.key(A0, (None, None, None)))
        if plan is None:
            return None
        Cf = np.array([[f.derivs(lam, 1)[0] for f in nep.get_fv()]], dtype=np.complex128)
        normA = np.sqrt(np.maximum(np.einsum("bs,st,bt->b", Cf.conj(), G, Cf).real, 0.0))
        lu = _DeviceRefactor.factor_batch_terms(plan, nep.n, D_dev, Cf, normA, expected_solves=int(lu_kw.get("expected_solves", 200)))[0]
        if lu is not None:
            lu.strategy = dict(plan["strategy"], numeric="device (stored pivot sequence)")
        return lu

    def _refine_setup(self):
        if self._W is None:
            n = self.lu.n
            self._W = torch.empty((4, n), dtype=CDT, device="cuda")                      # r, x, b, dx
            nep = self.nep
            if hasattr(nep, "get_fv") and hasattr(nep, "dev"):
                cf = np.ascontiguousarray([f.derivs(self.lam, 1)[0] for f in nep.get_fv()], dtype=np.complex128)
                self._cabs = np.ascontiguousarray(np.abs(cf))
                self._spmf = nep.dev.h
                # pure SPMF operator (compute_Mlincomb not overridden): M x is formed inside the criterion kernel
                from .nep import AbstractSPMF
                self._cf = cf if type(nep).compute_Mlincomb is AbstractSPMF.compute_Mlincomb else None

    def _residual(self, b, want_omega):
        """W[0] = b - M(lam) x  (x = W[1]); returns the backward error (None if not requested)."""
        W = self._W
        n = self.lu.n
        from . import dense
        if self._cabs is not None:
            om = np.zeros(1)
            fused = self._cf is not None
            Mx = None if fused else self.nep.compute_Mlincomb(self.lam, W[1].reshape(1, n))
            extra = None
            if want_omega and hasattr(self.nep, "refine_denominator_extra"):
                extra = self.nep.refine_denominator_extra(self.lam, W[1])          # (|P||x|) of non-SPMF operator parts
            check(lib.nep_cw_backward_error(self._spmf, hptr(self._cabs), hptr(self._cf) if fused else None,
                                            c_vp(W[1].data_ptr()), c_vp(b.data_ptr()),
                                            None if fused else c_vp(Mx.data_ptr()),
                                            c_vp(extra.data_ptr()) if extra is not None else None, c_vp(W[0].data_ptr()),
                                            hptr(om) if want_omega else None, stream_ptr()))
            return float(om[0]) if want_omega else None
        # NEP without an SPMF device handle: normwise backward error ||r|| / (||M||_F ||x|| + ||b||)
        Mx = self.nep.compute_Mlincomb(self.lam, W[1].reshape(1, n))
        dense.copy(b, W[2], n)
        dense.copy(Mx, W[0], n)
        dense.scal(W[0], -1.0, n)
        dense.axpy(1.0, W[2], W[0], n)
        if not want_omega:
            return None
        nr = np.empty(3)
        check(lib.nep_colnorms(n, 3, c_vp(W.data_ptr()), n, hptr(nr), stream_ptr()))
        return nr[0] / (self.lu.normA * nr[1] + nr[2]) if (nr[1] > 0 or nr[2] > 0) else 0.0

    def blind_plan(self):
        """refinement steps of the NEXT single-vector solve if it can be issued without reading omega back (None = that solve
        evaluates the criterion on the host).  Used by drivers that hand a whole step to the library (nep_iar_step)."""
        if self.umfpack_refinements <= 0:
            return 0
        if self._stable >= 4 and ((self.solves + 1) % 8) != 0:
            return self._plan
        return None

    _omega_log = None

    def blind_plan_recorded(self):
        """refinement sweeps of the next solve of a step that RECORDS omega of every iterate (nep_iar_step): the settled
        count once a record has been reviewed, the maximum before"""
        if self.umfpack_refinements <= 0:
            return 0
        # the record holds omega of x_0..x_3; two sweeps to start with (gun needs one), a rule that asks for more is a miss.
        # A solver of a NEP whose previous solver settled on a count starts with that count (nep._refine_hint: iar issues
        # all its steps before the first record is reviewed, so without the hint every step of every run takes the two
        # sweeps; gun: omega(x_1) <= 1.7 eps in every step of every run, one sweep, 46 us less per step); a wrong hint is a
        # miss like any other, and a miss withdraws it.  NEP_REFINE_HINT=0 turns it off.
        if self._recorded_plan is not None:
            return self._recorded_plan
        hint = self._hint()
        return min(self.umfpack_refinements, 2 if hint is None else hint)

    def _hint(self):
        """the sweep count a previous solver of this NEP settled on AT THIS SHIFT (None otherwise): conditioning and pivot growth
        of M(sigma) change with sigma, so a count learnt at another shift says nothing about this factorisation"""
        if os.environ.get("NEP_REFINE_HINT", "1") == "0":
            return None
        nep = getattr(self, "nep", None)
        h = getattr(nep, "_refine_hint", None)
        if h is None or getattr(nep, "_refine_hint_lam", None) != self._lam_key():
            return None
        return h

    def _lam_key(self):
        lam = getattr(self, "lam", None)
        return None if lam is None else complex(lam)

    def settled_plan(self):
        """True when the refinement count of this solver's NEP has settled (a reviewed record of this solver, or the count a
        previous solver of the NEP settled on): steps may then skip the record of the kept iterate 7 times out of 8"""
        if os.environ.get("NEP_IAR_RECORD_ALL"):
            return False
        if self._recorded_plan is not None:
            return True
        return self._hint() is not None

    def review_recorded(self, w, plan, final_recorded=True):
        """UMFPACK's stopping rule (the loop of solve_dev) replayed on the recorded omegas w[0..plan] of a solve that took
        `plan` sweeps without looking.  True: what was returned is what the checked loop returns, or an iterate at least as
        good; False: the checked loop would have continued, or would have taken a worsening sweep back.
        final_recorded=False: omega of the kept iterate x_plan was not evaluated (a step that took it on trust, 7 of 8 once
        the count has settled); the rule is replayed on x_0..x_{plan-1} and must not have stopped there."""
        umf = self.umfpack_refinements
        if not final_recorded:
            w_prev = np.inf
            for step in range(plan):
                omega = float(w[step])
                if np.isfinite(omega) and omega <= 2.0 * EPS:
                    # the checked loop would have stopped at x_step, which satisfies UMFPACK's criterion already; the sweeps taken
                    # beyond it start from a converged iterate (a sweep from there moves x by O(eps) |x|): accepted, and the
                    # following steps plan that many sweeps -- not a miss (a miss re-runs the whole call unfused)
                    self._recorded_plan = max(step, 1)
                    return True
                if not np.isfinite(omega) or omega > 0.5 * w_prev:
                    self._note_hint(None)          # the checked loop would have stopped before the sweeps that were taken
                    return False
                w_prev = omega
            return True
        w = [float(x) for x in w[:plan + 1]]
        if FactorizeLinSolver._omega_log is not None:       # diagnostics (scripts/diag/iar_omega_log.py)
            FactorizeLinSolver._omega_log.append(w)
        w_prev = np.inf; ret = None
        for step in range(umf + 1):
            if step > plan:
                # the rule would go on.  At the noise level of omega itself (its own evaluation carries a few eps of
                # round-off: 4.49e-16 observed on gun against the 4.44e-16 threshold) a further sweep cannot improve the
                # iterate -- accepted; anything larger is a miss
                if np.isfinite(w[plan]) and w[plan] <= 4.0 * EPS:
                    ret = plan
                    break
                self._note_hint(None)
                return False
            omega = w[step]
            if omega <= 2.0 * EPS:
                ret = step
                break
            if omega > 0.5 * w_prev:
                ret = step - 1 if omega > w_prev else step
                break
            if step == umf:
                ret = step
                break
            w_prev = omega
        self.last_omega = w[plan]
        # never plan fewer than one sweep: the accuracy of the raw solve changes within a run when the dense apex of the
        # block schedule comes on line (trsv_ml.hip: built behind the first solves), and a plan of 0 learnt before would
        # turn the first solve after it into a miss (a full re-run of the call)
        self._recorded_plan = max(ret, 1)
        ok = bool(np.isfinite(w[plan]) and (ret == plan or w[plan] <= max(4.0 * EPS, w[ret])))
        self._note_hint(self._recorded_plan if ok else None)
        return ok

    def _note_hint(self, plan):
        """plan = None: a miss -- the hint is withdrawn for good on this NEP object (a problem whose omega sits at the edge of the
        rule would otherwise alternate between learning the count and missing with it, every miss a re-run of the call)"""
        nep = getattr(self, "nep", None)
        if nep is not None:
            try:
                if plan is None:
                    nep._refine_hint = None
                    nep._refine_hint_off = True
                elif not getattr(nep, "_refine_hint_off", False):
                    nep._refine_hint = plan
                    nep._refine_hint_lam = self._lam_key()
            except AttributeError:
                pass

    def note_blind_solve(self, plan):
        self.solves += 1
        self.refine_steps_taken += plan

    def refine_coefficients(self):
        """(|f_t(lam)|, f_t(lam)) of a pure SPMF operator, or None when M x needs the NEP's own compute_Mlincomb"""
        self._refine_setup()
        return (self._cabs, self._cf) if (self._cabs is not None and self._cf is not None) else None

    def solve_dev(self, b, out=None, scale=1.0):
        """device solve; b: (n,) or (nrhs, n) tensor; out may alias b"""
        self.solves += 1
        single = b.dim() == 1 or b.shape[0] == 1
        if not single or self.umfpack_refinements <= 0 or not hasattr(self.nep, "compute_Mlincomb"):
            return self.lu.solve(b, out=out, scale=scale)
        from . import dense
        # Every evaluation of omega is a host synchronisation.  Once 4 consecutive checked solves needed the same
        # number of steps (same factors, same conditioning), 7 of 8 solves take that many steps blindly.
        blind = self._stable >= 4 and (self.solves % 8) != 0
        if blind and self._plan == 0:
            return self.lu.solve(b, out=out, scale=scale)
        self._refine_setup()
        n = self.lu.n
        W = self._W
        bd = b.reshape(1, n)
        x = W[1].reshape(1, n)
        r = W[0].reshape(1, n)
        X = torch.empty_like(b) if out is None else out
        self.lu.solve(bd, out=x)
        if blind:
            for i in range(self._plan):
                self._residual(bd, False)
                if i == self._plan - 1:
                    self.lu.solve_add(r, x, X, scale)       # X = scale (x + A^{-1} r); b is not needed any more
                else:
                    self.lu.solve_add(r, x, x)
                self.refine_steps_taken += 1
            return X.reshape(b.shape)
        self.refine_checks += 1
        steps = 0
        w_prev = np.inf
        for step in range(self.umfpack_refinements + 1):
            omega = self._residual(bd, True)
            if omega <= 2.0 * EPS:                       # UMFPACK: omega < eps.  The computed omega carries round-off of
                break                                    # its own; a step taken at (eps, 2 eps] cannot halve it
            if omega > 0.5 * w_prev:
                if omega > w_prev:                       # the last step made it worse: take it back
                    dense.axpy(-1.0, W[3], W[1], n)
                    omega = w_prev
                break
            if step == self.umfpack_refinements:
                break
            w_prev = omega
            self.lu.solve(r, out=W[3].reshape(1, n))
            dense.axpy(1.0, W[3], W[1], n)
            steps += 1
        self.last_omega = omega
        self.refine_steps_taken += steps
        if steps == self._plan:
            self._stable += 1
        else:
            self._plan, self._stable = steps, 0
        dense.copy(W[1], X, n)
        if scale != 1.0:
            dense.scal(X, scale, n)
        return X.reshape(b.shape)


class BackslashLinSolver(LinSolver):
    """src/LinSolvers.jl:147-159: `A \\ x`, i.e. a fresh factorisation at every lin_solve."""

    def __init__(self, nep, lam, permc_spec=None, **lu_kw):
        self.A = nep.compute_Mder(lam)
        self.permc_spec = permc_spec
        self.lu_kw = lu_kw

    def solve_dev(self, b, out=None, scale=1.0):
        lu = DeviceLU(self.A, permc_spec=self.permc_spec, expected_solves=1, **self.lu_kw)
        self.last_lu = lu
        return lu.solve(b, out=out, scale=scale)


def lin_solve(solver, b, tol=0, scale=1.0):
    """src/LinSolvers.jl:125-137.  b: NumPy vector/matrix (-> NumPy result) or device tensor
    (nrhs, n) / (n,) (-> device result).  `scale` multiplies the result on the device."""
    host = not is_dev(b)
    bd = to_dev(b) if host else b
    if isinstance(solver, GMRESLinSolver) or getattr(solver, "accepts_tol", False):
        x = solver.solve_dev(bd, scale=scale, tol=tol)          # `tol` is a hint for iterative solvers (LinSolvers.jl:184)
    else:
        x = solver.solve_dev(bd, scale=scale)
    if host:
        xh = to_host(x if x.dim() == 2 else x.reshape(1, -1))
        return xh[:, 0] if np.ndim(b) == 1 else xh
    return x


class LinSolverCreator:
    pass


class FactorizeLinSolverCreator(LinSolverCreator):
    """src/LinSolverCreators.jl:62-122 (factorisation recycling keyed by lam)."""

    def __init__(self, umfpack_refinements=10, max_factorizations=0, nep=None, precomp_values=(),
                 permc_spec=None, **lu_kw):
        if np.isscalar(precomp_values):
            precomp_values = [precomp_values]
        if len(precomp_values) > 0 and nep is None:
            raise ValueError("When you want to precompute factorizations you need to supply the keyword "
                             "argument `nep`")
        self.umfpack_refinements = umfpack_refinements
        self.max_factorizations = max_factorizations
        self.permc_spec = permc_spec
        self.lu_kw = lu_kw
        self.recycled_factorizations = {}
        for s in precomp_values:
            self.recycled_factorizations[complex(s)] = DeviceLU(nep.compute_Mder(s), permc_spec=permc_spec, **lu_kw)


class BackslashLinSolverCreator(LinSolverCreator):
    """`workers`: host processes used by drivers that know several shifts in advance (contour_beyn); None = auto,
    0 = factor in-process one node at a time."""

    def __init__(self, permc_spec=None, workers=None, **lu_kw):
        self.permc_spec = permc_spec
        self.workers = workers
        self.lu_kw = lu_kw


DefaultLinSolverCreator = FactorizeLinSolverCreator


def create_linsolver(creator, nep, lam):
    """src/LinSolverCreators.jl:107-122,143."""
    if hasattr(creator, "create_linsolver"):            # problem-specific creators (WEPLinSolverCreator, Waveguide.jl:504-519)
        return creator.create_linsolver(nep, lam)
    if isinstance(creator, BackslashLinSolverCreator):
        return BackslashLinSolver(nep, lam, creator.permc_spec, **creator.lu_kw)
    if isinstance(creator, GMRESLinSolverCreator):
        return GMRESLinSolver(nep, lam, creator.kwargs)
    key = complex(lam)
    if key in creator.recycled_factorizations:
        return FactorizeLinSolver(nep, lam, creator.umfpack_refinements, _lu=creator.recycled_factorizations[key])
    solver = FactorizeLinSolver(nep, lam, creator.umfpack_refinements, permc_spec=creator.permc_spec, **creator.lu_kw)
    if len(creator.recycled_factorizations) < creator.max_factorizations:
        creator.recycled_factorizations[key] = solver.lu
    return solver


class LinSolverCache:
    """src/rk_helper/linsolvercache.jl:7-26.

    `prefetch(shifts)`: a driver that knows its next shifts (nleigs: the node sequence sigma) announces them; their host
    factorisations (compute_Mder + SuperLU, which releases the GIL) then run on one background thread while the device
    works on the current step, and `_get` only builds the device schedule from the finished factors.  Only for the
    factorising creator; at most `ahead` factorisations are in flight or waiting to be used."""

    def __init__(self, nep, linsolvercreator, ahead=2):
        self.nep = nep
        self.linsolvercreator = linsolvercreator
        self.solvers = {}
        self.ahead = ahead
        self._pending = {}
        self._pool = None

    def _host_factors(self, sigma):
        c = self.linsolvercreator
        Ac = sp.csc_matrix(self.nep.compute_Mder(sigma), dtype=np.complex128)
        kw = {k: v for k, v in c.lu_kw.items() if k in ("diag_pivot_thresh", "symmetric_mode")}
        return _nep_hostlu.factor(Ac.data, Ac.indices, Ac.indptr, Ac.shape, permc_spec=c.permc_spec, **kw)

    def _device_plan_ready(self):
        """True when the sparsity pattern of this NEP's M(sigma) has a device-factorisation plan: the shifts are then factorised
        on the GPU when they are needed (3.5 ms each on gun) and nothing is prefetched on the host"""
        c = self.linsolvercreator
        if type(c) is not FactorizeLinSolverCreator or not _DeviceRefactor.enabled():
            return False
        if getattr(self, "_plan_key", None) is None:
            try:
                A = sp.csc_matrix(self.nep.compute_Mder(0.5 + 0.25j), dtype=np.complex128)   # any shift: the pattern is what counts
            except Exception:
                self._plan_key = False
                return False
            kw = c.lu_kw
            self._plan_key = _DeviceRefactor.key(A, (c.permc_spec, kw.get("diag_pivot_thresh"), kw.get("symmetric_mode")))

            class _Pattern:         # what _DeviceRefactor.key / maybe_start read of a csc matrix (no values kept)
                pass
            self._pattern = _Pattern(); self._pattern.indptr = A.indptr; self._pattern.indices = A.indices; self._pattern.shape = A.shape
        return self._plan_key is not False and _DeviceRefactor.lookup(self._plan_key) is not None

    def prefetch(self, shifts, keep_all=None):
        """`keep_all` (optional): every distinct shift the driver WILL use and keep (nleigs with reusefact = 2).  When the pattern has a
        device-factorisation plan they are factorised in ONE batched pass (nep_lu_factor_dev_batch: each of the ~600 launches of the
        numeric factorisation carries all of them, and their solve schedules are built together) -- config C3: five shifts, 96 -> 88 ms."""
        c = self.linsolvercreator
        if type(c) is not FactorizeLinSolverCreator or os.environ.get("NEP_LU_PREFETCH", "1") == "0":
            return
        if self._device_plan_ready():
            if keep_all is not None and not getattr(self, "_batched", False) and os.environ.get("NEP_LU_CACHE_BATCH", "1") != "0":
                self._batched = True
                todo = [complex(s) for s in keep_all if np.isfinite(complex(s)) and complex(s) not in self.solvers
                        and complex(s) not in c.recycled_factorizations]
                plan = _DeviceRefactor.lookup(self._plan_key)
                if len(todo) >= 2 and plan is not None:
                    mats = [sp.csc_matrix(self.nep.compute_Mder(k), dtype=np.complex128) for k in todo]
                    P = self._pattern
                    if all(M.shape == P.shape and np.array_equal(M.indptr, P.indptr) and np.array_equal(M.indices, P.indices) for M in mats):
                        lu_kw = dict(c.lu_kw); lu_kw.setdefault("expected_solves", 200)
                        lus = _DeviceRefactor.factor_batch(plan, mats[0].shape[0], np.stack([M.data for M in mats]),
                                                           expected_solves=lu_kw["expected_solves"])
                        for key, lu in zip(todo, lus):
                            if lu is not None:          # (a refused one is factorised on its own when it is needed)
                                self.solvers[key] = FactorizeLinSolver(self.nep, key, c.umfpack_refinements, _lu=lu)
            return
        for s in shifts:
            key = complex(s)
            if len(self._pending) >= self.ahead:
                break
            if not np.isfinite(key) or key in self.solvers or key in self._pending or key in c.recycled_factorizations:
                continue
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="nep-lu-prefetch")
            self._pending[key] = self._pool.submit(self._host_factors, key)

    def _get(self, sigma, add_to_cache):
        key = complex(sigma)
        if key in self.solvers:
            return self.solvers[key]
        c = self.linsolvercreator
        if key in self._pending:
            try:
                factors = self._pending.pop(key).result()
            except RuntimeError as e:  # "Factor is exactly singular"
                raise np.linalg.LinAlgError("SingularException: " + str(e))
            lu_kw = dict(c.lu_kw); lu_kw.setdefault("expected_solves", 200)
            # (the prefetched host factors seed the pattern's device-factorisation plan like a factorisation made in place would:
            # a process that only runs nleigs factorises its shifts on the GPU from the second call on)
            lu = DeviceLU(factors=factors, permc_spec=c.permc_spec, diag_pivot_thresh=lu_kw.get("diag_pivot_thresh"),
                          symmetric_mode=lu_kw.get("symmetric_mode"), plan_pattern=getattr(self, "_pattern", None),
                          **{k: v for k, v in lu_kw.items() if k == "expected_solves"})
            solver = FactorizeLinSolver(self.nep, sigma, c.umfpack_refinements, _lu=lu)
            if len(c.recycled_factorizations) < c.max_factorizations:
                c.recycled_factorizations[key] = lu
        elif self._pool is not None and not self._device_plan_ready():
            # keep host factorisations on the one worker thread (they toggle the process-wide BLAS thread count)
            self._pending[key] = self._pool.submit(self._host_factors, key)
            return self._get(sigma, add_to_cache)
        else:
            solver = create_linsolver(c, self.nep, sigma)         # DeviceLU(A): device factorisation when the plan exists
        if add_to_cache:
            self.solvers[key] = solver
        return solver

    def close(self):
        if self._pool is not None:
            for f in self._pending.values():
                f.cancel()
            self._pool.shutdown(wait=True)
            self._pool = None
            self._pending = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, sigma, y, add_to_cache):
        return lin_solve(self._get(sigma, add_to_cache), y)

    def solve_dev(self, sigma, y, add_to_cache, out=None, scale=1.0):
        return self._get(sigma, add_to_cache).solve_dev(y, out=out, scale=scale)


_GMRES_TRUE_RESIDUAL = bool(os.environ.get("NEP_GMRES_TRUE_RESIDUAL"))     # A/B: re-evaluate the residual after a converged cycle


class GMRESLinSolver(LinSolver):
    """src/LinSolvers.jl:171-188: matrix-free restarted GMRES on v -> compute_Mlincomb(nep, lam, v), i.e. every
    iteration is one K1 call (folded single-vector SpMV) + one K6 orthogonalisation (IterativeSolvers' default
    ModifiedGramSchmidt) on the device; only the (restart+1) x restart Hessenberg / Givens data live on the host.
    kwargs mirror IterativeSolvers.gmres!: Pl (left preconditioner: a vector/diagonal matrix d meaning Pl \\ r = r ./ d,
    or a callable acting on a device vector), reltol (alias tol), abstol, restart, maxiter."""

    def __init__(self, nep, lam, kwargs=None):
        kwargs = dict(kwargs or {})
        self.nep, self.lam = nep, lam
        self.n = nep.size(1)
        self.restart = int(kwargs.get("restart", min(20, self.n)))
        self.maxiter = int(kwargs.get("maxiter", self.n))
        self.reltol = kwargs.get("reltol", kwargs.get("tol", None))
        self.abstol = float(kwargs.get("abstol", 0.0))
        from . import dense as _d
        # IterativeSolvers' gmres orthogonalises with ModifiedGramSchmidt by default (orth_meth keyword)
        self.orth = {"mgs": _d.MGS, "cgs": _d.CGS, "dgks": _d.DGKS}[str(kwargs.get("orth_meth", os.environ.get("NEP_GMRES_ORTH", "mgs"))).lower()]
        Pl = kwargs.get("Pl", None)
        self._Pl_call = None
        self._Pl_inv = None
        if Pl is not None:
            if callable(Pl):
                self._Pl_call = Pl
            else:
                d = np.asarray(Pl) if np.ndim(Pl) == 1 else (Pl.diagonal() if hasattr(Pl, "diagonal") else np.diag(np.asarray(Pl)))
                self._Pl_inv = to_dev(1.0 / np.asarray(d, dtype=np.complex128))[0]
        self.iterations = 0
        self.fused_step = None          # optional callable (v, w): w = Pl^{-1} A v  (e.g. a captured hipGraph)

    def _prec(self, r):
        if self._Pl_inv is not None:
            check(lib.nep_hadamard(self.n, 1, c_vp(r.data_ptr()), self.n, c_vp(self._Pl_inv.data_ptr()), self.n, stream_ptr()))
        elif self._Pl_call is not None:
            out = self._Pl_call(r)
            if out is not None and out.data_ptr() != r.data_ptr():
                from . import dense
                dense.copy(out, r, self.n)
        return r

    def solve_dev(self, b, out=None, scale=1.0, tol=None):
        from . import dense
        n, m = self.n, self.restart
        if b.dim() == 2 and b.shape[0] > 1:
            X = torch.empty_like(b) if out is None else out
            for j in range(b.shape[0]):
                self.solve_dev(b[j], out=X[j], scale=scale, tol=tol)
            return X
        reltol = tol if tol else (self.reltol if self.reltol is not None else np.sqrt(np.finfo(float).eps))
        bvec = b.reshape(n)
        x = torch.zeros(n, dtype=CDT, device="cuda")
        V = torch.empty((m + 1, n), dtype=CDT, device="cuda")
        r = torch.empty(n, dtype=CDT, device="cuda")
        dense.copy(bvec, r, n); self._prec(r)
        beta = dense.nrm2(r)
        tolabs = max(reltol * beta, self.abstol)
        its = 0
        # Arnoldi columns are orthogonalised on the device without reading anything back (nep_orth_dev: DGKS / CGS); the
        # Givens update of column j runs on the host while the device already works on column j+1, so the only cost of the
        # convergence test is ONE speculative iteration at the end of a cycle instead of a device stall in every iteration
        # (waveguide, n = 1e6: 3300 stalls of 50-100 us per tiar run).  MGS keeps the step-synchronous loop.
        pipelined = self.orth in (dense.DGKS, dense.CGS) and not os.environ.get("NEP_GMRES_SYNC")
        if pipelined:
            Hdev = torch.zeros((m, m + 3), dtype=CDT, device="cuda")
            Hpin = torch.zeros((m, m + 3), dtype=CDT).pin_memory()
            Hnp = Hpin.numpy()
        while beta > tolabs and its < self.maxiter:
            dense.copy(r, V[0], n); dense.scal(V[0], 1.0 / beta, n)
            H = np.zeros((m + 1, m), dtype=np.complex128)
            cs = np.zeros(m, dtype=np.complex128); sn = np.zeros(m, dtype=np.complex128)
            g = np.zeros(m + 1, dtype=np.complex128); g[0] = beta
            j_done = 0

            def givens(j, h, hb):
                """column j of H (h: j+1 entries, hb: the subdiagonal) through the rotations; True = cycle finished"""
                nonlocal its, j_done
                H[:j + 1, j] = h; H[j + 1, j] = hb
                for i in range(j):                       # apply previous Givens rotations
                    t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                    H[i + 1, j] = -np.conj(sn[i]) * H[i, j] + np.conj(cs[i]) * H[i + 1, j]
                    H[i, j] = t
                a_, b_ = H[j, j], H[j + 1, j]
                d = np.sqrt(abs(a_) ** 2 + abs(b_) ** 2)
                cs[j] = a_ / d; sn[j] = b_ / d
                cs[j] = np.conj(cs[j]); sn[j] = np.conj(sn[j])
                H[j, j] = d; H[j + 1, j] = 0.0
                g[j + 1] = -np.conj(sn[j]) * g[j]
                g[j] = cs[j] * g[j]
                its += 1; j_done = j + 1
                return abs(g[j + 1]) <= tolabs or its >= self.maxiter

            def apply_op(j):
                w = V[j + 1]
                if self.fused_step is not None:            # w = Pl^{-1} A v as one pre-recorded launch sequence
                    self.fused_step(V[j], w)
                else:
                    dense.copy(self.nep.compute_Mlincomb(self.lam, V[j].reshape(1, n)), w, n)
                    self._prec(w)
                return w

            if pipelined:
                evs = [None] * m
                done = False
                for j in range(m):
                    w = apply_op(j)
                    dense.orthogonalize_and_normalize_dev(V, w, j + 1, Hdev[j], rows=n, ldv=n, method=self.orth)
                    Hpin[j, :j + 3].copy_(Hdev[j, :j + 3], non_blocking=True)
                    evs[j] = torch.cuda.Event(); evs[j].record()
                    if j >= 1:                             # column j-1 while the device computes column j
                        evs[j - 1].synchronize()
                        row = Hnp[j - 1]
                        if givens(j - 1, row[:j].copy(), row[j].real):
                            done = True
                            break
                if not done and j_done < m and its < self.maxiter:
                    jl = j_done
                    evs[jl].synchronize()
                    row = Hnp[jl]
                    givens(jl, row[:jl + 1].copy(), row[jl + 1].real)
            else:
                for j in range(m):
                    w = apply_op(j)
                    h, hb, _ = dense.orthogonalize_and_normalize(V, w, j + 1, rows=n, ldv=n, method=self.orth)
                    if givens(j, h, hb):
                        break
            y = np.linalg.solve(np.triu(H[:j_done, :j_done]), g[:j_done])
            dx = dense.gemm_ts(V, y.reshape(-1, 1), k=j_done, rows=n, ldz=n)          # (1, n)
            dense.axpy(1.0, dx, x, n)
            if abs(g[j_done]) <= tolabs and not _GMRES_TRUE_RESIDUAL:
                # converged by the recurrence's residual norm: IterativeSolvers.gmres leaves here too (it evaluates the true
                # residual only when it restarts), which saves one operator + preconditioner application per solve
                beta = abs(g[j_done])
                break
            # true (preconditioned) residual for the restart
            dense.copy(self.nep.compute_Mlincomb(self.lam, x.reshape(1, n)), r, n)
            dense.scal(r, -1.0, n); dense.axpy(1.0, bvec, r, n); self._prec(r)
            beta = dense.nrm2(r)
        self.iterations = its
        X = torch.empty_like(b) if out is None else out
        dense.copy(x, X, n)
        if scale != 1.0:
            dense.scal(X, scale, n)
        return X.reshape(b.shape)


class GMRESLinSolverCreator(LinSolverCreator):
    """src/LinSolverCreators.jl:124-145"""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
