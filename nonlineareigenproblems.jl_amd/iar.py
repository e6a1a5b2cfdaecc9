"""Infinite Arnoldi (iar) on the device backend -- same keyword surface as src/method_iar.jl:47-63.

Device-resident state: the basis V (n(m+1) x (m+1), 1.6 GB for gun m=100), z/y work vectors and the
Ritz block.  Per iteration only H's new column (k+1 numbers), the k x k eigenvector matrix of H, the
coefficient block and 2k norms cross PCIe.

Reference step (method_iar.jl:94-164)            device realisation
  y[:,2:k+1]=reshape(VV[1:nk,k],n,k)./(1:k)'      none: column k of V *is* an n x k block (ld n); the
                                                   1/j scaling is folded into the coefficient block
  y[:,1]=compute_Mlincomb!(nep,s,y,alpha)         K1 nep_mlincomb on that block
  y[:,1]=-lin_solve(M0inv,y[:,1])                 K5 nep_lu_solve(scale=-1) straight into V[0:n,k+1]
  vv=reshape(y[:,1:k+1])                          nep_iar_shift_scale (one pass over n*k entries)
  orthogonalize_and_normalize!(VV,vv,h,DGKS)      K6 nep_orth with the block-triangular row counts
  D,Z=eigen(H[1:k,1:k])                           host LAPACK (k x k)
  Q=VV[1:n,:]*Z                                   K7 nep_gemm_ts -> row-major Q^T
  err[k,s]=estimate_error(...) for s=1:k          K2 nep_resid_batch (one pass for all k pairs)
"""
import os
import threading
import time

import numpy as np
import torch

from . import dense, _hosteig
from ._lib import lib, check, c_vp, c_i32, NepError, NEP_ERR_BREAKDOWN
from .errmeasure import DefaultErrmeasure, estimate_errors, estimate_errors_async
from .exceptions import NoConvergenceException
from .linsolvers import DefaultLinSolverCreator, create_linsolver
from .nep import CDT, to_host, to_host_cm, stream_ptr

EPS = np.finfo(float).eps


class _RefinementMiss(Exception):
    """a step's recorded backward errors show that UMFPACK's rule wanted more refinement than the step took"""


class _NativeRunMiss(Exception):
    """nep_iar_run returned NEP_ERR_RETRY for a reason the step-at-a-time pipeline handles itself (a device eigen-decomposition
    that reported a failure: that pipeline redoes the one decomposition with LAPACK)"""


class _OrthPassMiss(Exception):
    """the device-side DGKS of a step still met the re-orthogonalisation criterion after its last ENQUEUED pass (the
    reference's IterativeSolvers DGKS repeats while ||w|| < ||c|| / sqrt(2), without a bound): the call is re-run with the
    step-synchronous loop, whose nep_orth repeats until the criterion is no longer met"""


def iar(nep, orthmethod=dense.DGKS, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6,
        errmeasure=None, sigma=0.0, gamma=1.0, v=None, logger=0, check_error_every=1, proj_solve=False,
        errhist=None, timers=None, return_device=False, inner_solver_method=None):
    """src/method_iar.jl:56-141"""
    if v is None:
        v = np.random.randn(nep.size(1))
    kw = dict(orthmethod=orthmethod, maxit=maxit, linsolvercreator=linsolvercreator, tol=tol, neigs=neigs, errmeasure=errmeasure,
              sigma=sigma, gamma=gamma, v=v, logger=logger, check_error_every=check_error_every, proj_solve=proj_solve,
              errhist=errhist, timers=timers, return_device=return_device, inner_solver_method=inner_solver_method)
    flags = {}
    while True:                      # both downgrades may be needed in one call (each at most once)
        try:
            return _iar(nep, **kw, **flags)
        except _RefinementMiss:      # never observed; the checked path decides every refinement on the host
            if flags.get("_native_step") is False:
                raise
            iar.refinement_misses += 1
            flags["_native_step"] = False
        except _OrthPassMiss:        # "twice is enough" failed for a step: exact DGKS semantics through the synchronous loop
            if flags.get("_force_sync"):
                raise
            iar.orth_pass_misses += 1
            flags["_force_sync"] = True
        except _NativeRunMiss:
            if flags.get("_native_run") is False:
                raise
            iar.native_run_misses += 1
            flags["_native_run"] = False
        if errhist is not None:
            del errhist[:]


iar.refinement_misses = 0            # calls that were re-run with checked solves (diagnostics, tests)
iar.orth_pass_misses = 0             # calls that were re-run because a step wanted more DGKS passes than were enqueued
iar.native_run_misses = 0           # calls whose one-call native run (nep_iar_run) was re-run through the step-at-a-time pipeline
iar.native_runs = 0                 # calls served by nep_iar_run
iar.dev_eig_fallbacks = 0            # checks whose device eigen-decomposition reported a failure and was redone by LAPACK


_CHECK_STREAMS = {}


def _check_stream():
    dev = torch.cuda.current_device()
    st = _CHECK_STREAMS.get(dev)
    if st is None:
        # the convergence checks are off the critical path: their stream gets the LOWEST priority the device offers, so that
        # the dispatcher serves the recurrence's kernels first when both streams have work (NEP_IAR_CHECK_PRIO overrides;
        # torch clamps to the device's range, a lower number is a higher priority)
        st = _CHECK_STREAMS[dev] = torch.cuda.Stream(priority=int(os.environ.get("NEP_IAR_CHECK_PRIO", "1")))
    return st


_EIG_STREAMS = {}
_EIG_WORK = {}


def _eig_streams(count, others=()):
    """streams for the device eigen-decompositions (3 ms one-wavefront kernels) that overlap with the streams in `others`
    (the recurrence's and the checks').  The runtime maps streams onto a small pool of hardware queues and gives no way to ask
    which; two streams on one queue serialise (measured: the convergence checks queued 4 ms per batch behind the
    decompositions).  So candidates are probed (nep_stream_pair_serializes, ~1.5 ms each, once per process and device) and the
    first ones that serialise with none of `others` nor with each other are kept."""
    dev = torch.cuda.current_device()
    sts = _EIG_STREAMS.setdefault(dev, [])
    if len(sts) >= count:
        return sts[:count]
    import ctypes as _C
    cands = [torch.cuda.Stream(priority=int(os.environ.get("NEP_IAR_EIG_PRIO", "0"))) for _ in range(int(os.environ.get("NEP_IAR_EIG_CANDIDATES", "8")))]
    probe = os.environ.get("NEP_IAR_EIG_PROBE", "1") != "0"
    for c in cands:
        if len(sts) >= count:
            break
        ok = True
        if probe:
            for o in list(others) + sts:
                r = c_i32(0)
                check(lib.nep_stream_pair_serializes(c_vp(o.cuda_stream), c_vp(c.cuda_stream), _C.byref(r)))
                if r.value:
                    ok = False
                    break
        if ok:
            sts.append(c)
    while len(sts) < count:              # no free hardware queue: better a shared one than none
        sts.append(cands[len(sts) % len(cands)])
        _eig_streams.shared = True
    return sts[:count]


_eig_streams.shared = False


_EIG_WORK_LOCK = threading.Lock()
_EIG_WORK_KEEP = 2          # idle blocks kept per device (260 MB each at m = 100); more concurrent calls allocate and free their own


def _eig_work_acquire(need):
    """scratch of the device eigen-decompositions, CHECKED OUT for one iar call (its checker thread): the eigenvalue and the
    eigenvector kernels of a batch are two launches that share this block, so two calls that run on one GPU at the same time
    must not share it (call A's inverse iteration would read call B's matrices, status words clean).  Sequential calls get
    the same block back."""
    dev = torch.cuda.current_device()
    with _EIG_WORK_LOCK:
        free = _EIG_WORK.setdefault(dev, [])
        for i, w in enumerate(free):
            if w.numel() >= need:
                return free.pop(i)
        if free:
            free.pop()                      # too small for this call: let it go instead of keeping both
    return torch.empty(need, dtype=torch.uint8, device="cuda")


def _eig_work_release(w):
    """back to the pool; the caller has drained the streams that used it"""
    with _EIG_WORK_LOCK:
        free = _EIG_WORK.setdefault(w.device.index, [])
        if len(free) < _EIG_WORK_KEEP:
            free.append(w)


def _native_errmeasure(errmeasure, nep):
    """(kind, fro) of an error measure nep_iar_run evaluates itself (0: ||M(lam)v|| / ||v||, 1: the SPMF backward error), or None"""
    from .errmeasure import ResidualErrmeasure, StandardSPMFErrmeasure
    e = getattr(errmeasure, "errm", errmeasure) if isinstance(errmeasure, DefaultErrmeasure) else errmeasure
    if type(e) is StandardSPMFErrmeasure and e.nep is nep:
        return 1, np.ascontiguousarray(e.coeffs, dtype=np.float64)
    if type(e) is ResidualErrmeasure and e.nep is nep:
        return 0, None
    return None


def _iar_native_run(nep, M0inv, orthmethod, m, tol, neigs, errkind, sigma, gamma, v, check_error_every, errhist, return_device):
    """the whole run as ONE foreign call (csrc/iar_run.hip nep_iar_run) -- what the Julia binding's `iar(nep::DeviceSPMF; ...)`
    method calls too (julia/NEPMI355X.jl); this host only marshals the inputs (derivative table, start vector, the f_t(lambda)
    callback) and shapes the outputs.  method_iar.jl:46-182."""
    import ctypes as _C
    from ._lib import IarOpts, IarResult, FV_EVAL, hptr, cdouble, NEP_ERR_RETRY, NEP_ERR_NOCONV
    n = nep.size(1)
    fv = nep.get_fv(); mt = len(fv)
    alpha = gamma ** np.arange(m + 1); alpha[0] = 0
    tab = nep.derivative_table(sigma, m)
    Ctab = np.asfortranarray((alpha[1:m + 1] / np.arange(1, m + 1))[:, None] * tab["fD"][1:m + 1, :], dtype=np.complex128)   # m x mt
    v0 = np.ascontiguousarray(v, dtype=np.complex128)
    rc_ = M0inv.refine_coefficients() if M0inv.umfpack_refinements > 0 else None
    if M0inv.umfpack_refinements > 0 and rc_ is None:
        return None
    hint = M0inv._recorded_plan if M0inv._recorded_plan is not None else M0inv._hint()
    o = IarOpts(m, int(check_error_every), dense._orth_code(orthmethod), int(max(0, M0inv.umfpack_refinements)), errkind[0],
                -1 if hint is None else int(hint), float(tol), float(neigs), cdouble(sigma.real, sigma.imag), cdouble(gamma.real, gamma.imag))
    res = IarResult()

    def fv_eval(ctx, nlam, lam_p, F_p):
        try:
            la = np.frombuffer((_C.c_double * (2 * nlam)).from_address(lam_p), dtype=np.complex128)
            F = np.frombuffer((_C.c_double * (2 * nlam * mt)).from_address(F_p), dtype=np.complex128).reshape(nlam, mt)
            for t, f in enumerate(fv):
                F[:, t] = f.values(la)
            return 0
        except Exception:              # an exception must not cross the foreign frame: reported as a failed callback
            import traceback
            traceback.print_exc()
            return 1
    cb = FV_EVAL(fv_eval)
    ldv = n * (m + 1)
    V = torch.empty((m + 1, ldv), dtype=CDT, device="cuda")
    Qd = torch.empty((m, n), dtype=CDT, device="cuda") if return_device else None
    Qh = None if return_device else torch.empty((m, n), dtype=CDT, pin_memory=True)
    lam = np.zeros(m, dtype=np.complex128)
    err = np.full((m, m), np.nan, order="F")
    st = lib.nep_iar_run(nep.dev.h, M0inv.lu.h, n, _C.addressof(o), hptr(v0), hptr(Ctab), mt,
                         hptr(rc_[0]) if rc_ else None, hptr(rc_[1]) if rc_ else None, hptr(errkind[1]) if errkind[1] is not None else None,
                         _C.cast(cb, c_vp), None, hptr(lam), c_vp(Qd.data_ptr()) if Qd is not None else None,
                         c_vp(Qh.data_ptr()) if Qh is not None else None, hptr(err), c_vp(V.data_ptr()), _C.addressof(res), stream_ptr())
    # what the run learnt about the refinement count belongs to this NEP and shift (FactorizeLinSolver._hint)
    if res.refine_hint_off:
        M0inv._note_hint(None)
    elif res.refine_plan >= 0 and M0inv.umfpack_refinements > 0:
        M0inv._recorded_plan = int(res.refine_plan)
        M0inv._note_hint(int(res.refine_plan))
    if st == NEP_ERR_RETRY:
        if res.retry_reason == 1:
            raise _RefinementMiss(0)
        if res.retry_reason == 2:
            raise _OrthPassMiss(0)
        raise _NativeRunMiss(res.retry_reason)
    if st not in (0, NEP_ERR_NOCONV):
        check(st)
    iar.native_runs += 1
    k = int(res.k); nret = int(res.nret)
    M0inv.solves += k
    if errhist is not None:
        for kc in range(1, k + 1):
            if kc % check_error_every == 0 or kc == m:
                errhist.append(err[kc - 1, :kc].copy())
    lam = lam[:nret].copy()
    Q = Qd[:nret] if return_device else Qh.numpy()[:nret].T
    if st == NEP_ERR_NOCONV:
        msg = "Number of iterations exceeded. maxit=%d." % m
        if res.nconv < 3:
            msg += "Try to change the inner_solver_method for better performance."
        raise NoConvergenceException(lam, to_host(Qd[:nret]) if return_device else Q, err[k - 1, :nret].copy(), msg)
    return lam, Q, V[:k]


def _iar(nep, orthmethod=dense.DGKS, maxit=30, linsolvercreator=None, tol=EPS * 10000, neigs=6,
         errmeasure=None, sigma=0.0, gamma=1.0, v=None, logger=0, check_error_every=1, proj_solve=False,
         errhist=None, timers=None, return_device=False, inner_solver_method=None, _native_step=True, _force_sync=False,
         _native_run=True):
    t_entry = time.perf_counter()
    n = nep.size(1); m = int(maxit)
    sigma = complex(sigma); gamma = complex(gamma)
    if linsolvercreator is None:
        linsolvercreator = DefaultLinSolverCreator()
    if errmeasure is None:
        errmeasure = DefaultErrmeasure(nep)
    if v is None:
        v = np.random.randn(n)
    # ---- the one-call route: pure SPMF operator, device LU, DGKS / CGS, an error measure the library evaluates itself
    M0inv = None
    if (_native_run and _native_step and not _force_sync and timers is None and not proj_solve and m <= dense.HESS_EIG_KMAX
            and dense._orth_code(orthmethod) in (0, 1) and os.environ.get("NEP_IAR_NATIVE_RUN", "1") != "0"
            and not any(os.environ.get(e) for e in ("NEP_IAR_SYNC", "NEP_IAR_PYSTEP", "NEP_IAR_TRACE", "NEP_IAR_ONE_STREAM", "NEP_IAR_PASSES"))
            and os.environ.get("NEP_IAR_EIG", "dev") != "host"):
        from .linsolvers import FactorizeLinSolver
        from .nep import AbstractSPMF
        errkind = _native_errmeasure(errmeasure, nep)
        pure = (isinstance(nep, AbstractSPMF) and type(nep).lincomb_rowscale is AbstractSPMF.lincomb_rowscale
                and type(nep).compute_Mlincomb is AbstractSPMF.compute_Mlincomb and type(nep).resid_norms is AbstractSPMF.resid_norms
                and hasattr(nep, "dev"))
        if errkind is not None and pure:
            M0inv = create_linsolver(linsolvercreator, nep, sigma)
            if type(M0inv) is FactorizeLinSolver and getattr(M0inv.lu, "h", None):
                out = _iar_native_run(nep, M0inv, orthmethod, m, tol, neigs, errkind, sigma, gamma, v, check_error_every, errhist,
                                      return_device)
                if out is not None:
                    return out
    tm = timers if timers is not None else {}
    for key in ("mlincomb", "solve", "orth", "ritz", "resid", "host_eig"):
        tm.setdefault(key, 0.0)
    sync = torch.cuda.synchronize if timers is not None else (lambda: None)

    # initialization (method_iar.jl:76-86)
    ldv = n * (m + 1)
    H = np.zeros((m + 1, m), dtype=np.complex128)
    alpha = gamma ** np.arange(m + 1); alpha[0] = 0
    # the synchronous little uploads below come BEFORE the linear solver: once the device is busy with the factorisation and
    # the apex build behind it, each of them waits for a slot between 300-600 us kernels (5 ms of host time for the lot) --
    # and the start vector BEFORE the 1.6 GB zero fill of the basis is enqueued (a pageable upload behind the fill held the host
    # for 0.46 ms; now the fill runs while the host assembles the call, the start vector goes in by a device copy behind it)
    v0 = np.asarray(v, dtype=np.complex128)
    v0d = torch.from_numpy(v0 / np.linalg.norm(v0)).to("cuda")
    t_marks = [("v0", time.perf_counter())]
    # derivative table at sigma (DerSPMF, NEPTypes.jl:1108-1128).  The coefficient rows
    # C[j-1,:] = alpha_j/j * f^(j)(sigma) do not depend on k: uploaded once, each step uses the first k rows
    tab = nep.derivative_table(sigma, m, rowscale=alpha[1:m + 1] / np.arange(1, m + 1))
    t_marks.append(("tab", time.perf_counter()))
    z = torch.empty(n, dtype=CDT, device="cuda")
    active = (np.arange(1, m + 2) * n).astype(np.int64)   # column j has (j+1) non-zero blocks
    active_d0 = torch.from_numpy(active).to("cuda")
    V = torch.zeros((m + 1, ldv), dtype=CDT, device="cuda")     # the fill overlaps with the host side of the factorisation below
    V[0, :n].copy_(v0d)                                         # (the fill on a side stream next to the factorisation: no gain)
    t_marks.append(("V", time.perf_counter()))
    # Asynchronous pipeline (default): nothing on the Arnoldi critical path waits for the device.  The DGKS decision
    # is taken on the device (nep_orth_dev), H's new column travels to pinned host memory behind an event that the eigen
    # worker waits for, the residual norms of the Ritz pairs come back the same way (nep_resid_batch_dev) -- the host
    # enqueues step k+1.. while the device is still executing step k.  `timers` (instrumented run), MGS and
    # NEP_IAR_SYNC=1 use the step-synchronous loop; both produce the same iterates.
    use_async = (timers is None and dense._orth_code(orthmethod) in (0, 1) and not os.environ.get("NEP_IAR_SYNC")
                 and not proj_solve and not _force_sync)
    pnep = None
    if proj_solve:                                       # method_iar.jl:89-92
        from .projection import create_proj_NEP, inner_solve, DefaultInnerSolver
        pnep = create_proj_NEP(nep, maxsize=min(n, m + 1))
        if inner_solver_method is None:
            inner_solver_method = DefaultInnerSolver()
    if use_async:
        active_d = active_d0
        Hdev = torch.zeros((m, m + 4), dtype=CDT, device="cuda")     # row k-1: h[0..k), beta, flags, 4 recorded omegas
        Hpin = torch.zeros((m, m + 4), dtype=CDT).pin_memory()
        Hnp = Hpin.numpy()
        evs = [None] * (m + 1)
        filled = [False] * (m + 1)
    t_ls = time.perf_counter()
    t_marks.append(("pre", t_ls))
    if M0inv is None:
        M0inv = create_linsolver(linsolvercreator, nep, sigma)
    sync(); tm["linsolver_setup"] = tm.get("linsolver_setup", 0.0) + time.perf_counter() - t_ls
    t_setup_done = time.perf_counter()
    if timers is not None and hasattr(M0inv, "lu"):
        tm["host_factorization"] = tm.get("host_factorization", 0.0) + M0inv.lu.t_factor
    # native step (csrc/driver.hip nep_iar_step): K1 -> K5 (+ refinement) -> shift -> K6 -> H row to pinned memory as
    # ONE foreign call per Arnoldi step.  Needs a pure SPMF operator and a device LU.  The refinement criterion is never
    # read back inside a step: the step records omega of every iterate behind the H row and fill_H replays UMFPACK's
    # stopping rule on the record (FactorizeLinSolver.review_recorded); a miss re-runs the call with checked solves.
    cstep = None
    if use_async and _native_step and not os.environ.get("NEP_IAR_PYSTEP"):
        from .linsolvers import FactorizeLinSolver
        from .nep import AbstractSPMF
        import ctypes as _C
        pure = (isinstance(nep, AbstractSPMF) and type(nep).lincomb_rowscale is AbstractSPMF.lincomb_rowscale
                and type(nep).compute_Mlincomb is AbstractSPMF.compute_Mlincomb and "Cdev" in tab)
        if pure and type(M0inv) is FactorizeLinSolver and getattr(M0inv.lu, "h", None):
            rc_ = M0inv.refine_coefficients() if M0inv.umfpack_refinements > 0 else None
            if M0inv.umfpack_refinements <= 0 or rc_ is not None:
                from ._lib import hptr
                work3 = torch.empty(3 * n, dtype=CDT, device="cuda")
                hh = c_vp()
                check(lib.nep_iar_create(nep.dev.h, M0inv.lu.h, n, m, c_vp(V.data_ptr()), ldv, c_vp(tab["Cdev"].data_ptr()), tab["m"],
                                         c_vp(active_d.data_ptr()), c_vp(work3.data_ptr()),
                                         hptr(rc_[0]) if rc_ else None, hptr(rc_[1]) if rc_ else None, len(nep.get_fv()),
                                         c_vp(Hdev.data_ptr()), c_vp(Hpin.data_ptr()), dense._orth_code(orthmethod), _C.byref(hh)))
                cstep = hh
    t_marks += [("ls", t_setup_done), ("cstep", time.perf_counter())]
    err = np.full((m, m), np.nan)
    lam = np.zeros(0, dtype=np.complex128); QT = None; idx = np.zeros(0, dtype=int)
    # ---- main loop.  The small dense eigenproblem of step k (host LAPACK, method_iar.jl:112; 7.5 ms at
    # k=100, ~190 ms summed over a run) is solved on worker threads WHILE the device runs the following
    # Arnoldi steps (mlincomb, solve, DGKS); the Ritz extraction + residuals of step k are enqueued as soon
    # as its decomposition is available, at most LAG steps late and always in order.  The arithmetic and
    # the returned quantities are those of the sequential loop; when the convergence test of step k ends
    # the iteration, the (at most LAG+1) speculative Arnoldi steps beyond k are simply dropped.
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    from ._affinity import cpu_budget
    # host eig of up to LAG+1 consecutive steps in flight: as many workers as the CPU budget of this rank allows (measured on
    # gun, 16-CPU budget: LAG 3 -> 86 ms per run, 5 -> 75, 9 -> 72, 15 -> 73)
    LAG = int(os.environ.get("NEP_IAR_LAG", str(max(1, min(12, cpu_budget() - 3)))))
    pool = ThreadPoolExecutor(max_workers=LAG + 1)
    state = {"lam": lam, "QT": QT, "idx": idx, "conv_eig": 0, "k_checked": 0}

    trace = {} if os.environ.get("NEP_IAR_TRACE") else None
    plans = [0] * (m + 1)
    t_marks.append(("pool", time.perf_counter()))

    def arnoldi_step(k):
        if cstep is not None:
            plan = M0inv.blind_plan_recorded()
            if plan > 0 and M0inv.settled_plan():
                plan |= 0x100                 # the kept iterate's backward error: recorded in every 8th step only
            plans[k] = plan
            t0 = time.perf_counter()
            check(lib.nep_iar_step(cstep, k, plan, stream_ptr()))
            if trace is not None:
                trace["native_s"] = trace.get("native_s", 0.0) + time.perf_counter() - t0
                trace["native_n"] = trace.get("native_n", 0) + 1
            M0inv.note_blind_solve(plan & 0xff)
            evs[k] = "native"
            return
        t0 = time.perf_counter()
        # z = sum_{j=1..k} alpha_{j+1}/j * M^(j)(sigma) * V_k block j
        nep.lincomb_rowscale(tab, k, V.data_ptr() + 16 * (k - 1) * ldv, n, z)
        sync(); t1 = time.perf_counter()
        # new vector, block 0: -M(sigma)^{-1} z ; blocks 1..k: shifted/scaled old column
        vv = V[k]
        M0inv.solve_dev(z, out=vv[:n].reshape(1, n), scale=-1.0)
        sync(); t2 = time.perf_counter()
        check(lib.nep_iar_shift_scale(n, k, c_vp(V.data_ptr() + 16 * (k - 1) * ldv),
                                      c_vp(vv.data_ptr()), stream_ptr()))
        if use_async:
            dense.orthogonalize_and_normalize_dev(V, vv, k, Hdev[k - 1], rows=n * (k + 1), ldv=ldv, active_dev=active_d,
                                                  method=orthmethod)
            Hpin[k - 1, :k + 2].copy_(Hdev[k - 1, :k + 2], non_blocking=True)
            evs[k] = torch.cuda.Event()
            evs[k].record()
            return
        h, beta, _ = dense.orthogonalize_and_normalize(V, vv, k, rows=n * (k + 1), ldv=ldv,
                                                       active_rows=active, method=orthmethod)
        H[:k, k - 1] = h; H[k, k - 1] = beta
        sync(); t3 = time.perf_counter()
        tm["mlincomb"] += t1 - t0; tm["solve"] += t2 - t1; tm["orth"] += t3 - t2

    def timed_eig(Hk):
        t = time.perf_counter()
        r = _hosteig.eig(Hk, hessenberg=True)       # LAPACK through ctypes: runs without the GIL (numpy/scipy hold it)
        return r, time.perf_counter() - t

    def fill_H(kk):
        """columns 1..kk of H from the pinned buffer (their copies are complete once evs[kk] is)"""
        for j in range(1, kk + 1):
            if not filled[j]:
                row = Hnp[j - 1]
                if int(row[j + 1].imag) & 2:
                    raise NepError(NEP_ERR_BREAKDOWN, "orthogonalisation breakdown in step %d: ||w|| = %g" % (j, row[j].real))
                if int(row[j + 1].imag) & 1 and dense._orth_code(orthmethod) == 0:
                    raise _OrthPassMiss(j)        # another DGKS pass was wanted after the last enqueued one
                if cstep is not None and M0inv.umfpack_refinements > 0:
                    if not M0inv.review_recorded(row[j + 2:j + 4].view(np.float64), plans[j] & 0xff,
                                                 final_recorded=not (plans[j] & 0x100 and j % 8 != 0)):
                        raise _RefinementMiss(j)
                H[:j, j - 1] = row[:j]
                H[j, j - 1] = row[j].real
                filled[j] = True

    def timed_eig_async(kk):
        if evs[kk] == "native":
            check(lib.nep_iar_wait(cstep, kk))      # ctypes releases the GIL
        else:
            evs[kk].synchronize()      # releases the GIL; H's columns <= kk are in pinned memory afterwards
        if trace is not None:
            trace["dev_done_%d" % kk] = time.perf_counter()
        fill_H(kk)
        return timed_eig(H[:kk, :kk].copy())

    def finish_check(kc, fut):
        (D, Z), t_eig = fut.result()
        tm["host_eig"] += t_eig
        t4 = time.perf_counter()
        laml = sigma + gamma / D
        if proj_solve:
            # method_iar.jl:118-131: orthonormal basis QQ of span(V[0:n, 0:kc]) (on the device: Gram matrix by K9,
            # scaling by K7, twice), Galerkin projection, inner solve started from RR*Z
            # (rank revealing: eigen-decomposition of the Gram matrix, directions below 1e-13 of the largest are dropped
            # -- the first-block rows of the Krylov basis become numerically dependent, and kc may exceed n)
            R_tot = np.eye(kc, dtype=complex)
            Qd = V; ldq = ldv; kq = kc
            for _ in range(2):
                QTm = dense.gemm_ts(Qd, np.eye(kq, dtype=complex), rowmajor=True, k=kq, rows=n, ldz=ldq)
                G = dense.gemm_h_rm(QTm, QTm, n, kq, kq)                             # K9 Gram matrix
                wg, Ug = np.linalg.eigh((G + G.conj().T) / 2)
                keep = wg > 1e-13 * wg[-1]
                T = Ug[:, keep] / np.sqrt(wg[keep])[None, :]                       # kq x r
                Rc = (np.sqrt(wg[keep])[:, None] * Ug[:, keep].conj().T)           # r x kq,  block = Q Rc
                Qd = dense.gemm_ts(Qd, T, k=kq, rows=n, ldz=ldq)                   # (r, n) column-major
                ldq = n; kq = int(np.sum(keep))
                R_tot = Rc @ R_tot
            pnep.set_projectmatrices(Qd, Qd)
            lamp, Qp = inner_solve(inner_solver_method, pnep, V=R_tot @ Z, lamv=laml.copy(), neigs=kc, sigma=np.mean(laml))
            laml = np.asarray(lamp); Qp = np.asarray(Qp)
            QTl = dense.gemm_ts(Qd, Qp, rowmajor=True, k=kq, rows=n, ldz=n)
        else:
            QTl = dense.gemm_ts(V, Z, rowmajor=True, k=kc, rows=n, ldz=ldv)       # (n, kc) row-major
        sync(); t5 = time.perf_counter()
        e = estimate_errors(errmeasure, laml, QTl) if len(laml) else np.zeros(0)
        t6 = time.perf_counter()
        tm["ritz"] += t5 - t4; tm["resid"] += t6 - t5
        ne = min(len(e), m)
        conv = int(np.sum(e < tol))
        idxl = np.argsort(e, kind="stable")
        err[kc - 1, :ne] = e[idxl][:ne]
        if errhist is not None:
            errhist.append(err[kc - 1, :ne].copy())
        if kc == m or conv >= neigs:
            nrof = int(min(len(laml), neigs))
            laml = laml[idxl[:nrof]]
            idxl = idxl[:nrof]
        state.update(lam=laml, QT=QTl, idx=idxl, conv_eig=conv, k_checked=kc)

    # The convergence checks (Ritz block K7 + residual batch K2 of step kc) read columns of V that are final by the time
    # eig(H_kc) exists -- the eigen worker waited for step kc's event -- and write only their own buffers, so they run on a
    # second stream next to the Arnoldi recurrence (latency-bound small kernels at gun size) instead of in line with it.
    # Only with the native step: there this thread touches none of the scratch the checks use (csrc/spmv.hip: coef / part /
    # ring belong to the residual batch, cwpart / cwring to the refinement inside the step).
    check_thread = cstep is not None and not os.environ.get("NEP_IAR_ONE_STREAM") and hasattr(errmeasure, "batch_async")
    # ONE check stream per device for the life of the process: torch's caching allocator keeps freed blocks per stream, and a
    # fresh stream per call (32 of them in torch's pool) made every stream build its own cache of Ritz blocks
    check_stream = _check_stream() if (check_thread and not os.environ.get("NEP_IAR_CHECK_MAIN_STREAM")) else None

    # eigen-decompositions on the device (csrc/hesseig.hip) instead of LAPACK on host worker threads: no eig thread, no waiter
    # per step -- 133 ms of host CPU per headline call gone; NEP_IAR_EIG=host keeps the round-3 route (and is the fallback for
    # maxit beyond the LDS-resident limit, or when a decomposition reports a failure)
    dev_eig = (check_thread and check_stream is not None and m <= dense.HESS_EIG_KMAX
               and os.environ.get("NEP_IAR_EIG", "dev") != "host")
    # its stream: one whose hardware queue is shared neither with this thread's stream nor with the check stream (probed once
    # per process and device, on this thread, before the first step is enqueued)
    eig_stream = _eig_streams(1, others=(torch.cuda.current_stream(), check_stream))[0] if dev_eig else None

    def launch_check(kc, fut):
        """eigen-decomposition of step kc is available: enqueue Ritz block (K7) + residual batch (K2), no waiting"""
        (D, Z), t_eig = fut.result()
        laml = sigma + gamma / D
        if check_stream is None:
            QTl = dense.gemm_ts(V, Z, rowmajor=True, k=kc, rows=n, ldz=ldv)
            return kc, laml, QTl, estimate_errors_async(errmeasure, laml, QTl)
        with torch.cuda.stream(check_stream):
            QTl = dense.gemm_ts(V, Z, rowmajor=True, k=kc, rows=n, ldz=ldv)
            return kc, laml, QTl, estimate_errors_async(errmeasure, laml, QTl)

    def consume_check(kc, laml, QTl, perr):
        e = perr.get()
        conv = int(np.sum(e < tol))
        idxl = np.argsort(e, kind="stable")
        err[kc - 1, :kc] = e[idxl]
        if errhist is not None:
            errhist.append(err[kc - 1, :kc].copy())
        if kc == m or conv >= neigs:
            nrof = int(min(len(laml), neigs))
            laml = laml[idxl[:nrof]]
            idxl = idxl[:nrof]
        state.update(lam=laml, QT=QTl, idx=idxl, conv_eig=conv, k_checked=kc)

    k = 1
    pending = deque()              # (k, future) in increasing k; checks are always consumed in order
    # the k x k eigenproblems gain nothing from a threaded BLAS (7.5 ms at k=100 with 1 or 64 threads) while its
    # spinning worker threads slow the launching thread down: pin BLAS to one thread for the duration of the loop
    import nep_amd_hostlu as _nep_hostlu
    ctl = _nep_hostlu.blas_controller()
    blas_guard = ctl.limit(limits=1) if (ctl is not None and os.environ.get("NEP_IAR_BLAS_GUARD", "1") != "0") else None
    if blas_guard is not None:
        blas_guard.__enter__()
    t_marks.append(("blas", time.perf_counter()))
    try:
        if use_async and check_thread:
            # native step: this thread only issues nep_iar_step (one foreign call per step, GIL released); the checker thread
            # waits for the eigen-decompositions in order, enqueues their checks on check_stream and consumes the results.
            # `slots` bounds how far the recurrence runs ahead of the checks (LAG + 1 decompositions in flight, as before).
            import queue, threading
            todo = queue.Queue(); failure = []
            # neigs = Inf: the iteration always runs to maxit, nothing the recurrence does ahead of the checks can be wasted,
            # so it is not throttled at all (the eigen-decompositions of the last steps -- half of all eig time -- then
            # queue up behind the device instead of pacing it)
            unthrottled = np.isinf(neigs) and not os.environ.get("NEP_IAR_THROTTLE")
            # (device decompositions go out in batches of up to NEP_IAR_EIG_BATCH steps: the look-ahead is that batch, whatever the CPU budget)
            slots = threading.Semaphore(m + 1 if unthrottled else (max(LAG + 1, int(os.environ.get("NEP_IAR_EIG_BATCH", "16"))) if dev_eig else LAG + 1))

            def checker():
                inflight = deque()
                try:
                    while True:
                        item = todo.get()
                        if item is None:
                            break
                        if state["conv_eig"] >= neigs:
                            slots.release(); continue
                        t0 = time.perf_counter()
                        fut_ = item[1]; fut_.result(); t1 = time.perf_counter()
                        inflight.append(launch_check(*item))
                        slots.release()
                        t2 = time.perf_counter()
                        while inflight and state["conv_eig"] < neigs and (len(inflight) > LAG or inflight[0][3].ready()):
                            consume_check(*inflight.popleft())
                        if trace is not None:
                            t3 = time.perf_counter()
                            trace["chk_wait"] = trace.get("chk_wait", 0.0) + t1 - t0
                            trace["chk_launch"] = trace.get("chk_launch", 0.0) + t2 - t1
                            trace["chk_consume"] = trace.get("chk_consume", 0.0) + t3 - t2
                    while inflight and state["conv_eig"] < neigs:
                        consume_check(*inflight.popleft())
                except BaseException as exc:          # re-raised on the calling thread
                    failure.append(exc)
                    slots.release()
                finally:
                    # the library's thread-local scratch of this thread goes back to the shared pool when the thread ends:
                    # nothing this thread enqueued may still be pending then (dropped speculative checks)
                    try:
                        (check_stream.synchronize() if check_stream is not None else torch.cuda.current_stream().synchronize())
                    except Exception:
                        pass

            def checker_dev():
                """the same checks with eig(H_kc) on the device.  (A) The decompositions of consecutive steps go out as BATCHES:
                one launch, one workgroup per step (a decomposition is a serial chain, 3 ms at k = 100, ten Arnoldi steps: the
                steps' decompositions have to overlap each other, and more than two or three extra streams stall the
                recurrence's own queue), on one of NS eig streams behind the event of the batch's last step -- nothing of it
                needs the host.  (B) When a batch's eigenvalues have reached the pinned mirror (an event behind the first
                kernel; only the inverse iterations are still running) the host forms lambda = sigma + gamma / D and f_t(lambda)
                per step and enqueues Ritz GEMM (B operand = the device eigenvector block) + residual batch on the check stream.
                (C) The 2 kc norms come back behind another event.  One thread polls the event queues; no LAPACK, no waiters."""
                BMAX = max(1, int(os.environ.get("NEP_IAR_EIG_BATCH", "16")))
                LASTB = max(1, int(os.environ.get("NEP_IAR_EIG_LAST", "8")))
                T100 = float(os.environ.get("NEP_IAR_EIG_MS100", "3.3"))     # ms of one decomposition at k = 100 (scales as k^2)
                TSTEP = float(os.environ.get("NEP_IAR_EIG_MSSTEP", "0.35"))  # ms per Arnoldi step (gun, k ~ 100)
                est = eig_stream
                wsz = (dense.hess_eig_worksize(m) + 15) // 16 * 16
                work = _eig_work_acquire(BMAX * wsz)
                wdev = torch.empty((m, m + 2), dtype=CDT, device="cuda")
                wpin = torch.zeros((m, m + 2), dtype=CDT).pin_memory()
                wnp = wpin.numpy()
                pendA = deque(); stA = deque(); stC = deque()
                done = False
                t_poll = float(os.environ.get("NEP_IAR_POLL_US", "30")) * 1e-6
                force_fail = int(os.environ.get("NEP_IAR_EIG_FAIL_AT", "0"))   # tests: treat this step's decomposition as failed
                # Batch plan.  A batch occupies the eig stream for the time of its LARGEST decomposition whatever its size, and
                # cannot start before its last step has run: batches of about twice (decomposition time / step time) steps keep
                # the stream half idle, so the last batch starts the moment step m is done; that last batch is kept smaller,
                # because its checks (Ritz GEMM + residual batch, 0.17 ms each at k = 100) all come after its 3 ms.
                # neigs = Inf: every check step is known in advance -> boundaries planned backwards from m.  Otherwise (the
                # recurrence is throttled to LAG + 1 steps ahead of the checks) a batch is whatever is pending.
                plan_end = None
                if unthrottled:
                    allk = [kk for kk in range(1, m + 1) if kk % check_error_every == 0 or kk == m]
                    ends = []; e_ = len(allk)
                    size = min(LASTB, e_)
                    while e_ > 0:
                        ends.append(allk[e_ - 1]); e_ -= size
                        if e_ > 0:
                            size = int(min(BMAX, e_, max(1, np.ceil(2.0 * T100 * (allk[e_ - 1] / 100.0) ** 2 / (TSTEP * check_error_every)))))
                    plan_end = set(ends)

                def host_redo(kc):
                    """the device decomposition of step kc reported a failure: LAPACK on the host, Ritz block from its Z"""
                    iar.dev_eig_fallbacks += 1
                    (D, Z), _ = timed_eig(H[:kc, :kc].copy())
                    with torch.cuda.stream(check_stream):
                        QTl = dense.gemm_ts(V, Z, rowmajor=True, k=kc, rows=n, ldz=ldv)
                    return D, QTl

                def launch_batch(count):
                    kcs = [pendA.popleft() for _ in range(count)]
                    k0 = kcs[0]; nb = len(kcs); kmax = kcs[-1]
                    kstep = (kcs[1] - k0) if nb > 1 else 0
                    with torch.cuda.stream(est):
                        sp_ = stream_ptr()
                        check(lib.nep_iar_stream_wait(cstep, kmax, sp_))
                        wrow = c_vp(wdev.data_ptr() + 16 * (k0 - 1) * (m + 2))
                        mrow = c_vp(wpin.data_ptr() + 16 * (k0 - 1) * (m + 2))
                        rc_ = lib.nep_hess_eigvals_batch_dev(nb, k0, kstep, c_vp(Hdev.data_ptr()), m + 4, wrow, kstep * (m + 2),
                                                            c_vp(work.data_ptr()), wsz, mrow, kstep * (m + 2), sp_)
                        if rc_ != 0 or os.environ.get("NEP_IAR_EIG_LAUNCH_FAIL"):
                            # the launch itself was refused (e.g. a device that does not grant the kernel's 160 KB of LDS): not a
                            # reason to abort the run -- the batch's decompositions go to LAPACK on the host (host_redo), behind an
                            # event that says its last step has run
                            evW = torch.cuda.Event(); evW.record()
                            stA.append((kcs, None, kmax, evW, None))
                            return
                        evW = torch.cuda.Event(); evW.record()
                        Zb = torch.empty((nb, kmax, kmax), dtype=CDT, device="cuda")
                        check(lib.nep_hess_eigvecs_batch_dev(nb, k0, kstep, wrow, kstep * (m + 2), c_vp(Zb.data_ptr()), kmax, kmax * kmax,
                                                             c_vp(work.data_ptr()), wsz, mrow, kstep * (m + 2), sp_))
                        evZ = torch.cuda.Event(); evZ.record()
                    stA.append((kcs, Zb, kmax, evW, evZ))

                def batch_ready():
                    """number of pending steps that form the next batch (0: wait for more)"""
                    if not pendA:
                        return 0
                    cnt = 1
                    while cnt < len(pendA) and cnt < BMAX and pendA[cnt] - pendA[cnt - 1] == pendA[1] - pendA[0] and (plan_end is None or pendA[cnt - 1] not in plan_end):
                        cnt += 1
                    if plan_end is None or pendA[cnt - 1] in plan_end or cnt >= BMAX or done:
                        return cnt
                    return 0

                try:
                    while True:
                        progressed = False
                        # ---- new steps
                        while not done:
                            try:
                                item = todo.get_nowait() if (pendA or stA or stC) else todo.get()
                            except queue.Empty:
                                break
                            progressed = True
                            if item is None:
                                done = True
                            elif state["conv_eig"] >= neigs:
                                slots.release()
                            else:
                                pendA.append(item[0])
                        # ---- (A) batches onto the eig stream (its order is the order of the steps: nothing to wait for here)
                        while state["conv_eig"] < neigs:
                            cnt = batch_ready()
                            if not cnt:
                                break
                            launch_batch(cnt); progressed = True
                        # ---- (B) eigenvalues on the host: Ritz values, coefficients, Ritz block + residual batch
                        while stA and state["conv_eig"] < neigs and stA[0][3].query():
                            progressed = True
                            kcs, Zb, kmax, evW, evZ = stA.popleft()
                            waited = False
                            for b_, kc in enumerate(kcs):
                                if trace is not None:
                                    trace["dev_done_%d" % kc] = time.perf_counter()
                                fill_H(kc)
                                if Zb is None or wnp[kc - 1, kc].real != 0 or kc == force_fail:   # launch refused / QR iteration gave up (never observed)
                                    D, QTl = host_redo(kc)
                                else:
                                    D = wnp[kc - 1, :kc].copy()
                                    with torch.cuda.stream(check_stream):
                                        if not waited:
                                            check_stream.wait_event(evZ); waited = True
                                        QTl = dense.gemm_ts_dev(V, Zb[b_], kc, kmax, rowmajor=True, k=kc, rows=n, ldz=ldv)
                                laml = sigma + gamma / D
                                with torch.cuda.stream(check_stream):
                                    perr = estimate_errors_async(errmeasure, laml, QTl)
                                stC.append((kc, laml, QTl, perr, Zb))
                                slots.release()
                        # ---- (C) norms on the host
                        while stC and state["conv_eig"] < neigs and stC[0][3].ready():
                            progressed = True
                            kc, laml, QTl, perr, Zb = stC.popleft()
                            if Zb is not None and (wnp[kc - 1, kc + 1].real != 0 or kc == -force_fail):   # an inverse iteration did not grow: redo on the host
                                D, QTl = host_redo(kc)
                                laml = sigma + gamma / D
                                with torch.cuda.stream(check_stream):
                                    perr = estimate_errors_async(errmeasure, laml, QTl)
                            consume_check(kc, laml, QTl, perr)
                        if state["conv_eig"] >= neigs:
                            while pendA:
                                pendA.popleft(); slots.release()
                            while stA:
                                for _ in stA.popleft()[0]:
                                    slots.release()
                            stC.clear()
                        if done and not pendA and not stA and not stC:
                            break
                        if not progressed:
                            time.sleep(t_poll)
                except BaseException as exc:          # re-raised on the calling thread
                    failure.append(exc)
                    slots.release()
                finally:
                    try:
                        est.synchronize()             # dropped speculative decompositions still read Hdev / write wdev
                        check_stream.synchronize()
                        _eig_work_release(work)       # (only behind a clean drain: a block with work pending is dropped, not shared)
                    except Exception:
                        pass

            th = threading.Thread(target=checker_dev if dev_eig else checker, name="nep-iar-check", daemon=True)
            th.start()
            t_marks.append(("thread", time.perf_counter()))
            try:
                BATCH = max(1, min(4, LAG // 2))
                if unthrottled:
                    BATCH = int(os.environ.get("NEP_IAR_BATCH", "8"))
                while k <= m and state["conv_eig"] < neigs and not failure:
                    # as many steps as there are free check slots (at most BATCH) go to the device in ONE foreign call: the
                    # interpreter lock is released for all of it and re-acquired once (with one call per step this thread
                    # queued for the lock behind the checker after every step: 330 us per step instead of 120)
                    nb = 0
                    while nb < BATCH and k + nb <= m:
                        due = ((k + nb) % check_error_every == 0) or (k + nb == m)
                        if due:
                            if nb == 0:
                                while not slots.acquire(timeout=0.05):
                                    if failure or not th.is_alive():
                                        break
                            elif not slots.acquire(blocking=False):
                                break
                        nb += 1
                    if failure:
                        break
                    plan = M0inv.blind_plan_recorded()
                    if plan > 0 and M0inv.settled_plan():
                        plan |= 0x100             # the kept iterate's backward error: recorded in every 8th step only
                    t0 = time.perf_counter()
                    check(lib.nep_iar_steps(cstep, k, nb, plan, stream_ptr()))
                    if trace is not None:
                        trace["native_s"] = trace.get("native_s", 0.0) + time.perf_counter() - t0
                        trace["native_n"] = trace.get("native_n", 0) + nb
                    for kk in range(k, k + nb):
                        plans[kk] = plan
                        M0inv.note_blind_solve(plan & 0xff)
                        evs[kk] = "native"
                        if trace is not None:
                            trace["enq_%d" % kk] = time.perf_counter()
                        if (kk % check_error_every == 0) or (kk == m):
                            todo.put((kk, None if dev_eig else pool.submit(timed_eig_async, kk)))
                    k += nb
            finally:
                todo.put(None)
                th.join()
            if failure:
                raise failure[0]
        elif use_async:
            pend_err = deque()         # checks whose device work is enqueued, in increasing k
            while k <= m and state["conv_eig"] < neigs:
                arnoldi_step(k)
                if (k % check_error_every == 0) or (k == m):
                    pending.append((k, pool.submit(timed_eig_async, k)))
                # waiting for the oldest decomposition when more than LAG are in flight is what bounds how far the
                # host runs ahead of the device
                while pending and (len(pending) > LAG or pending[0][1].done()):
                    pend_err.append(launch_check(*pending.popleft()))
                while pend_err and state["conv_eig"] < neigs and (len(pend_err) > LAG or pend_err[0][3].ready()):
                    consume_check(*pend_err.popleft())
                k += 1
            while (pending or pend_err) and state["conv_eig"] < neigs:
                if pend_err:
                    consume_check(*pend_err.popleft())
                else:
                    pend_err.append(launch_check(*pending.popleft()))
        else:
            while k <= m and state["conv_eig"] < neigs:
                arnoldi_step(k)
                if (k % check_error_every == 0) or (k == m):
                    pending.append((k, pool.submit(timed_eig, H[:k, :k].copy())))
                # consume finished eigen-decompositions; never let the check lag more than LAG steps
                while pending and state["conv_eig"] < neigs and (len(pending) > LAG or pending[0][1].done()):
                    finish_check(*pending.popleft())
                k += 1
            while pending and state["conv_eig"] < neigs:
                finish_check(*pending.popleft())
    finally:
        for _, f in pending:
            f.cancel()
        pool.shutdown(wait=True)
        if check_stream is not None:
            check_stream.synchronize()       # dropped speculative checks may still read V / write their blocks
        if cstep is not None:
            lib.nep_iar_destroy(cstep)
        if trace is not None:
            t_end = time.perf_counter()
            ks = [kk for kk in (1, 10, 25, 50, 75, 100) if "enq_%d" % kk in trace and "dev_done_%d" % kk in trace]
            print("iar trace (ms after entry): setup %.1f (%s) | " % ((t_setup_done - t_entry) * 1e3, " ".join("%s %.2f" % (a_, (b_ - t_entry) * 1e3) for a_, b_ in t_marks))
                  + " ".join("k=%d enq %.1f dev %.1f" % (kk, (trace["enq_%d" % kk] - t_entry) * 1e3, (trace["dev_done_%d" % kk] - t_entry) * 1e3) for kk in ks)
                  + " | end %.1f | native steps %d, %.1f ms inside nep_iar_step; checker: wait eig %.1f launch %.1f consume %.1f ms" % ((t_end - t_entry) * 1e3, trace.get("native_n", 0), trace.get("native_s", 0.0) * 1e3, trace.get("chk_wait", 0) * 1e3, trace.get("chk_launch", 0) * 1e3, trace.get("chk_consume", 0) * 1e3))
        if use_async and os.environ.get("NEP_IAR_PASSES"):
            torch.cuda.synchronize()
            print("orth passes per step:", [int(Hnp[j - 1][j + 1].real) for j in range(1, m + 1)], "flags",
                  [int(Hnp[j - 1][j + 1].imag) for j in range(1, m + 1)])
        if blas_guard is not None:
            blas_guard.__exit__(None, None, None)
    lam, QT, idx, conv_eig = state["lam"], state["QT"], state["idx"], state["conv_eig"]
    k = state["k_checked"] if state["k_checked"] > 0 else k - 1
    if conv_eig < neigs and neigs != np.inf:
        Q = to_host(dense.rowmajor_to_cols(QT, idx[:len(lam)])) if QT is not None else None
        msg = "Number of iterations exceeded. maxit=%d." % maxit
        if conv_eig < 3:
            msg += "Try to change the inner_solver_method for better performance."
        raise NoConvergenceException(lam, Q, err[k - 1, :len(lam)], msg)
    nc = min(len(lam), conv_eig)
    lam = lam[:nc]
    Qd = dense.rowmajor_to_cols(QT, idx[:nc])          # (nc, n) = column-major n x nc
    if return_device:
        return lam, Qd, V[:k]
    return lam, to_host_cm(Qd), V[:k]
