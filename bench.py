#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X NEP backend (BASELINE.json metric:
"eigenpairs/sec + compute_Mlincomb GB/s, gun SPMF iar m=100").

One "step" = one complete `iar(gun_spmf_scaled, sigma=0, gamma=1, maxit=100, neigs=Inf, v=ones,
tol=1e-10, check_error_every=1)` call on one GPU (config C2 of SURVEY.md section 8d, n = 9956), including
the numeric factorisation of M(sigma) -- on the device (csrc/lufac.hip) once the plan of the sparsity pattern exists (it is
built from the first call's host SuperLU factorisation during warm-up), on the host otherwise; `ms_per_step_host_lu`
reports the same step with the host factorisation every time.  The NEP's matrices are
resident in HBM before the timed region.  iar is a sequential Krylov recurrence and does not shard
("replicas only"), so with --gpus N every rank runs an independent replica of the same
workload (weak scaling) and `value` is the whole-job rate: sum over ranks of converged eigenpairs
divided by the max-over-ranks wall time.

The JSON line also carries
  value_excl_setup   the same rate with the linear-solver set-up (numeric LU + device-built block inverses)
                     taken out of every step (set-up time from the instrumented run)
  kernels / phase_share   per-phase time of one instrumented iar run and each phase's share of it
  roofline      the kernel pair with the largest share of device time (K6: k_orth_dots + k_orth_update), ONE Gram-Schmidt
                pass at a FIXED shape (iar step k = maxit), algorithmic bytes / average HIP-event time per pass; the committed
                rocprofv3 summary of exactly this loop (`python bench.py --only orth`, profiles/r2_orth_kernel_stats.csv) gives the
                same average
  roofline_k5   the fixed-shift solve (time-dominant phase of round 1) against SURVEY.md section 8d ALGORITHMIC bytes
  roofline_compute_Mlincomb   the kernel the metric names at k = maxit and k = 1 on the gun matrices;
                roofline_wep_scale: the same kernels on the n = 1e6 waveguide matrices (the HBM-bound size)
  cpu_baseline  the CPU oracle on the SAME configuration (maxit = 100), host core count and threads used, plus the
                single-thread C port and the OpenMP all-cores variant of compute_Mlincomb
  beyn_sharded  config C4 (the path that shards over ranks) with parity against the CPU oracle
  c3_nleigs, c5_wep   the other BASELINE configurations (rank 0, one run each)
"""
import argparse
import json
import os
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # small host BLAS only; big pools stall the launch thread
    os.environ.setdefault(_v, "8")
# OpenBLAS' idle workers spin for 2^28 cycles (~0.1 s) after every job before they sleep: with a k x k LAPACK call every few
# hundred microseconds they never do -- seven threads at 100 % for the whole benchmark, and the container's 16-CPU cgroup quota
# throttled in 26 of 27 scheduler periods (scripts/diag/cfs_throttle.py, cpu_by_thread.py).  2^12 cycles: they sleep at once.
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np
import torch
import torch.distributed as dist

MFMA64_PEAK_TFLOPS = 78.6        # dense FP64 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
RED_DEV = "cuda"        # device of the tensors that go through torch.distributed reductions


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=9956)
    ap.add_argument("--maxit", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-maxit", type=int, default=0, help="CPU oracle iterations (0 = the benchmark's own maxit)")
    ap.add_argument("--permc", default=None)
    ap.add_argument("--no-beyn", action="store_true", help="skip the sharded contour_beyn extra (C4)")
    ap.add_argument("--no-beyn-parity", action="store_true", help="skip the CPU-oracle twin of C4 (about 5 s)")
    ap.add_argument("--no-wep-roofline", action="store_true", help="skip the waveguide-scale K1 roofline extra")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 (nleigs) summary")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 (waveguide tiar, n = 1e6) summary")
    ap.add_argument("--no-cold", action="store_true", help="skip the fresh-process first-call measurement (about 3 s)")
    ap.add_argument("--no-c5-oracle", action="store_true", help="skip the CPU oracle of the C5 twin (about 40 s)")
    ap.add_argument("--c5-oracle-full", action="store_true",
                    help="also run the CPU oracle of C5 at the FULL size by the reference's own route (matrix-free Schur complement + "
                         "Sylvester-SMW preconditioned GMRES): tens of minutes of host time, one run; the record of the last such run is "
                         "profiles/r4_c5_oracle_full.json")
    ap.add_argument("--c5-nx", type=int, default=1003)
    ap.add_argument("--c5-nz", type=int, default=999)
    ap.add_argument("--only", default=None, choices=["orth", "k5", "mlincomb", "wepscale", "c5step"],
                    help="run only one fixed-shape kernel loop (the command the committed rocprofv3 summaries come from)")
    ap.add_argument("--reps", type=int, default=50)
    return ap.parse_args()


def event_loop(fn, reps, warm=5):
    """average ms per call of fn, HIP events on the current stream around `reps` back-to-back calls"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def mlincomb_roofline(na, nep, k, reps=50):
    """algorithmic bytes / average HIP-event time of nep_mlincomb at k columns"""
    n = nep.n
    V = torch.randn((k, n), dtype=torch.float64, device="cuda").to(torch.complex128)
    V = V + 1j * torch.randn((k, n), dtype=torch.float64, device="cuda")
    fD = np.column_stack([f.derivs(0.0, k + 1) for f in nep.get_fv()])
    Cm = fD[1:k + 1] / np.arange(1, k + 1)[:, None]
    z = torch.empty(n, dtype=torch.complex128, device="cuda")
    Cdev = na.to_dev(Cm)          # coefficient table resident on the device, as in iar (nep_mlincomb_dev)
    ms = event_loop(lambda: nep.dev.mlincomb_dev(Cdev, k, k, V, n, z), reps)
    return nep.dev.algorithmic_bytes(k), ms


def orth_roofline(na, n, k, reps=20):
    """K6 at the FIXED shape of iar step k (block-triangular basis, rows = n(k+1)): one classical Gram-Schmidt pass =
    k_orth_dots + k_orth_update (+ 2 small kernels).  Algorithmic bytes: SURVEY.md section 8d K6 restricted to the non-zero
    blocks: 2*16*sum_j active_j + 3*16*rows.  Average over `reps` identical asynchronous nep_orth_dev launches."""
    from nep_amd import dense
    rows = n * (k + 1)
    active = (np.arange(1, k + 1) * n).astype(np.int64)
    V = torch.randn((k, rows), dtype=torch.float64, device="cuda").to(torch.complex128)
    w = torch.randn(rows, dtype=torch.float64, device="cuda").to(torch.complex128)
    act_d = torch.from_numpy(active).to("cuda")
    out = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
    ms = event_loop(lambda: dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=act_d,
                                                                  method=dense.CGS), reps, warm=2)
    byts = 2 * 16 * int(active.sum()) + 3 * 16 * rows
    return {"bound": "hbm", "kernel": "K6 nep_orth_dev, one Gram-Schmidt pass (k_orth_dots + k_orth_update + 2 small "
            "kernels) at the fixed shape of iar step k=%d: rows=%d, block-triangular basis" % (k, rows),
            "algorithmic_bytes": byts, "ms_per_pass": ms, "launches_timed": reps, "achieved": byts / ms / 1e6,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS}


def orth_run_weighted(na, n, m, reps=3):
    """K6 over the shapes of a WHOLE iar run (step k = 1 .. m: rows = n (k + 1), block-triangular basis), one nep_orth_dev call
    per shape as the pipeline enqueues it (first pass + the gated second pass, which costs its launches also when the device
    decision switches it off): sum of the algorithmic bytes of the first passes / sum of the HIP-event times.  The fixed-shape
    `roofline` above is this sum's largest term; the steps below k ~ 35 are launch-bound."""
    from nep_amd import dense
    rows_max = n * (m + 1)
    # columns of norm ~ 1 and almost orthogonal, w generic: the DGKS criterion of the first pass is not met, so the second
    # pass is the gated-off one (as in 77 of the 100 steps of the headline run; the other 23 run it for real)
    V = (torch.randn((m, rows_max), dtype=torch.float64, device="cuda") / np.sqrt(rows_max)).to(torch.complex128)
    w = torch.randn(rows_max, dtype=torch.float64, device="cuda").to(torch.complex128)
    out = torch.zeros(m + 2, dtype=torch.complex128, device="cuda")
    tot_b = 0; tot_ms = 0.0; small_ms = 0.0
    for k in range(1, m + 1):
        rows = n * (k + 1)
        active = torch.from_numpy((np.arange(1, k + 1) * n).astype(np.int64)).to("cuda")

        def call():
            dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows_max, active_dev=active, method=dense.DGKS)
        ms = event_loop(call, reps, warm=1)
        b = 2 * 16 * n * (k * (k + 1) // 2) + 3 * 16 * rows
        tot_b += b; tot_ms += ms
        if k <= 35:
            small_ms += ms
    del V
    return {"algorithmic_bytes_first_passes": tot_b, "ms_all_steps": tot_ms, "ms_steps_1_to_35": small_ms,
            "achieved": tot_b / tot_ms / 1e6, "frac": tot_b / tot_ms / 1e6 / HBM_PEAK_GBS,
            "note": "sum over the %d step shapes of one iar run; DGKS as enqueued by the pipeline (gated second pass included)" % m}


def _file_digest(path):
    import hashlib
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def k5_roofline(na, nep, args, reps=100):
    """K5: one fixed-shift solve M(sigma) x = b on the gun matrices; bytes = SURVEY.md section 8d K5
    (nnz L + nnz U)(16 + 4) + 8(n + 1) + 3*16 n; the schedule itself moves `moved_bytes` (explicit block inverses)"""
    import scipy.sparse as sp
    lu = na.DeviceLU(sp.csc_matrix(nep.compute_Mder(0.0), dtype=np.complex128), permc_spec=args.permc, expected_solves=200)
    n = lu.n
    B = torch.randn(n, dtype=torch.float64, device="cuda").to(torch.complex128)
    X = torch.empty_like(B)
    ms = event_loop(lambda: lu.solve(B, out=X), reps, warm=24)      # (the first solves of a factor walk the apex levels, NEP_ML_APEX_AT)
    ba = lu.algorithmic_bytes
    return {"bound": "hbm (dependent-launch latency in practice)", "kernel": "K5 nep_lu_solve, one right-hand side, gun M(sigma): "
            "%s" % ("elimination-tree block schedule, %d levels, %d blocks" % (lu.levels, lu.blocks) if lu.block_schedule
                    else "level schedule"),
            "launches_per_solve": lu.launches_last_solve(), "nnz_L_plus_U": lu.nnzL + lu.nnzU, "algorithmic_bytes": ba,
            "moved_bytes": lu.solve_bytes, "ms_per_solve": ms, "achieved": ba / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ba / ms / 1e6 / HBM_PEAK_GBS, "moved_GBps": lu.solve_bytes / ms / 1e6}


def beyn_sharded(na, args, world, rank):
    """config C4: contour_beyn on the unscaled gun SPMF, N=64 nodes sharded i = r (mod P) over the ranks, one RCCL
    all-gather of the 2 n k partial moment block.  Strong scaling (total work fixed).  Timed with barriers.  Parity on rank
    0: count and eigenvalues against the CPU oracle on the same probe block, backward errors re-evaluated on the host."""
    import baseline_configs as bc
    if os.environ.get("NEP_BENCH_BEYN_FAIL"):          # test hook: the extra fails on every rank, the headline line must survive
        raise RuntimeError("injected failure of the sharded contour_beyn extra (NEP_BENCH_BEYN_FAIL)")
    nep = na.nep_gallery("gun_spmf", args.n)
    nep.dev
    Vh = na.probe_block(nep.n, 32)
    # worker processes for the host factorisations of THIS rank's nodes (64/world of them): no more workers than nodes
    from nep_amd._affinity import cpu_budget
    from nep_amd.linsolvers import _DeviceRefactor
    if not any(p["state"] == "ready" for p in _DeviceRefactor.plans.values()) or os.environ.get("NEP_BEYN_HOST_LU"):
        # (with a plan for the pattern the nodes are factorised on the device in one batch and no worker process is needed)
        na.HostLUPool.warm(max(1, min(16, -(-64 // max(world, 1)), cpu_budget() - 2)))
    distd = dist.is_available() and dist.is_initialized()
    integ = na.MatrixTrapezoidalSharded if distd else na.MatrixTrapezoidal
    if distd and RED_DEV == "cpu":
        na.MatrixTrapezoidalSharded.comm = na.HostStagedComm()
    bc.c4_device(na, nep, integ, Vh=Vh, N=8 * world)          # warm-up (graph capture, allocator pools)
    if distd:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = {}
    lam, V = bc.c4_device(na, nep, integ, Vh=Vh, info=info)
    if distd:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank = None
    if distd:
        mine = {"rank": rank, "nodes": int(info.get("nodes", -1)), "wall_s": round(dt, 5),
                "exchange_s": None if info.get("exchange_s") is None else round(float(info["exchange_s"]), 6)}
        try:          # which part of this rank's share does not shrink with the number of ranks: its (batched) factorisation, instrumented
            pi_ = {"phases_s": {}}
            bc.c4_device(na, nep, integ, Vh=Vh, info=pi_)
            mine["factorise_nodes_s"] = round(float(pi_["phases_s"].get("factorise_nodes", 0.0)), 6)
            mine["solve_nodes_s"] = round(float(pi_["phases_s"].get("solve_nodes_and_accumulate", 0.0)), 6)
        except Exception:
            mine["factorise_nodes_s"] = None
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([dt], dtype=torch.float64, device=RED_DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    # a further, untimed call with a device synchronisation behind every phase: which parts shard (factorise, solve) and which
    # every rank repeats (probe upload, moment download, SVD / eig on the host, eigenvectors, residual filter) -- recorded for
    # one rank as well, so that the serial fraction of the strong-scaling curve is known before an 8-GPU node measures it
    phases = None
    try:
        pinfo = {"phases_s": {}}
        bc.c4_device(na, nep, integ, Vh=Vh, info=pinfo)
        torch.cuda.synchronize()
        ph = {k_: round(float(v), 6) for k_, v in pinfo["phases_s"].items()}
        shard = ph.get("factorise_nodes", 0.0) + ph.get("solve_nodes_and_accumulate", 0.0)
        tot = sum(ph.values())
        phases = {"seconds": ph, "sharded_s": round(shard, 6), "replicated_s": round(tot - shard, 6),
                  "replicated_fraction_of_this_run": round((tot - shard) / tot, 4) if tot > 0 else None,
                  "note": "each phase closed by a device synchronisation (the timed call above has none); with P ranks the sharded "
                          "part divides by P, the replicated part and the exchange do not"}
        if not distd:
            # expected strong-scaling curve from MEASURED pieces of this device (no "sharded / P"): with P ranks a rank factorises and
            # solves 64 / P nodes -- the batched device LU is a chain of ~150 launches whose time depends little on the batch size, so
            # that stage does NOT divide by P; it is run here at the batch sizes an 8-GPU node will see.  Exchange: one all-gather of
            # 2 n k complex128 per rank over xGMI, (P - 1) / P of the gathered block crosses a link at ~50 GB/s effective per
            # direction (ring); the replicated tail is what this run measured.
            pred = {}
            rep_s = tot - shard
            for P in (2, 4, 8):
                pi = {"phases_s": {}}
                bc.c4_device(na, nep, integ, Vh=Vh, info={"phases_s": {}}, N=64 // P)        # (first call at this N: warm-up)
                bc.c4_device(na, nep, integ, Vh=Vh, info=pi, N=64 // P)
                torch.cuda.synchronize()
                f_ = float(pi["phases_s"].get("factorise_nodes", 0.0)); s_ = float(pi["phases_s"].get("solve_nodes_and_accumulate", 0.0))
                exch = (2 * nep.n * 32 * 16) * (P - 1) / 50e9
                tP = f_ + s_ + rep_s + exch
                pred["P%d" % P] = {"factorise_nodes_s": round(f_, 6), "solve_nodes_s": round(s_, 6), "exchange_model_s": round(exch, 6),
                                   "seconds": round(tP, 6), "speedup_vs_1": round(tot / tP, 3) if tP > 0 else None}
            phases["predicted"] = dict(pred, note="factorise / solve measured on THIS GPU at 64 / P nodes (instrumented: a device "
                                       "synchronisation behind each phase), replicated part as measured at P = 1, exchange modelled")
    except Exception as e:
        phases = {"error": repr(e)[:200]}
    if distd:
        dist.barrier()
    out = {"workload": "contour_beyn gun SPMF n=%d N=64 k=32 radius=1e4 sigma=250^2 tol=1e-6" % nep.n,
           "per_rank": per_rank, "wall": "max over ranks", "phases": phases,
           "eigenpairs": int(len(lam)), "seconds": dt, "eigenpairs_per_s": len(lam) / dt, "rank_p": int(info.get("p", -1)),
           "nodes_per_rank": int(info.get("nodes", 64)), "scaling": "strong",
           "exchange": "one all_gather of 2*n*k complex128 per rank (%.1f MB)" % (2 * nep.n * 32 * 16 / 1e6) if distd else "none"}
    if rank == 0:
        errs = bc.c4_host_errors(nep.n, lam, V)
        out["max_backward_error"] = max(errs + [0.0])
        out["inside_contour"] = int(np.sum(abs(np.asarray(lam) - 250.0 ** 2) <= 1e4))   # the rest: accurate pairs just outside, kept last (method_beyncontour.jl:153-163)
        if not args.no_beyn_parity:
            t0 = time.perf_counter(); io = {}
            lo, Vo = bc.c4_oracle(na, nep.n, info=io)
            to = time.perf_counter() - t0
            ok, worst = bc.match(lam, lo, 1e-8)
            out["parity"] = {"oracle_eigenpairs": int(len(lo)), "oracle_rank_p": int(io.get("p", -1)), "oracle_seconds": to,
                             "same_count": len(lo) == len(lam), "eigenvalues_match_1e-8": bool(ok), "max_rel_eig_diff": worst}
    return out


def c3_summary(na, args):
    import baseline_configs as bc
    nep = bc.c3_device_nep(na, args.n)
    info = {}
    bc.c3_device(na, nep, info=info)                       # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lam, X, res = bc.c3_device(na, nep, info=info)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    errs = bc.c3_host_errors(args.n, lam, X)
    out = {"workload": "gun nleigs variant R1 (PEP + LowRankFactorizedNEP, r = %d), maxit=100, leja=0, reusefact=2" % info.get("lowrank_r", -1),
           "eigenpairs": int(len(lam)), "seconds": dt, "eigenpairs_per_s": len(lam) / dt, "factorizations": int(info.get("nfact", -1)),
           "max_backward_error": max(errs + [0.0])}
    t0 = time.perf_counter()
    lo, Xo, ro = bc.c3_oracle(na, args.n)
    to = time.perf_counter() - t0
    ok, worst = bc.match(lam, lo, 1e-8)
    out["parity"] = {"oracle_eigenpairs": int(len(lo)), "oracle_seconds": to, "same_count": len(lo) == len(lam),
                     "eigenvalues_match_1e-8": bool(ok), "max_rel_eig_diff": worst}
    return out


def c5_step_roofline(na, nx=1003, nz=999, N=37, reps=50):
    """config C5's inner loop: ONE preconditioned operator step w = Pl^{-1} S v of the Schur-complement GMRES (Waveguide.jl:398-446,
    the reference's linear solver for this problem) at n = 1e6, timed with HIP events as the solver issues it (one hipGraph replay) and
    in its two halves.  Algorithmic bytes: S v in its matrix-free form = vector and diagonal read, result written (48 N) (assembled
    form, NEP_WEP_STENCIL=0: K1 at k = 1 on the three waveguide matrices, SURVEY.md section 8d, plus the interior vector once more for
    the boundary update); Pl^{-1} = the interior vector read and written once (32 N) -- the transforms and
    tridiagonal sweeps of the Sylvester solve are traffic of the implementation, not of the operation."""
    from nep_amd import wep_linsolvers as wl
    nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); nep.dev
    sigma = -3 - 3.5j
    P = na.wep_generate_preconditioner(nep, N, sigma)
    cr = na.WEPLinSolverCreator(solver_type="gmres", kwargs=(("Pl", P), ("reltol", 1e-9), ("restart", 60), ("maxiter", 300)), refinements=0)
    solver = na.create_linsolver(cr, nep, sigma)
    ops = solver.ops
    Nn = ops.N
    v = torch.randn(Nn, dtype=torch.float64, device="cuda").to(torch.complex128); w = torch.empty_like(v)
    if getattr(ops, "stencil", None) is not None:
        # matrix-free Schur complement (round 4): v and the diagonal read once, the result written once (48 N) + the boundary vectors
        b_op = 48.0 * Nn + 6 * 16.0 * nz
        op_form = "matrix-free five-point stencil + P^{-1} (nep_wep_schur_matvec): 48 bytes per interior unknown"
    else:
        b_op = nep.dev.algorithmic_bytes(1) + 16.0 * Nn
        op_form = "assembled: K1 at k = 1 on the three stacked sparse terms + P^{-1} + C1"
    b_pre = 32.0 * Nn
    ms_op = event_loop(lambda: ops.matvec(v, w), reps, warm=5)
    prec = solver.gmres._Pl_call
    ms_pre = event_loop(lambda: prec(w), reps, warm=5)
    out = {"workload": "WEP JARLEBRING nx=%d nz=%d (N = %d interior unknowns), sigma = -3-3.5i, preconditioner %d x %d regions" % (nx, nz, Nn, N, N + 4),
           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "schur_matvec": {"form": op_form, "algorithmic_bytes": b_op, "ms": ms_op, "achieved": b_op / ms_op / 1e6, "frac": b_op / ms_op / 1e6 / HBM_PEAK_GBS},
           "preconditioner": {"algorithmic_bytes": b_pre, "ms": ms_pre, "achieved": b_pre / ms_pre / 1e6, "frac": b_pre / ms_pre / 1e6 / HBM_PEAK_GBS}}
    # what bounds Pl^{-1} is FP64 arithmetic, not HBM: three prime-factor transforms of nx columns of length nz = N1 N2 by dense
    # symmetric-half stages (csrc/wep.hip: (H1 + H2) real-by-complex multiply-add pairs per entry and stage pair, 8 flops each, H = (N-1)/2)
    # + two tridiagonal sweeps (~40 flops per entry each) + the mm x mm complex matrix-vector product of the SMW correction
    try:
        import ctypes
        info = (ctypes.c_int32 * 4)()
        na._lib.check(na._lib.lib.nep_wep_sylv_info(P.sylv, info))
        N1, N2 = int(info[0]), int(info[1])
        fused3 = bool(getattr(P, "_fused", False))
        ntr = 3 if fused3 else 4
        fl = ntr * nx * nz * 8.0 * ((N1 - 1) / 2 + (N2 - 1) / 2) + 2 * nx * nz * 40.0 + 8.0 * P.mm * P.mm
        out["preconditioner"].update({"transforms": ntr, "flops": fl, "fp64_TFLOPs": fl / ms_pre / 1e9, "fp64_peak_TFLOPs": 78.6,
                                      "fp64_frac": fl / ms_pre / 1e9 / 78.6,
                                      "binding": "FP64 vector arithmetic of the small dense DFT stages + launch chain (9 launches), not HBM: "
                                                 "the 16 MB block stays in the 256 MB last-level cache"})
    except Exception as e:
        out["preconditioner"]["flops_error"] = repr(e)[:120]
    fused = getattr(solver.gmres, "fused_step", None)
    if fused is not None:
        ms_st = event_loop(lambda: fused(v, w), reps, warm=5)
        what = "copy in + hipGraph replay of S v and Pl^{-1} + copy out (what one GMRES iteration launches besides its Gram-Schmidt pass)"
    else:
        def direct():
            ops.matvec(v, w); prec(w)
        ms_st = event_loop(direct, reps, warm=5)
        what = ("S v and Pl^{-1} issued directly: nep_wep_schur_matvec + nep_wep_smw_apply, 11 launches, no staging copies (what one GMRES "
                "iteration launches besides its Gram-Schmidt pass)")
    out["step_as_issued"] = {"what": what, "algorithmic_bytes": b_op + b_pre, "ms": ms_st,
                             "achieved": (b_op + b_pre) / ms_st / 1e6, "frac": (b_op + b_pre) / ms_st / 1e6 / HBM_PEAK_GBS}
    return out



def c5_summary(na, args):
    import baseline_configs as bc
    # the timed run is NOT instrumented (tiar with `timers` synchronises after every phase); the phase split comes from a second run
    t0 = time.perf_counter()
    lam, Q, res, info = bc.c5_device(na, nx=args.c5_nx, nz=args.c5_nz, solver="gmres")
    dt = time.perf_counter() - t0
    tm = {}
    try:
        _, _, _, info_tm = bc.c5_device(na, nx=args.c5_nx, nz=args.c5_nz, solver="gmres", timers=tm)
        tm["_instrumented_run_solver_s"] = info_tm["solve_s"]
    except Exception as e:
        tm = {"error": repr(e)[:200]}
    extra = {}
    try:        # SURVEY.md section 8d rule (ii): every pair re-evaluated in FP64 on the HOST by the oracle's matrix-free operator
        Qh = na.to_host(Q) if not isinstance(Q, np.ndarray) else Q
        hres = bc.c5_host_residuals(args.c5_nx, args.c5_nz, lam, Qh)
        extra["max_residual_host_fp64"] = max(hres + [0.0])
    except Exception as e:
        extra["max_residual_host_fp64"] = repr(e)[:200]
    if not args.no_c5_oracle:
        try:    # CPU baseline + parity of this configuration on its 303 x 299 twin (the oracle's assembled-matrix LU route does
                # not finish in minutes at n = 1e6): device twin and oracle twin by eigenvalue
            t1 = time.perf_counter()
            lt, Qt, rest, it = bc.c5_device(na, 303, 299, solver="lu")
            t_dev_twin = time.perf_counter() - t1
            lo, Qo, t_or = bc.c5_oracle_twin(303, 299)
            ok, worst = bc.match(lt, lo, 1e-8)
            extra["cpu_baseline_twin"] = {"kind": "port", "workload": "the same tiar call on the 303 x 299 twin (n = 91 195), oracle: assembled M(sigma) + SuperLU",
                                          "value": len(lo) / t_or, "unit": "eigenpairs/s", "eigenpairs": int(len(lo)), "seconds": t_or,
                                          "device_twin_eigenpairs": int(len(lt)), "device_twin_seconds_incl_generation": t_dev_twin,
                                          "device_twin_solve_s": it["solve_s"], "same_count": len(lo) == len(lt),
                                          "eigenvalues_match_1e-8": bool(ok), "max_rel_eig_diff": worst}
        except Exception as e:
            extra["cpu_baseline_twin"] = {"error": repr(e)[:300]}
    if args.c5_oracle_full:
        try:
            extra["cpu_baseline_full"] = c5_oracle_full_record(bc, args.c5_nx, args.c5_nz, lam)
        except Exception as e:
            extra["cpu_baseline_full"] = {"error": repr(e)[:300]}
    else:       # the committed record of the last full-size oracle run (scripts/c5_oracle_full.py through gpurun), marked as such
        try:
            with open(os.path.join(ROOT, "profiles", "r4_c5_oracle_full.json")) as f:
                rec = json.load(f)
            if (rec.get("nx"), rec.get("nz")) == (args.c5_nx, args.c5_nz):
                ok, worst = bc.match(lam, [complex(a, b) for a, b in rec["eigenvalues"]], 1e-8)
                extra["cpu_baseline_full"] = {**{k_: rec[k_] for k_ in ("kind", "workload", "value", "unit", "eigenpairs", "seconds_solver", "threads", "host")
                                                 if k_ in rec},
                                              "measured": "recorded run (profiles/r4_c5_oracle_full.json), not this process; --c5-oracle-full re-measures",
                                              "this_run_eigenvalues_match_1e-8": bool(ok), "this_run_max_rel_eig_diff": worst}
        except (OSError, KeyError, ValueError):
            pass
    return {**extra, "workload": "WEP JARLEBRING nx=%d nz=%d (n=%d) tiar sigma=-3-3.5i maxit=60 tol=1e-8, Schur complement + "
                        "Sylvester-SMW preconditioned GMRES (the reference's solver for this problem; 37 x 41 regions, inner reltol 1e-9, one "
                        "refinement sweep)" % (args.c5_nx, args.c5_nz, info["n"]),
            "eigenpairs": int(len(lam)), "max_residual": max(res + [0.0]), "seconds_incl_generation": dt,
            "seconds_solver": info["solve_s"], "eigenpairs_per_s": len(lam) / info["solve_s"],
            "generate_s": info["generate_s"], "preconditioner_setup_s": info.get("preconditioner_setup_s"),
            "phases_s": {k_: (round(v_, 4) if isinstance(v_, float) else v_) for k_, v_ in tm.items()},
            "phases_note": "phase split of a SECOND, instrumented run (a device synchronisation after every phase); seconds_solver is the first, "
                           "uninstrumented run",
            "eigenvalues": [[float(l.real), float(l.imag)] for l in lam[:8]]}


def c5_oracle_full_record(bc, nx, nz, lam_dev=None, N=37, progress=None):
    """the CPU oracle of C5 at full size by the reference's route, as a cpu_baseline object (and, given the device run's
    eigenvalues, SURVEY.md section 8d rules (i)/(iii) at full size)"""
    import platform
    try:
        from threadpoolctl import threadpool_info
        nthreads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] + [1])
    except Exception:
        nthreads = None
    lo, Qo, info = bc.c5_oracle_full(nx, nz, N=N, progress=progress)
    rec = {"kind": "port", "nx": nx, "nz": nz, "n": info["n"],
           "workload": "the same tiar call (maxit 60, tol 1e-8, same start vector) at nx=%d nz=%d by the reference's own route: matrix-free Schur "
                       "complement + Sylvester-SMW preconditioner (%d x %d regions) + GMRES(60) reltol 1e-9 + one refinement sweep, oracle/wep_linsolvers.py"
                       % (nx, nz, N, N + 4),
           "value": len(lo) / info["seconds_solver"], "unit": "eigenpairs/s", "eigenpairs": int(len(lo)),
           "seconds_solver": info["seconds_solver"], "preconditioner_setup_s": info["preconditioner_setup_s"], "tiar_s": info["tiar_s"],
           "generate_s": info["generate_s"], "gmres_iterations_total": int(sum(info["gmres_iterations"])),
           "gmres_solves": len(info["gmres_iterations"]), "threads": nthreads, "cpus_allowed": len(os.sched_getaffinity(0)),
           "host": platform.processor() or platform.machine(),
           "eigenvalues": [[float(l.real), float(l.imag)] for l in lo]}
    if lam_dev is not None:
        ok, worst = bc.match(lam_dev, lo, 1e-8)
        rec.update(same_count=len(lo) == len(lam_dev), **{"eigenvalues_match_1e-8": bool(ok)}, max_rel_eig_diff=worst)
    return rec


def wep_scale_roofline(na):
    """K1 on the waveguide (config C5) matrices, nx=1003 nz=999 (n = 1 003 995, 8.02 M non-zeros): the HBM-bound size
    at which the >= 60 % target of BASELINE.json is meaningful (the gun call moves 2-18 MB and is launch-bound)."""
    from nep_amd import wep
    wd = wep.WaveguideData(1003, 999, "JARLEBRING")
    dev = na.SPMFDevice(wd.big_matrices())
    n = wd.n
    out = {"workload": "WEP JARLEBRING nx=1003 nz=999: 3 real sparse terms, n=%d, nnz=%d" % (n, dev.nnz), "peak": HBM_PEAK_GBS,
           "unit": "GB/s"}
    for k in (1, 8, 60):
        V = torch.randn((k, n), dtype=torch.float64, device="cuda").to(torch.complex128)
        Cdev = na.to_dev(np.random.default_rng(0).standard_normal((k, dev.mt)) + 0j)
        z = torch.empty(n, dtype=torch.complex128, device="cuda")
        ms = event_loop(lambda: dev.mlincomb_dev(Cdev, k, k, V, n, z), 10, warm=3)
        b = dev.algorithmic_bytes(k)
        out["k=%d" % k] = {"algorithmic_bytes": b, "ms_per_launch": ms, "achieved": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS}
        del V
    # K2 (nep_resid_batch: the residuals of all k Ritz pairs of a tiar check in one pass over the matrices, src/errmeasure.jl:128-130,
    # 186-190) at the same size; algorithmic bytes SURVEY.md section 8d K2: matrices + 16 n k (read Q, row-major), norms only
    for k in (8, 60):
        QT = torch.randn((n, k), dtype=torch.float64, device="cuda").to(torch.complex128)
        F = np.random.default_rng(1).standard_normal((dev.mt, k)) + 0j
        o = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
        ms = event_loop(lambda: dev.resid_batch_dev(F, QT, k, k, o), 10, warm=3)
        b = dev.matrix_bytes + 16 * n * k
        out["K2 k=%d" % k] = {"kernel": "nep_resid_batch_dev (row-major Ritz block; k_tile_resid_sp / _spp: super-panels, LDS-DMA tiles)", "algorithmic_bytes": b, "ms_per_launch": ms, "achieved": b / ms / 1e6,
                              "frac": b / ms / 1e6 / HBM_PEAK_GBS}
        del QT
    # the same residual batch on a COLUMN-major Ritz block (nep_resid_batch_cm_dev: what K7 writes with y_rowmajor = 0 and what a Julia
    # host holds; the row-major rows above are what this package's own drivers call)
    try:
        import ctypes as C_
        from nep_amd._lib import lib, hptr, c_vp
        for k in (8, 60):
            Qc = torch.randn((k, n), dtype=torch.float64, device="cuda").to(torch.complex128)
            Fm = np.asfortranarray(np.random.default_rng(1).standard_normal((dev.mt, k)) + 0j)
            oc = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
            call = lambda: lib.nep_resid_batch_cm_dev(dev.h, k, hptr(Fm), c_vp(Qc.data_ptr()), n, -1, c_vp(oc.data_ptr()), None, 0, None)
            if call() == 0:
                ms = event_loop(call, 10, warm=3)
                b = dev.matrix_bytes + 16 * n * k
                out["K2 column-major k=%d" % k] = {"kernel": "nep_resid_batch_cm_dev (column-major Ritz block, the layout of a Julia host; same kernels)", "algorithmic_bytes": b, "ms_per_launch": ms,
                                                   "achieved": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS}
            del Qc
    except Exception as e:
        out["K2 column-major"] = {"error": repr(e)[:200]}
    # K7 (tall-skinny FP64-MFMA GEMM, the Ritz block Q = V Z of tiar / iar: src/method_tiar.jl:188-189, src/method_iar.jl:115) at the
    # tiar m = 60 shape, B resident on the device (nep_gemm_ts_dev: fragment expansion kernel + the GEMM kernel per call), HIP
    # events around 50 back-to-back calls after 30 warm-up calls (the first ~15 ms of FP64-MFMA work after memory-bound kernels run
    # 13 % slower -- clocks ramping -- scripts/diag/k7_timing.py); peak = the dense FP64 matrix rate of MI355X_MICROARCH.md
    try:
        from nep_amd._lib import lib, check, c_vp
        from nep_amd.nep import stream_ptr
        for k in (60, 64):
            Zb = torch.complex(torch.randn((k, n), dtype=torch.float64, device="cuda"), torch.randn((k, n), dtype=torch.float64, device="cuda"))
            Bd = torch.complex(torch.randn((k, k), dtype=torch.float64, device="cuda"), torch.randn((k, k), dtype=torch.float64, device="cuda"))
            Yb = torch.empty((n, k), dtype=torch.complex128, device="cuda")
            ms = event_loop(lambda: check(lib.nep_gemm_ts_dev(c_vp(Zb.data_ptr()), n, n, k, c_vp(Bd.data_ptr()), k, 0, k,
                                                              c_vp(Yb.data_ptr()), k, 1, stream_ptr())), 50, warm=30)
            fl = 8.0 * n * k * k
            out["K7 k=p=%d" % k] = {"kernel": "nep_gemm_ts_dev (k_expand_B + k_gemm_ts_res), row-major Y", "bound": "mfma", "flops": fl,
                                    "algorithmic_bytes": 16.0 * n * 2 * k, "ms_per_launch": ms, "achieved": fl / ms / 1e9,
                                    "peak": MFMA64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / MFMA64_PEAK_TFLOPS,
                                    "hbm_GBps": 16.0 * n * 2 * k / ms / 1e6}
            del Zb, Yb
    except Exception as e:
        out["K7"] = {"error": repr(e)[:200]}
    # HBM bytes of the K2 / K7 launches from the stored PMC passes (scripts/make_profiles_r5.sh: separate --pmc FETCH_SIZE / WRITE_SIZE
    # runs of `bench.py --only wepscale`; 2*FETCH + WRITE per the gfx950 note), with the digest of the kernel source they were taken on
    try:
        wfile = next(f for f in (os.path.join(ROOT, "profiles", "pmc2", "r%d_wepscale_traffic.json" % r_) for r_ in (6, 5)) if os.path.exists(f))
        pj = json.load(open(wfile))
        cur = _file_digest(os.path.join(ROOT, "nonlineareigenproblems.jl_amd", "csrc", "spmv_tile.hip"))
        stale = bool(pj.get("_meta", {}).get("spmv_tile_hip_digest") != cur)

        def traffic_of(prefix, alg):
            # the launches of that kernel whose traffic lies within a factor 2 of this row's algorithmic bytes (one kernel name serves
            # k = 8 and k = 60): their median
            v = []
            for name, d in pj.items():
                if name.startswith(prefix) and isinstance(d, dict):
                    v += [x * 1024.0 * 1024.0 for x in d.get("hbm_MB_launches", [d.get("hbm_MB_per_launch", 0.0)])]
            v = sorted(x for x in v if 0.5 * alg <= x <= 2.0 * alg)
            return v[len(v) // 2] if v else None
        for key, prefix in (("K2 k=8", "k_tile_resid_sp<double, false"), ("K2 k=60", "k_tile_resid_sp<double, false"),
                            ("K2 column-major k=8", "k_tile_resid_cm<double"), ("K2 column-major k=60", "k_tile_resid_sp<double, true")):
            if key in out and isinstance(out[key], dict):
                t = traffic_of(prefix, out[key]["algorithmic_bytes"])
                out[key]["traffic"] = t
                out[key]["traffic_over_algorithmic"] = None if t is None else t / out[key]["algorithmic_bytes"]
                out[key]["traffic_stale"] = stale
                out[key]["traffic_file"] = os.path.relpath(wfile, ROOT)
        mfile = next(f for f in (os.path.join(ROOT, "profiles", "pmc2", "r%d_mfma_counters.json" % r_) for r_ in (6, 5)) if os.path.exists(f))
        mf = json.load(open(mfile))
        for key in ("K7 k=p=60", "K7 k=p=64"):
            if key in out:
                out[key]["mfma_utilisation_pmc"] = {"value": mf.get("k_gemm_ts_res<8, 16, true> grid=131072", {}).get("mfma_utilisation"),
                                                    "file": os.path.relpath(mfile, ROOT) + " (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), k = p = 60)",
                                                    "gemm_hip_digest_then": mf.get("_meta", {}).get("gemm_hip_digest"),
                                                    "gemm_hip_digest_now": _file_digest(os.path.join(ROOT, "nonlineareigenproblems.jl_amd", "csrc", "gemm.hip"))}
    except Exception as e:
        out["pmc_note"] = repr(e)[:200]
    return out


def cold_call_summary():
    """config C2 in a FRESH process (scripts/cold_call.py): what a first call costs -- context, upload, symbolic schedule,
    host factorisation, allocator pools, the device-LU plan being built behind it -- and how many calls it takes to reach the
    steady state the headline `value` is quoted at"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cold_call.py"), "6"], capture_output=True, text=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"error": (r.stderr or r.stdout)[-300:]}
    d = json.loads(line[-1])
    calls = d["calls_ms"]; steady = min(calls[-2:])
    d["cold_call_ms"] = calls[0]
    d["calls_to_steady_state"] = next((i + 1 for i, c in enumerate(calls) if c <= 1.1 * steady), len(calls))
    return d


def cpu2_summary(ms_step):
    """the headline step in a fresh process restricted to TWO CPUs (scripts/cpu2_call.py): the budget a replica has when 8
    ranks share the benchmark box's 16-CPU quota.  Also the same with eig(H_k) on host LAPACK threads (the round-3 route)."""
    import subprocess
    out = {}
    for mode in ("dev", "host"):
        env = dict(os.environ, NEP_IAR_EIG=mode)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cpu2_call.py"), "10", "2"], capture_output=True, text=True,
                           timeout=300, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            out[mode] = {"error": (r.stderr or r.stdout)[-300:]}
        else:
            out[mode] = json.loads(line[-1])
    if "ms_per_call_mean" in out.get("dev", {}):
        out["ms_per_step_cpu2"] = out["dev"]["ms_per_call_mean"]
        out["ratio_to_ms_per_step"] = out["dev"]["ms_per_call_mean"] / ms_step
    return out


def _cpu_budget():
    try:
        from nep_amd._affinity import cpu_budget
        return int(cpu_budget())
    except Exception:
        return None


def host_cores():
    try:
        import subprocess
        return int(subprocess.check_output(["nproc", "--all"]).decode().strip())
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_baseline(args, dev=None):
    """CPU oracle (NumPy/SciPy restatement of the reference path, SuperLU for UMFPACK) on the SAME configuration as the
    timed GPU step (maxit = args.maxit unless --cpu-maxit bounds it), BLAS threads as set for this process; plus
    compute_Mlincomb at k = 100 through the C port (1 thread, the reference's per-term gemv + CSC scatter structure) and its
    OpenMP all-cores variant."""
    import baseline_configs as bc
    from oracle import gallery as og, cref
    try:
        from threadpoolctl import threadpool_info
        nthreads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] + [1])
    except Exception:
        nthreads = int(os.environ.get("OPENBLAS_NUM_THREADS", "1"))
    m = args.cpu_maxit if args.cpu_maxit > 0 else args.maxit
    tm = {}
    t0 = time.perf_counter()
    ohist = []
    lam, Q = bc.c2_oracle(args.n, maxit=m, permc=args.permc or "MMD_AT_PLUS_A", timers=tm, hist=ohist)
    t_iar = time.perf_counter() - t0
    parity = None
    if dev is not None and m == args.maxit:       # SURVEY.md section 8d rules (i), (iii), (iv) on the headline configuration itself
        parity = bc.c2_parity(dev[0], dev[1], lam, ohist)
    onep = og.gun_spmf_scaled(args.n)
    lib = cref.load()
    terms = cref.CscTerms(onep.get_Av())
    k = 100
    rng = np.random.default_rng(0)
    V = np.asfortranarray(rng.standard_normal((args.n, k)) + 1j * rng.standard_normal((args.n, k)))
    Cm = np.asfortranarray(rng.standard_normal((k, terms.mt)) + 0j)
    nnz = sum(A.nnz for A in onep.get_Av())
    byts = nnz * 12 + 4 * terms.mt * (args.n + 1) + 16 * args.n * k + 16 * args.n

    def tloop(f, reps):
        f()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        return (time.perf_counter() - t) / reps
    t_ml = tloop(lambda: cref.mlincomb(lib, terms, Cm, V), 20)
    W = np.empty((args.n, terms.mt), dtype=np.complex128, order="F")
    try:                                   # "all cores" = the CPU budget of this process (cgroup quota), not OMP_NUM_THREADS
        import ctypes
        from nep_amd._affinity import cpu_budget
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(cpu_budget()))
    except Exception:
        pass
    t_omp = tloop(lambda: cref.mlincomb_omp(lib, terms, Cm, V, W), 50)
    return {
        "value": len(lam) / t_iar, "unit": "eigenpairs/s", "cores": int(nthreads), "host_cores": host_cores(),
        "cpu_budget": _cpu_budget(), "kind": "port",
        "sample": "oracle (NumPy/SciPy restatement of the reference path, SuperLU for UMFPACK, %d BLAS threads on a %d-core host) "
                  "iar on the same gun SPMF n=%d, maxit=%d%s: %d eigenpairs in %.2f s (orth %.2f s, mlincomb %.2f s, solve %.2f s, "
                  "residuals %.2f s)" % (nthreads, host_cores(), args.n, m, "" if m == args.maxit else " instead of %d" % args.maxit,
                                         len(lam), t_iar, tm.get("orth", 0), tm.get("mlincomb", 0), tm.get("solve", 0), tm.get("resid", 0)),
        "same_config_as_value": bool(m == args.maxit),
        "eigenpairs": int(len(lam)), "seconds": t_iar, "parity": parity,
        "mlincomb_k100": {"algorithmic_bytes": byts,
                          "c_port_1_thread": {"ms": t_ml * 1e3, "GBps": byts / t_ml / 1e9,
                                              "structure": "per-term gemv + CSC scatter (src/NEPTypes.jl:1006-1007)"},
                          "c_openmp_all_cores": {"ms": t_omp * 1e3, "GBps": byts / t_omp / 1e9, "threads": int(lib.ref_omp_threads()),
                                                 "structure": "row-parallel: per-thread gemv block + CSR products"}},
    }


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks HERE (one process per GPU, torch.distributed.run on
    127.0.0.1 with a free port) and hand its exit code back.  The ranks run this same file with WORLD_SIZE / RANK / LOCAL_RANK set, so
    the code path is the one `python -m torch.distributed.run ... bench.py --gpus N` takes; rank 0 prints the one JSON line."""
    import socket
    import subprocess
    if not os.environ.get("NEP_BENCH_SHARE_GPU") and torch.cuda.device_count() < args.gpus:
        sys.exit("bench.py --gpus %d: this node shows %d GPU(s) (NEP_BENCH_SHARE_GPU=1 rehearses the multi-rank path on one GPU; "
                 "not a measurement)" % (args.gpus, torch.cuda.device_count()))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, NEP_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and os.environ.get("NEP_FORCE_DIST")):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d -- the flag and the launcher disagree" % (args.gpus, world))
    use_dist = world > 1 or bool(os.environ.get("NEP_FORCE_DIST") and "RANK" in os.environ)   # forced: 1-rank RCCL smoke test
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NEP_BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a box with ONE GPU -- every rank computes on cuda:0, the
    # process group is gloo and the contour exchange is staged through host memory (RCCL refuses two ranks on one device).
    # Not a measurement; it exists so that the N > 1 code path of this file can be executed before an 8-GPU node sees it.
    share_gpu = bool(os.environ.get("NEP_BENCH_SHARE_GPU"))
    if use_dist and share_gpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    global RED_DEV
    RED_DEV = "cpu" if (use_dist and share_gpu) else "cuda"
    import nep_amd as na
    import baseline_configs as bc
    assert na.device_count() >= 1, "bench.py needs a GPU: the backend has no CPU fallback"

    nep = na.nep_gallery("gun_spmf_scaled", args.n)
    nep.dev  # build + upload the stacked CSR (inputs resident before the timed region)
    torch.cuda.synchronize()

    if args.only:         # fixed-shape kernel loops: the commands behind profiles/r2_*_kernel_stats.csv
        if args.only == "orth":
            print(json.dumps(orth_roofline(na, nep.n, args.maxit, reps=args.reps)))
        elif args.only == "k5":
            print(json.dumps(k5_roofline(na, nep, args, reps=args.reps)))
        elif args.only == "wepscale":
            print(json.dumps(wep_scale_roofline(na)))
        elif args.only == "c5step":
            print(json.dumps(c5_step_roofline(na)))
        else:
            for k in (args.maxit, 1):
                b, ms = mlincomb_roofline(na, nep, k, reps=args.reps)
                print(json.dumps({"kernel": "K1 nep_mlincomb k=%d" % k, "algorithmic_bytes": b, "ms_per_launch": ms,
                                  "achieved": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS}))
        return

    if use_dist and world > 1 and os.environ.get("NEP_BENCH_SEED_BCAST", "1") != "0":
        # the pattern's one host factorisation (ordering, pivots, fill) on rank 0 only, factors broadcast, plans built per GPU:
        # no rank runs SuperLU in its first call (8 ranks on a shared CPU quota would start 8 of them at once)
        na.seed_plan_from_rank0(nep, 0.0, permc_spec=args.permc)
    warm_ms = []
    for _ in range(args.warmup):
        tw = time.perf_counter()
        bc.c2_device(na, nep, args.maxit, args.permc)
        torch.cuda.synchronize(); warm_ms.append(round((time.perf_counter() - tw) * 1e3, 2))
    # one-time plan of the device-side numeric LU for this sparsity pattern (csrc/lufac.hip): started by the first host
    # factorisation on a background thread (0.3 s); like graph capture and allocator pools it belongs to the warm-up
    from nep_amd.linsolvers import _DeviceRefactor
    tw = time.perf_counter()
    _DeviceRefactor.wait()
    plan_wait_ms = round((time.perf_counter() - tw) * 1e3, 2)
    # steady state: the first calls that factorise on the DEVICE still carry one-offs (first use of the factorisation kernels,
    # refinement hint of the NEP object, pools); `value` is quoted at the steady state -- what the first calls cost is reported
    # separately (`cold_call`).  Settle calls are untimed warm-up like the W above and are listed with it.
    settle_ms = []
    for _ in range(max(0, 4 - args.warmup)):
        tw = time.perf_counter()
        bc.c2_device(na, nep, args.maxit, args.permc)
        torch.cuda.synchronize(); settle_ms.append(round((time.perf_counter() - tw) * 1e3, 2))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    import resource

    def cpu_seconds():
        r = resource.getrusage(resource.RUSAGE_SELF)
        return r.ru_utime + r.ru_stime

    barrier()
    native0 = na.iar.native_runs
    t0 = time.perf_counter(); c0 = cpu_seconds()
    pairs = 0
    for _ in range(args.steps):
        # the timed step returns what the reference's iar returns: eigenvalues AND the n x 46 eigenvector block on the HOST
        # (return_device=False: 7.3 MB device-to-host inside the timed region); the device-resident form is timed separately below
        lam, Qh_last = bc.c2_device(na, nep, args.maxit, args.permc, return_device=False)
        pairs += len(lam)
    barrier()
    dt = time.perf_counter() - t0
    native_steps = na.iar.native_runs - native0
    cpu_s_per_call = (cpu_seconds() - c0) / args.steps        # host CPU of this rank, all threads, per timed step
    nres = max(2, min(5, args.steps))
    torch.cuda.synchronize(); tr0 = time.perf_counter()
    for _ in range(nres):
        lam, Q = bc.c2_device(na, nep, args.maxit, args.permc)
    torch.cuda.synchronize()
    ms_resident = (time.perf_counter() - tr0) / nres * 1e3
    if use_dist:
        t = torch.tensor([dt, float(pairs)], dtype=torch.float64, device=RED_DEV)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); pairs = int(round(float(tsum[1])))

    # the same step with the numeric factorisation on the host (SuperLU) every time, for comparison
    ms_host_lu = None
    if os.environ.get("NEP_LU_DEV", "1") != "0" and not args.only:
        os.environ["NEP_LU_DEV"] = "0"
        try:
            bc.c2_device(na, nep, args.maxit, args.permc)
            torch.cuda.synchronize(); th = time.perf_counter()
            for _ in range(4):
                bc.c2_device(na, nep, args.maxit, args.permc)
            torch.cuda.synchronize(); ms_host_lu = (time.perf_counter() - th) / 4 * 1e3
        finally:
            os.environ["NEP_LU_DEV"] = "1"

    out = None
    if rank == 0:
        # independent parity check of the returned pairs (host FP64, reference residual criterion)
        Qh = Qh_last                     # the host block the last TIMED step returned
        errs = bc.host_backward_errors(nep.get_Av(), nep.get_fv(), lam, Qh)
        maxres = max(errs + [0.0])
        # set-up share: median of 5 separately timed create_linsolver calls (host factorisation + device schedule; the
        # device-side block inverses run asynchronously behind it and are waited for here)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ls = na.create_linsolver(na.FactorizeLinSolverCreator(permc_spec=args.permc, max_factorizations=0), nep, 0.0)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
            del ls
        t_setup = float(np.median(ts))
        # instrumented run: time per phase (step-synchronous loop)
        tm = {}
        bc.c2_device(na, nep, args.maxit, args.permc, timers=tm)
        dev_hist = []
        dev_lam, _ = bc.c2_device(na, nep, args.maxit, args.permc, hist=dev_hist, return_device=False)
        k = args.maxit
        byts, ms = mlincomb_roofline(na, nep, k)
        byts1, ms1 = mlincomb_roofline(na, nep, 1)
        achieved = byts / (ms * 1e-3) / 1e9
        ms_step = dt / args.steps * 1e3
        per_step_pairs = pairs / (args.steps * world)
        out = {
            "metric": "eigenpairs/sec (gun SPMF iar m=%d) + compute_Mlincomb GB/s" % args.maxit,
            "value": pairs / dt, "unit": "eigenpairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": "nep_gallery gun SPMF (n=%d, 4 sparse terms, gun-like stand-in K,M + reference W1,W2), "
                                   "shift_and_scale(250^2, 330^2-220^2), iar sigma=0 maxit=%d neigs=Inf tol=1e-10 "
                                   "check_error_every=1 DGKS umfpack_refinements=10; numeric factorisation of M(sigma) "
                                   "(UMFPACK-like symmetric strategy) inside the step" % (args.n, args.maxit),
                       "parallelism": "replicas x%d (iar does not shard)" % world,
                       "factorization": "numeric LU of M(sigma) inside every step: on the device (csrc/lufac.hip, right-looking on the "
                                        "fill pattern and pivot sequence of the pattern's first host SuperLU factorisation, health "
                                        "word + host fallback) when the plan of the pattern exists (built during warm-up), else host SuperLU",
                       "eigenpairs_per_step": per_step_pairs,
                       "max_backward_error": maxres},
            "returns": "eigenvalues + eigenvectors on the host (download inside the timed step)",
            "ms_per_step_device_resident": ms_resident,     # the same call leaving the eigenvector block in HBM (mean of a few calls after the timed region)
            "value_excl_setup": world * per_step_pairs / max(ms_step * 1e-3 - t_setup, 1e-9),
            "linsolver_setup_ms": t_setup * 1e3,
            "ms_per_step_host_lu": ms_host_lu,
            "cpu_s_per_call": cpu_s_per_call,     # host CPU seconds per timed step (getrusage, all threads of this rank)
            "eig_route": os.environ.get("NEP_IAR_EIG", "dev") + " (eig(H_k) of every step: csrc/hesseig.hip on the device | LAPACK on host threads)",
            "driver": "the whole call after the factorisation is ONE C-ABI call, nep_iar_run (csrc/iar_run.hip): %d of the %d timed steps took it (the "
                      "rest: the step-at-a-time Python pipeline); the same entry point the Julia method iar(::DeviceSPMF) calls"
                      % (native_steps, args.steps),
            "warmup_calls_ms": warm_ms,           # the first call carries every one-off: symbolic schedule, host LU, allocator pools
            "settle_calls_ms": settle_ms,         # extra untimed calls after the device-LU plan became ready (when W < 4)
            "plan_wait_ms": plan_wait_ms,         # time the warm-up still had to wait for the device-LU plan (background thread)
            "compute_Mlincomb_GBps": achieved,
            "roofline_compute_Mlincomb": {"bound": "hbm", "kernel": "nep_mlincomb, k=%d columns" % k,
                         "note": "the kernel BASELINE's metric names; at gun size one call moves 17.8 MB (2.2 us at 8 TB/s), "
                                 "i.e. launch-bound by construction; the HBM-bound size is roofline_wep_scale",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes": byts, "ms_per_launch": ms,
                         "single_vector": {"algorithmic_bytes": byts1, "ms_per_launch": ms1,
                                           "achieved": byts1 / (ms1 * 1e-3) / 1e9}},
            "kernels": {"note": "wall ms per phase of one instrumented (step-synchronous) iar run",
                        **{k_: round(v * 1e3, 3) for k_, v in tm.items()}},
        }
        dev_phases = ("linsolver_setup", "mlincomb", "solve", "orth", "ritz", "resid")
        tot = sum(tm.get(p, 0.0) for p in dev_phases)
        out["phase_share"] = {p: round(tm.get(p, 0.0) / tot, 3) for p in dev_phases} if tot > 0 else {}
        # `roofline` = the kernel pair with the largest share of device time (rocprofv3: k_orth_dots + k_orth_update,
        # profiles/r2_iar_kernel_stats.csv), one pass at a fixed shape, averaged
        try:
            rf = orth_roofline(na, nep.n, args.maxit)
            try:
                # HBM bytes of the two kernels from the PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
                # scripts/pmc_collect.sh, 2*FETCH + WRITE per the gfx950 note).  The file names the digest of csrc/orth.hip it
                # was collected on: a kernel change makes `traffic_stale` true instead of going unnoticed.
                tfile = next((f for f in (os.path.join(ROOT, "profiles", "pmc2", "r%d_gun_traffic.json" % r_) for r_ in (6, 5, 4, 3, 2))
                              if os.path.exists(f)), None)
                pj = json.load(open(tfile))
                kb = 0.0
                for name, d in pj.items():
                    if (name.startswith("k_orth_dots") or name.startswith("k_orth_update")) and isinstance(d, dict) and d.get("hbm_MB_per_launch", 0) > 100:
                        kb += d["hbm_MB_per_launch"] * 1024.0
                rf["traffic"] = kb * 1024.0
                meta = pj.get("_meta", {})
                cur = _file_digest(os.path.join(ROOT, "nonlineareigenproblems.jl_amd", "csrc", "orth.hip"))
                rf["traffic_source"] = {"file": os.path.relpath(tfile, ROOT), "collected_at_commit": meta.get("commit"),
                                        "orth_hip_digest_then": meta.get("orth_hip_digest"), "orth_hip_digest_now": cur,
                                        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (scripts/pmc_collect.sh gun), 2*FETCH + WRITE"}
                rf["traffic_stale"] = bool(meta.get("orth_hip_digest") != cur)
            except Exception:
                rf["traffic"] = None
            try:
                rf["run_weighted"] = orth_run_weighted(na, nep.n, args.maxit)
            except Exception as e:
                rf["run_weighted"] = {"error": repr(e)[:200]}
            rf["share_of_step"] = out["phase_share"].get("orth")
            out["roofline"] = rf
        except Exception as e:
            out["roofline"] = {"error": repr(e)[:200]}
        try:
            out["roofline_k5"] = k5_roofline(na, nep, args)
            out["roofline_k5"]["share_of_step"] = out["phase_share"].get("solve")
        except Exception as e:
            out["roofline_k5"] = {"error": repr(e)[:200]}
        try:                                      # context: what a plain streaming read reaches on this box, same run
            xs = torch.rand(1 << 26, dtype=torch.float64, device="cuda")          # 512 MiB
            ms_s = event_loop(lambda: xs.sum(), 20, warm=2)
            out["stream_read_reference"] = {"kernel": "torch.sum over 512 MiB float64", "GB/s": 8.0 * (1 << 26) / ms_s / 1e6}
            del xs
        except Exception:
            pass
        if world == 1 and not args.no_wep_roofline:
            try:
                out["roofline_wep_scale"] = wep_scale_roofline(na)
            except Exception as e:
                out["roofline_wep_scale"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cold:
            try:
                c2 = cpu2_summary(ms_step)
                out["cpu2"] = c2
                out["ms_per_step_cpu2"] = c2.get("ms_per_step_cpu2")
                cc = cold_call_summary()
                out["cold_call"] = cc
                out["cold_call_ms"] = cc.get("cold_call_ms")
                out["eigenpairs_per_s_cold"] = cc.get("eigenpairs_per_s_cold")
            except Exception as e:
                out["cold_call"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, dev=(dev_lam, dev_hist))
        if world == 1 and not args.no_c3:
            try:
                out["c3_nleigs"] = c3_summary(na, args)
            except Exception as e:
                out["c3_nleigs"] = {"error": repr(e)[:300]}
    # extra (outside the headline timed region): the path that DOES shard -- Beyn's quadrature nodes over the ranks
    beyn = None
    if not args.no_beyn:
        # never lose the headline line because of the extra: an exception is recorded, and with several ranks a watchdog
        # covers the case of a rank stuck in the collective (the multi-rank RCCL exchange cannot be exercised on the 1-GPU
        # development box): after 120 s rank 0 prints the line without the extra and every rank leaves
        watchdog = None
        if world > 1:
            import threading

            def bail():
                if rank == 0:
                    out["beyn_sharded"] = {"error": "sharded contour_beyn extra did not finish within 120 s"}
                    os.write(1, (json.dumps(out) + "\n").encode())
                os._exit(0)
            watchdog = threading.Timer(120.0, bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            beyn = beyn_sharded(na, args, world, rank)
        except Exception as e:
            beyn = {"error": repr(e)[:300]}
        if watchdog is not None:
            watchdog.cancel()
    if rank == 0 and world == 1 and not args.no_c5:
        try:
            out["c5_wep"] = c5_summary(na, args)
            try:
                if args.c5_nx == 1003 and args.c5_nz == 999:
                    out["c5_wep"]["roofline_operator_step"] = c5_step_roofline(na)
            except Exception as e:
                out["c5_wep"]["roofline_operator_step"] = {"error": repr(e)[:300]}
        except Exception as e:
            out["c5_wep"] = {"error": repr(e)[:300]}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["beyn_sharded"] = beyn
        print(json.dumps(out))
    na.HostLUPool.shutdown()


if __name__ == "__main__":
    main()
