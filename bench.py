#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X NEP backend (BASELINE.json metric:
"eigenpairs/sec + compute_Mlincomb GB/s, gun SPMF iar m=100").

One "step" = one complete `iar(gun_spmf_scaled, sigma=0, gamma=1, maxit=100, neigs=Inf, v=ones,
tol=1e-10, check_error_every=1)` call on one GPU (config C2 of SURVEY.md section 8d, n = 9956), including
the one-off host factorisation of M(sigma) and the upload of its factors.  The NEP's matrices are
resident in HBM before the timed region.  iar is a sequential Krylov recurrence and does not shard
("replicas only"), so with --gpus N every rank runs an independent replica of the same
workload (weak scaling) and `value` is the whole-job rate: sum over ranks of converged eigenpairs
divided by the max-over-ranks wall time.

The JSON line also carries
  roofline      the kernels that dominate the device time of the timed region: K6, one Gram-Schmidt pass
                (k_orth_dots + k_orth_update) at the shape of the last Arnoldi step; algorithmic bytes / HIP-event
                time against the 8 TB/s HBM peak, HBM traffic from the committed PMC passes
  roofline_compute_Mlincomb   the kernel the metric names (k_vc + k_spmv) at k=100 and k=1 on the gun matrices
                (launch-bound at this size); roofline_wep_scale: the same kernels on the n = 1e6 waveguide matrices
  kernels       per-phase GPU time of one instrumented iar run (so the time-dominant kernel is visible)
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference) on a bounded sample
"""
import argparse
import json
import os
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # small host BLAS only; big pools stall the launch thread
    os.environ.setdefault(_v, "8")
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=9956)
    ap.add_argument("--maxit", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-maxit", type=int, default=60)
    ap.add_argument("--permc", default=None)
    ap.add_argument("--no-beyn", action="store_true", help="skip the sharded contour_beyn extra")
    ap.add_argument("--no-wep-roofline", action="store_true", help="skip the waveguide-scale K1 roofline extra")
    return ap.parse_args()


def one_step(na, nep, args, timers=None, hist=None):
    creator = na.FactorizeLinSolverCreator(permc_spec=args.permc, max_factorizations=0)
    lam, Q, V = na.iar(nep, sigma=0.0, gamma=1.0, maxit=args.maxit, neigs=np.inf, v=np.ones(nep.n), tol=1e-10,
                       linsolvercreator=creator, timers=timers, errhist=hist, return_device=True)
    return lam, Q


def mlincomb_roofline(na, nep, k, reps=50):
    """algorithmic bytes / average HIP-event time of nep_mlincomb at k columns"""
    n = nep.n
    V = torch.randn((k, n), dtype=torch.float64, device="cuda").to(torch.complex128)
    V = V + 1j * torch.randn((k, n), dtype=torch.float64, device="cuda")
    fD = np.column_stack([f.derivs(0.0, k + 1) for f in nep.get_fv()])
    Cm = fD[1:k + 1] / np.arange(1, k + 1)[:, None]
    z = torch.empty(n, dtype=torch.complex128, device="cuda")
    Cdev = na.to_dev(Cm)          # coefficient table resident on the device, as in iar (nep_mlincomb_dev)
    for _ in range(5):
        nep.dev.mlincomb_dev(Cdev, k, k, V, n, z)
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        nep.dev.mlincomb_dev(Cdev, k, k, V, n, z)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    byts = nep.dev.algorithmic_bytes(k)
    return byts, ms


def orth_roofline(na, n, k, reps=10):
    """K6 at the shape of iar step k (block-triangular basis, rows = n(k+1)): one classical Gram-Schmidt pass =
    k_orth_dots + k_orth_update, the two kernels with the largest share of device time in the timed region
    (profiles/r1_iar_kernel_stats_v5.csv).  Algorithmic bytes: SURVEY.md section 8d K6 restricted to the non-zero
    blocks: 2*16*sum_j active_j + 3*16*rows.  Timed with HIP events around asynchronous nep_orth_dev launches."""
    from nep_amd import dense
    rows = n * (k + 1)
    active = (np.arange(1, k + 1) * n).astype(np.int64)
    V = torch.randn((k, rows), dtype=torch.float64, device="cuda").to(torch.complex128)
    w0 = torch.randn(rows, dtype=torch.float64, device="cuda").to(torch.complex128)
    w = w0.clone()
    act_d = torch.from_numpy(active).to("cuda")
    out = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
    for _ in range(2):
        dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=act_d, method=dense.CGS)
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=act_d, method=dense.CGS)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    byts = 2 * 16 * int(active.sum()) + 3 * 16 * rows
    return {"bound": "hbm", "kernel": "K6 nep_orth_dev, one Gram-Schmidt pass (k_orth_dots + k_orth_update + 3 small "
            "kernels) at iar step k=%d: rows=%d, block-triangular basis" % (k, rows), "algorithmic_bytes": byts,
            "ms_per_pass": ms, "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS}


def beyn_sharded(na, args, world, rank):
    """config C4: contour_beyn on the unscaled gun SPMF, N=64 nodes sharded i = r (mod P) over the ranks, one RCCL
    all-gather of the 2 n k partial moment block.  Strong scaling (total work fixed).  Timed with barriers."""
    import torch.distributed as dist
    nep = na.nep_gallery("gun_spmf", args.n)
    nep.dev
    Vh = na.probe_block(nep.n, 32)
    # worker processes for the host factorisations of THIS rank's nodes (64/world of them): no more workers than nodes
    na.HostLUPool.warm(max(2, min(16, -(-64 // max(world, 1)))))
    distd = dist.is_available() and dist.is_initialized()
    integ = na.MatrixTrapezoidalSharded if distd else na.MatrixTrapezoidal
    kw = dict(sigma=250.0 ** 2, radius=1e4, N=64, k=32, neigs=10 ** 6, tol=1e-6, sanity_check=True, Vh=Vh)
    na.contour_beyn(nep, integ, **dict(kw, N=8 * world))          # warm-up (graph capture, allocator pools)
    if distd:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = {}
    lam, V = na.contour_beyn(nep, integ, info=info, **kw)
    if distd:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distd:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    return {"workload": "contour_beyn gun SPMF n=%d N=64 k=32 radius=1e4 sigma=250^2 tol=1e-6" % nep.n,
            "eigenpairs": int(len(lam)), "seconds": dt, "eigenpairs_per_s": len(lam) / dt, "rank_p": int(info.get("p", -1)),
            "nodes_per_rank": int(info.get("nodes", 64)), "scaling": "strong",
            "exchange": "one all_gather of 2*n*k complex128 per rank (%.1f MB)" % (2 * nep.n * 32 * 16 / 1e6) if distd else "none"}


def wep_scale_roofline(na):
    """K1 on the waveguide (config C5) matrices, nx=1003 nz=999 (n = 1 003 995, 8.02 M non-zeros): the HBM-bound size
    at which the >= 60 % target of BASELINE.json is meaningful (the gun call moves 2-18 MB and is launch-bound)."""
    from nep_amd import wep
    wd = wep.WaveguideData(1003, 999, "JARLEBRING")
    dev = na.SPMFDevice(wd.big_matrices())
    n = wd.n
    out = {"workload": "WEP JARLEBRING nx=1003 nz=999: 3 real sparse terms, n=%d, nnz=%d" % (n, dev.nnz), "peak": HBM_PEAK_GBS,
           "unit": "GB/s"}
    for k in (1, 60):
        V = torch.randn((k, n), dtype=torch.float64, device="cuda").to(torch.complex128)
        Cdev = na.to_dev(np.random.default_rng(0).standard_normal((k, dev.mt)) + 0j)
        z = torch.empty(n, dtype=torch.complex128, device="cuda")
        for _ in range(3):
            dev.mlincomb_dev(Cdev, k, k, V, n, z)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dev.mlincomb_dev(Cdev, k, k, V, n, z)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        b = dev.algorithmic_bytes(k)
        out["k=%d" % k] = {"algorithmic_bytes": b, "ms_per_launch": ms, "achieved": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS}
        del V
    return out


def cpu_baseline(args):
    """CPU oracle on a bounded sample of the same workload: iar with maxit=cpu_maxit (DGKS cost grows
    ~ m^3, the full m=100 run needs ~1 min on 8 cores) + the C port of compute_Mlincomb at k=100."""
    from oracle import gallery as og, solvers as osol, neps as oneps, cref
    try:
        from threadpoolctl import threadpool_info
        nthreads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        nthreads = os.cpu_count()
    onep = og.gun_spmf_scaled(args.n)
    m = args.cpu_maxit
    t0 = time.perf_counter()
    der = oneps.DerSPMF(onep, 0.0, m)
    tm = {}
    creator = osol.FactorizeLinSolverCreator(permc_spec=args.permc or "COLAMD")
    lam, Q, _ = osol.iar(der, sigma=0.0, gamma=1.0, maxit=m, neigs=np.inf, v=np.ones(args.n), tol=1e-10,
                         errmeasure=osol.StandardSPMFErrmeasure(onep), linsolvercreator=creator, timers=tm)
    t_iar = time.perf_counter() - t0
    # C port (single thread) of compute_Mlincomb, reference structure, k = 100
    lib = cref.load()
    terms = cref.CscTerms(onep.get_Av())
    k = 100
    rng = np.random.default_rng(0)
    V = np.asfortranarray(rng.standard_normal((args.n, k)) + 1j * rng.standard_normal((args.n, k)))
    Cm = np.asfortranarray(rng.standard_normal((k, terms.mt)) + 0j)
    cref.mlincomb(lib, terms, Cm, V)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        cref.mlincomb(lib, terms, Cm, V)
    t_ml = (time.perf_counter() - t0) / reps
    nnz = sum(A.nnz for A in onep.get_Av())
    byts = nnz * 12 + 4 * terms.mt * (args.n + 1) + 16 * args.n * k + 16 * args.n
    return {
        "value": len(lam) / t_iar, "unit": "eigenpairs/s", "cores": int(nthreads), "kind": "port",
        "sample": "oracle (NumPy/SciPy restatement of the reference path, SuperLU for UMFPACK) iar on the same gun "
                  "SPMF n=%d with maxit=%d instead of %d: %d eigenpairs in %.2f s (orth %.2f s, mlincomb %.2f s, "
                  "solve %.2f s, residuals %.2f s)" % (args.n, m, args.maxit, len(lam), t_iar, tm.get("orth", 0),
                                                       tm.get("mlincomb", 0), tm.get("solve", 0), tm.get("resid", 0)),
        "mlincomb_GBps_k100": byts / t_ml / 1e9, "mlincomb_ms_k100": t_ml * 1e3,
        "mlincomb_kind": "C port, 1 thread, per-term gemv + CSC scatter (src/NEPTypes.jl:1006-1007)",
        "eigenpairs": int(len(lam)), "seconds": t_iar,
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or bool(os.environ.get("NEP_FORCE_DIST") and "RANK" in os.environ)   # forced: 1-rank RCCL smoke test
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    import nep_amd as na
    assert na.device_count() >= 1, "bench.py needs a GPU: the backend has no CPU fallback"

    nep = na.nep_gallery("gun_spmf_scaled", args.n)
    nep.dev  # build + upload the stacked CSR (inputs resident before the timed region)
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(na, nep, args)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    pairs = 0
    for _ in range(args.steps):
        lam, Q = one_step(na, nep, args)
        pairs += len(lam)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt, float(pairs)], dtype=torch.float64, device="cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); pairs = int(round(float(tsum[1])))

    out = None
    if rank == 0:
        # independent parity check of the returned pairs (host FP64, reference residual criterion)
        Qh = na.to_host(Q)
        Av = nep.get_Av(); fv = nep.get_fv()
        maxres = 0.0
        fro = nep.fro_norms()
        for s in range(len(lam)):
            r = sum(f(lam[s]) * (A @ Qh[:, s]) for A, f in zip(Av, fv))
            den = sum(c * abs(f(lam[s])) for c, f in zip(fro, fv)) * np.linalg.norm(Qh[:, s])
            maxres = max(maxres, np.linalg.norm(r) / den)
        # instrumented run: GPU time per phase
        tm = {}
        one_step(na, nep, args, timers=tm)
        k = args.maxit
        byts, ms = mlincomb_roofline(na, nep, k)
        byts1, ms1 = mlincomb_roofline(na, nep, 1)
        achieved = byts / (ms * 1e-3) / 1e9
        # HBM traffic of the same two kernels from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE runs of scripts/pmc_k1.py; FETCH_SIZE doubled per the gfx950 correction of
        # MI355X_MICROARCH.md).  Recorded measurement, not live: PMC collection needs rocprofv3 around the process.
        traffic = None; traffic_src = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_k1_traffic.json")))["gun"]
            kb = 0.0
            for name, d in pj.items():
                if name.startswith("k_vc") or name.startswith("k_spmv grid"):
                    kb += 2 * d["FETCH_SIZE"]["avg_KB"] + d["WRITE_SIZE"]["avg_KB"]
            traffic = kb * 1024.0
            traffic_src = "profiles/r1_pmc_k1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, k=100, FETCH x2 gfx950 correction)"
        except Exception:
            pass
        out = {
            "metric": "eigenpairs/sec (gun SPMF iar m=%d) + compute_Mlincomb GB/s" % args.maxit,
            "value": pairs / dt, "unit": "eigenpairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": "nep_gallery gun SPMF (n=%d, 4 sparse terms, gun-like stand-in K,M + reference W1,W2), "
                                   "shift_and_scale(250^2, 330^2-220^2), iar sigma=0 maxit=%d neigs=Inf tol=1e-10 "
                                   "check_error_every=1 DGKS umfpack_refinements=10; host SuperLU factorisation "
                                   "(UMFPACK-like symmetric strategy) inside the step" % (args.n, args.maxit),
                       "parallelism": "replicas x%d (iar does not shard)" % world,
                       "eigenpairs_per_step": pairs / (args.steps * world),
                       "max_backward_error": maxres},
            "compute_Mlincomb_GBps": achieved,
            "roofline_compute_Mlincomb": {"bound": "hbm", "kernel": "nep_mlincomb = k_vc + k_spmv, k=%d columns" % k,
                         "note": "the kernel BASELINE's metric names; at gun size one call moves 17.8 MB (2.2 us at 8 TB/s) "
                                 "behind two kernel boundaries, i.e. launch-bound by construction; the HBM-bound size is "
                                 "roofline_wep_scale",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": byts, "ms_per_launch": ms,
                         "single_vector": {"algorithmic_bytes": byts1, "ms_per_launch": ms1,
                                           "achieved": byts1 / (ms1 * 1e-3) / 1e9}},
            "kernels": {"note": "wall ms per phase of one instrumented iar run (torch.cuda.synchronize around each phase)",
                        **{k_: round(v * 1e3, 3) for k_, v in tm.items()}},
        }
        # `roofline` = the dominant kernels of the timed region: K6 (k_orth_dots + k_orth_update hold the largest share
        # of device time, profiles/r1_iar_kernel_stats_v5.csv), measured live at the shape of the last step
        try:
            rf = orth_roofline(na, nep.n, args.maxit)
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc2", "gun_traffic.json")))
                kb = 0.0
                for name, d in pj.items():
                    if (name.startswith("k_orth_dots") or name.startswith("k_orth_update")) and d.get("hbm_MB_per_launch", 0) > 100:
                        kb += d["hbm_MB_per_launch"] * 1024.0
                rf["traffic"] = kb * 1024.0
                rf["traffic_source"] = ("profiles/pmc2/gun_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                        "scripts/kernel_bench.py gun, same shape; 2*FETCH + WRITE per the gfx950 note)")
            except Exception:
                rf["traffic"] = None
            try:                                      # context: what a plain streaming read reaches on this box, same run
                xs = torch.rand(1 << 26, dtype=torch.float64, device="cuda")          # 512 MiB
                xs.sum(); torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    xs.sum()
                e1.record(); torch.cuda.synchronize()
                rf["stream_read_reference"] = {"kernel": "torch.sum over 512 MiB float64", "GB/s": 8.0 * (1 << 26) * 20 / e0.elapsed_time(e1) / 1e6}
                del xs
            except Exception:
                pass
            out["roofline"] = rf
        except Exception as e:
            out["roofline"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_wep_roofline:
            try:
                out["roofline_wep_scale"] = wep_scale_roofline(na)
            except Exception as e:
                out["roofline_wep_scale"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    # extra (outside the headline timed region): the path that DOES shard -- Beyn's quadrature nodes over the ranks
    beyn = None
    if not args.no_beyn:
        try:
            beyn = beyn_sharded(na, args, world, rank)
        except Exception as e:                       # never lose the headline line because of the extra
            beyn = {"error": repr(e)[:300]}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["beyn_sharded"] = beyn
        print(json.dumps(out))
    na.HostLUPool.shutdown()


if __name__ == "__main__":
    main()
