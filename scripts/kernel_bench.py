#!/usr/bin/env python
"""Micro-benchmarks of the individual HIP kernels at BASELINE sizes (HIP-event timing), meant to be run
plain or under `rocprofv3 --kernel-trace --stats` / `rocprofv3 --pmc FETCH_SIZE` (separate passes).

    python scripts/kernel_bench.py [gun|wep|lu|all] [--reps N]

Prints one JSON line per measurement: algorithmic bytes/flops (SURVEY.md section 8d formulas), ms per call,
achieved GB/s or TFLOP/s and the fraction of the MI355X peak (HBM 8 TB/s, FP64 MFMA 78.6 TFLOP/s).
"""
import argparse
import json
import os
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # small host BLAS only; big pools stall the launch thread
    os.environ.setdefault(_v, "8")
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

HBM = 8000.0
MFMA64 = 78.6


def ev_time(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def crandn(*shape):
    return torch.complex(torch.randn(*shape, dtype=torch.float64, device="cuda"),
                         torch.randn(*shape, dtype=torch.float64, device="cuda"))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_mlincomb(na, dev, n, k, label, reps):
    V = crandn(k, n)
    Cm = np.random.default_rng(0).standard_normal((k, dev.mt)) + 0j
    Cdev = na.to_dev(Cm)
    z = torch.empty(n, dtype=torch.complex128, device="cuda")
    ms = ev_time(lambda: dev.mlincomb_dev(Cdev, k, k, V, n, z), reps)
    b = dev.algorithmic_bytes(k)
    emit(kernel="K1 nep_mlincomb (k_vc+k_spmv)", case=label, n=n, k=k, nnz=dev.nnz, bytes=b, ms=ms,
         GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / HBM)


def bench_resid(na, nep, n, k, label, reps):
    QT = crandn(n, k)
    lams = np.linspace(0.1, 0.2, k) + 0.01j
    errm = na.ResidualErrmeasure(nep)
    t0 = time.perf_counter(); errm.batch(list(lams), QT); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        errm.batch(list(lams), QT)
    ms = (time.perf_counter() - t) / reps * 1e3
    b = nep.dev.matrix_bytes + 16 * n * k
    emit(kernel="K2 nep_resid_batch (wall incl. sync)", case=label, n=n, k=k, bytes=b, ms=ms, GBps=b / ms / 1e6,
         frac_hbm=b / ms / 1e6 / HBM)


def bench_orth(na, rows, k, label, reps, active=None):
    V = crandn(k, rows)
    w0 = crandn(rows)
    w = w0.clone()

    # the asynchronous path (nep_orth_dev: what iar / GMRES issue and what bench.py's `roofline` times) so that the PMC passes of
    # scripts/pmc_collect.sh count the kernels of the bench line: k_orth_dots + k_orth_update_rows
    out = torch.empty(k + 2, dtype=torch.complex128, device="cuda")
    act_dev = torch.from_numpy(np.asarray(active, dtype=np.int64)).to("cuda") if active is not None else None

    def run():
        na.dense.copy(w0, w)
        na.dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=act_dev, method=1)    # one CGS pass
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    if active is None:
        b = 2 * 16 * rows * k + 3 * 16 * rows
    else:
        b = 2 * 16 * int(np.sum(np.minimum(active[:k], rows))) + 3 * 16 * rows
    emit(kernel="K6 nep_orth one pass (wall incl. 1 sync + copy)", case=label, rows=rows, k=k, bytes=b, ms=ms,
         GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / HBM)


def bench_gemm(na, rows, k, p, label, reps, rowmajor=True):
    Z = crandn(k, rows)
    B = np.random.default_rng(1).standard_normal((k, p)) + 1j * np.random.default_rng(2).standard_normal((k, p))
    out = torch.empty((rows, p) if rowmajor else (p, rows), dtype=torch.complex128, device="cuda")
    ms = ev_time(lambda: na.gemm_ts(Z, B, rowmajor=rowmajor, out=out), max(reps, 30), warm=30)    # clocks settle after ~15 ms of MFMA work
    fl = 8.0 * rows * k * p
    b = 16.0 * rows * (k + p)
    emit(kernel="K7 nep_gemm_ts (incl. host B expansion + H2D)", case=label, rows=rows, k=k, p=p, flops=fl, bytes=b, ms=ms,
         TFLOPs=fl / ms / 1e9, frac_mfma=fl / ms / 1e9 / MFMA64, GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / HBM)


def bench_gemm_h(na, rows, k, p, label, reps):
    WT = crandn(rows, k); YT = crandn(rows, p)
    t0 = time.perf_counter(); na.dense.gemm_h_rm(WT, YT, rows, k, p)
    t = time.perf_counter()
    for _ in range(reps):
        na.dense.gemm_h_rm(WT, YT, rows, k, p)
    ms = (time.perf_counter() - t) / reps * 1e3
    fl = 8.0 * rows * k * p
    b = 16.0 * rows * (k + p)
    emit(kernel="K9 nep_gemm_h_rm C = W^H Y (wall incl. reduction, D2H, sync)", case=label, rows=rows, k=k, p=p, flops=fl,
         bytes=b, ms=ms, TFLOPs=fl / ms / 1e9, frac_mfma=fl / ms / 1e9 / MFMA64, GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / HBM)


def bench_lu(na, A, label, reps, nrhs=1):
    import scipy.sparse as sp
    t = time.perf_counter()
    lu = na.DeviceLU(sp.csc_matrix(A, dtype=np.complex128), expected_solves=200)
    torch.cuda.synchronize()
    tsetup = time.perf_counter() - t
    n = lu.n
    B = crandn(nrhs, n)
    X = torch.empty_like(B)
    ms = ev_time(lambda: lu.solve(B, out=X), reps)
    xh = na.to_host(X)[:, 0]; bh = na.to_host(B)[:, 0]
    res = float(np.linalg.norm(A @ xh - bh) / np.linalg.norm(bh))
    b = lu.solve_bytes + (nrhs - 1) * 4 * 16 * n
    ba = lu.algorithmic_bytes + (nrhs - 1) * 3 * 16 * n
    sched = (dict(schedule="etree blocks", levels=lu.levels, blocks=lu.blocks, split_levels=lu.split_levels, max_block=lu.mid_block)
             if lu.block_schedule else
             dict(schedule="levels+mid+tail", levels_plain=[lu.levL_full, lu.levU_full], dependent_steps=[lu.levL, lu.levU],
                  tail=lu.tail, mid_rows=lu.mid_rows, mid_block=lu.mid_block))
    # second factorisation of the same pattern: symbolic analysis comes from the pattern cache
    t = time.perf_counter()
    lu2 = na.DeviceLU(sp.csc_matrix(A, dtype=np.complex128), expected_solves=200)
    torch.cuda.synchronize()
    tsetup2 = time.perf_counter() - t
    emit(kernel="K5 nep_lu_solve", case=label, n=n, nrhs=nrhs, nnzL=lu.nnzL, nnzU=lu.nnzU, launches=lu.launches_last_solve(),
         moved_bytes=b, algorithmic_bytes=ba, ms=ms, GBps_algorithmic=ba / ms / 1e6, frac_hbm_algorithmic=ba / ms / 1e6 / HBM,
         GBps_moved=b / ms / 1e6, rel_residual=res, host_factor_s=lu.t_factor, device_setup_s=lu.t_create, setup_s=tsetup,
         second_setup=dict(host_factor_s=lu2.t_factor, device_setup_s=lu2.t_create, total_s=tsetup2), **sched)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import nep_amd as na
    if args.which in ("gun", "all"):
        nep = na.nep_gallery("gun_spmf_scaled")
        n = nep.n
        for k in (1, 10, 100):
            bench_mlincomb(na, nep.dev, n, k, "gun", args.reps)
        bench_resid(na, nep, n, 100, "gun", args.reps)
        bench_orth(na, n * 101, 100, "gun iar step 100 (block-triangular basis)", 5,
                   active=(np.arange(1, 102) * n).astype(np.int64))
        bench_orth(na, n, 100, "gun tiar step 100", args.reps)
        bench_gemm(na, n, 100, 100, "gun Ritz block", args.reps)
        bench_gemm_h(na, n, 100, 100, "gun projection block", args.reps)
        A0 = nep.compute_Mder(0.0)
        bench_lu(na, A0, "gun M(sigma)", args.reps)
        bench_lu(na, A0, "gun M(sigma), Beyn block", args.reps, nrhs=32)
    if args.which == "gunlu":
        nep = na.nep_gallery("gun_spmf_scaled")
        A0 = nep.compute_Mder(0.0)
        bench_lu(na, A0, "gun M(sigma)", args.reps)
        bench_lu(na, A0, "gun M(sigma), Beyn block", args.reps, nrhs=32)
    if args.which in ("wep", "all"):
        from nep_amd import wep
        wd = wep.WaveguideData(1003, 999, "JARLEBRING")
        dev = na.SPMFDevice(wd.big_matrices())
        n = wd.n
        for k in (1, 8, 60):
            bench_mlincomb(na, dev, n, k, "wep 1003x999 (3 sparse terms)", max(3, args.reps // 2))
        bench_orth(na, n, 60, "wep tiar step 60", 5)
        bench_gemm(na, n, 60, 60, "wep Ritz block", 5)
        bench_gemm(na, n, 60, 60, "wep basis block col-major", 5, rowmajor=False)
        bench_gemm_h(na, n, 60, 60, "wep projection block", 5)
    if args.which in ("lu", "all"):
        for nx, nz in ((303, 299), (1003, 999)):
            nepw = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING")
            bench_lu(na, nepw.compute_Mder(-3 - 3.5j), "wep %dx%d M(sigma)" % (nx, nz), 10)
            del nepw


if __name__ == "__main__":
    main()
