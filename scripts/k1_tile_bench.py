"""K1 (compute_Mlincomb): the one-launch footprint-tile kernel (csrc/spmv_tile.hip, mode 1) against the two-launch / folded
forms (mode 2) at waveguide scale (n = 1 003 995) and on the gun matrices; HIP-event averages, algorithmic bytes of SURVEY.md
section 8d.  `python scripts/k1_tile_bench.py [wep|gun|all]`; tile shape through NEP_K1_TILE_XP / NEP_K1_TILE_ZP."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nep_amd as na
from nep_amd._lib import lib, check


def event_loop(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(label, dev, ks, reps):
    n = dev.n
    print(json.dumps({"case": label, "n": n, "nnz": dev.nnz, "tiles": dev.tile_info(), "xp": os.environ.get("NEP_K1_TILE_XP"),
                      "zp": os.environ.get("NEP_K1_TILE_ZP")}), flush=True)
    rng = np.random.default_rng(0)
    for k in ks:
        V = torch.randn((k, n), dtype=torch.float64, device="cuda").to(torch.complex128)
        V = V + 1j * torch.randn((k, n), dtype=torch.float64, device="cuda")
        Cdev = na.to_dev(rng.standard_normal((k, dev.mt)) + 1j * rng.standard_normal((k, dev.mt)))
        out = {}
        zs = {}
        for mode in (2, 1):
            check(lib.nep_k1_set_mode(mode))
            z = torch.empty(n, dtype=torch.complex128, device="cuda")
            ms = event_loop(lambda: dev.mlincomb_dev(Cdev, k, k, V, n, z), reps)
            out["classic" if mode == 2 else "tiles"] = ms
            zs[mode] = z.clone()
        check(lib.nep_k1_set_mode(0))
        b = dev.algorithmic_bytes(k)
        err = float((zs[1] - zs[2]).abs().max() / zs[2].abs().max())
        print(json.dumps({"case": label, "k": k, "algorithmic_bytes": b, "ms_classic": out["classic"], "ms_tiles": out["tiles"],
                          "frac_classic": b / out["classic"] / 1e6 / 8000, "frac_tiles": b / out["tiles"] / 1e6 / 8000,
                          "rel_diff": err}), flush=True)
        del V


def run_k2(label, dev, ks, reps):
    """K2 (nep_resid_batch_dev): tiles against the wave-per-row kernel; algorithmic bytes = matrices + 16 n k"""
    n = dev.n
    rng = np.random.default_rng(1)
    for k in ks:
        QT = torch.randn((n, k), dtype=torch.float64, device="cuda").to(torch.complex128)
        F = rng.standard_normal((dev.mt, k)) + 0j
        o = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
        out = {}; vals = {}
        for mode in (2, 1):
            check(lib.nep_k1_set_mode(mode))
            ms = event_loop(lambda: dev.resid_batch_dev(F, QT, k, k, o), reps)
            out[mode] = ms; vals[mode] = o.cpu().numpy().copy()
        check(lib.nep_k1_set_mode(0))
        b = dev.matrix_bytes + 16 * n * k
        # column-major Q: the tiled kernel whose panel loads are contiguous (nep_resid_batch_cm_dev)
        from nep_amd._lib import hptr, c_vp
        Fm = np.asfortranarray(F)
        Qc = QT.t().contiguous()                     # (k, n): column-major n x k
        oc = torch.zeros(2 * k, dtype=torch.float64, device="cuda")
        rc_cm = lib.nep_resid_batch_cm_dev(dev.h, k, hptr(Fm), c_vp(Qc.data_ptr()), n, -1, c_vp(oc.data_ptr()), None, 0, None)
        if rc_cm == 0:
            ms_cm = event_loop(lambda: lib.nep_resid_batch_cm_dev(dev.h, k, hptr(Fm), c_vp(Qc.data_ptr()), n, -1, c_vp(oc.data_ptr()), None, 0, None), reps)
            print(json.dumps({"case": label + " K2 column-major", "k": k, "algorithmic_bytes": b, "ms_tiles_cm": ms_cm,
                              "frac_tiles_cm": b / ms_cm / 1e6 / 8000,
                              "rel_diff": float(abs(oc.cpu().numpy() - vals[2]).max() / abs(vals[2]).max())}), flush=True)
        del Qc
        print(json.dumps({"case": label + " K2", "k": k, "algorithmic_bytes": b, "ms_classic": out[2], "ms_tiles": out[1],
                          "frac_classic": b / out[2] / 1e6 / 8000, "frac_tiles": b / out[1] / 1e6 / 8000,
                          "rel_diff": float(abs(vals[1] - vals[2]).max() / abs(vals[2]).max())}), flush=True)
        del QT


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("gun", "all"):
    nep = na.nep_gallery("gun_spmf_scaled")
    run("gun", nep.dev, (1, 2, 8, 16, 17, 32, 64, 100), 200)
    run_k2("gun", nep.dev, (10, 50, 100), 50)
if what in ("wep", "all"):
    from nep_amd import wep
    wd = wep.WaveguideData(1003, 999, "JARLEBRING")
    dev = na.SPMFDevice(wd.big_matrices())
    run("wep", dev, tuple(int(x) for x in os.environ.get("NEP_TILE_BENCH_KS", "1,2,4,8,16,32,60").split(",")), 20)
    run_k2("wep", dev, (8, 30, 60), 10)
