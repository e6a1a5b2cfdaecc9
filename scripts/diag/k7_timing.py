"""K7 through the C ABI at the waveguide shape: does the HIP-event time depend on the measuring loop (repetitions, order, what ran before)?"""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, numpy as np
import nep_amd as na
from nep_amd._lib import lib, check, c_vp
from nep_amd.nep import stream_ptr

n = 1003995
def loop(fn, reps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

bufs = {}
for k in (60, 64):
    Zb = torch.complex(torch.randn((k, n), dtype=torch.float64, device="cuda"), torch.randn((k, n), dtype=torch.float64, device="cuda"))
    Bd = torch.complex(torch.randn((k, k), dtype=torch.float64, device="cuda"), torch.randn((k, k), dtype=torch.float64, device="cuda"))
    Yb = torch.empty((n, k), dtype=torch.complex128, device="cuda")
    bufs[k] = (Zb, Bd, Yb)
def run(k):
    Zb, Bd, Yb = bufs[k]
    check(lib.nep_gemm_ts_dev(c_vp(Zb.data_ptr()), n, n, k, c_vp(Bd.data_ptr()), k, 0, k, c_vp(Yb.data_ptr()), k, 1, stream_ptr()))
for rnd in range(3):
    for k in (60, 64, 60):
        for reps in (20, 100):
            ms = loop(lambda: run(k), reps, 3)
            print(json.dumps({"round": rnd, "k": k, "reps": reps, "ms": round(ms, 4), "TFLOPs": round(8.0 * n * k * k / ms / 1e9, 1),
                              "Z_ptr_mod_4096": Zb.data_ptr() % 4096 if False else bufs[k][0].data_ptr() % 4096, "Y_ptr_mod_4096": bufs[k][2].data_ptr() % 4096}))
