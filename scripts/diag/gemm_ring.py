"""K7 at the waveguide shape (1 003 995 x 60 times 60 x 60, B resident in LDS): ring depth 4 vs 8 k-steps, accuracy vs torch"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
rows = 1003995
for (k, p) in ((60, 60), (64, 64), (60, 30)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    Z = torch.randn((k, rows), dtype=torch.complex128, device="cuda", generator=g)          # column-major rows x k
    B = (np.random.default_rng(0).standard_normal((k, p)) + 1j * np.random.default_rng(1).standard_normal((k, p)))
    for rowmajor in (True, False):
        out = None
        for rep in range(3):
            Y = na.gemm_ts(Z, B, rowmajor=rowmajor, out=out); out = Y
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(10):
            na.gemm_ts(Z, B, rowmajor=rowmajor, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ref = (Z[:, :4096].T @ torch.from_numpy(B).cuda())
        got = Y[:4096, :p] if rowmajor else Y[:p, :4096].T
        err = float((got - ref).abs().max() / ref.abs().max())
        fl = 8.0 * rows * k * p
        print("k=%d p=%d rowmajor=%s: %.3f ms  %.1f TFLOP/s  %.2f TB/s  err %.1e" % (k, p, rowmajor, ms, fl / ms / 1e9, 16.0 * rows * (k + p) / ms / 1e9, err), flush=True)
