"""Practical HBM streaming rates of this MI355X with plain torch kernels (context for the roofline fractions):
read-only reduction, copy, and a*x+y, on 2 GiB float64 arrays.  Usage: python scripts/diag/hbm_stream.py"""
import time, json
import torch

n = 1 << 28                       # 2 GiB of float64
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.rand(n, dtype=torch.float64, device="cuda")
def t(f, reps=20):
    f(); torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
out = {}
ms = t(lambda: x.sum());                 out["read (sum)"] = 8 * n / ms / 1e6
ms = t(lambda: torch.dot(x, y));         out["read 2 arrays (dot)"] = 16 * n / ms / 1e6
ms = t(lambda: y.copy_(x));              out["copy (read + write)"] = 16 * n / ms / 1e6
ms = t(lambda: y.add_(x, alpha=1.5));    out["axpy (2 reads + 1 write)"] = 24 * n / ms / 1e6
print(json.dumps({k: "%.0f GB/s" % v for k, v in out.items()}))
