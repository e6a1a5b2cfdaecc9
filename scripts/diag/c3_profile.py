import os, sys, time, cProfile, pstats, io
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
nep = bc.c3_device_nep(na)
for _ in range(3):
    info = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lam, X, res, *_ = bc.c3_device(na, nep, info=info)
    torch.cuda.synchronize(); print("c3 %.3f s, %d pairs" % (time.perf_counter() - t0, len(lam)), {k: v for k, v in info.items() if not hasattr(v, "__len__")})
pr = cProfile.Profile(); pr.enable()
bc.c3_device(na, nep); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:5500])
