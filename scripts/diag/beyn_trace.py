import os, sys, time
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
nep = na.nep_gallery("gun_spmf"); nep.dev
Vh = na.probe_block(nep.n, 32)
na.HostLUPool.warm(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
bc.c4_device(na, nep, Vh=Vh, N=16)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    info = {}
    lam, V = bc.c4_device(na, nep, Vh=Vh, info=info)
    torch.cuda.synchronize(); print("C4 %.1f ms, %d pairs" % ((time.perf_counter() - t) * 1e3, len(lam)), flush=True)
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable(); bc.c4_device(na, nep, Vh=Vh); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
na.HostLUPool.shutdown()
