"""a few headline iar calls (for kernel traces): python scripts/diag/iar_one.py [calls]"""
import os, sys, time
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nep_amd as na, torch
import baseline_configs as bc
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    t0 = time.perf_counter(); lam, Q = bc.c2_device(na, nep); torch.cuda.synchronize(); print("call ms %.2f pairs %d" % ((time.perf_counter() - t0) * 1e3, len(lam)))
