"""device memory in use over repeated C3 (nleigs) and C4 (contour_beyn) calls: should plateau"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 61
def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20
nep4 = na.nep_gallery("gun_spmf"); nep4.dev
Vh = na.probe_block(nep4.n, 32)
na.HostLUPool.warm(8)
for i in range(N):
    lam, V = bc.c4_device(na, nep4, Vh=Vh)
    if i % 10 == 0: print("C4 call %d: %.0f MiB in use, %d pairs" % (i, used(), len(lam)), flush=True)
nep3 = bc.c3_device_nep(na)
for i in range(N):
    lam = bc.c3_device(na, nep3)[0]
    if i % 10 == 0: print("C3 call %d: %.0f MiB in use, %d pairs" % (i, used(), len(lam)), flush=True)
na.HostLUPool.shutdown()
