import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20
for i in range(6):
    lam, Q, res, info = bc.c5_device(na, 1003, 999, solver="gmres")
    print("C5 call %d: %.0f MiB in use, %d pairs" % (i, used(), len(lam)), flush=True)
