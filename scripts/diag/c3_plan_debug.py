import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
nep = bc.c3_device_nep(na)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lam = bc.c3_device(na, nep)[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("c3 %.3f s, %d pairs" % (dt, len(lam)), [(p["state"], p["uses"], p["fails"]) for p in _DeviceRefactor.plans.values()], flush=True)
    if i == 1: _DeviceRefactor.wait()
