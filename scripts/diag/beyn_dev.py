"""C4 with the batched device factorisation vs the host pool"""
import os, sys, time
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
from nep_amd.linsolvers import _DeviceRefactor
nep = na.nep_gallery("gun_spmf"); nep.dev
na.HostLUPool.warm(14)
Vh = na.probe_block(nep.n, 32)
# plan: one host factorisation of a matrix of this pattern
lu0 = na.DeviceLU(nep.compute_Mder(250.0 ** 2))
_DeviceRefactor.wait()
print("plans", [(p["state"], p["uses"]) for p in _DeviceRefactor.plans.values()])
for mode in ("dev", "host", "dev", "host"):
    if mode == "host": os.environ["NEP_BEYN_HOST_LU"] = "1"
    else: os.environ.pop("NEP_BEYN_HOST_LU", None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = {}
    lam, V = bc.c4_device(na, nep, Vh=Vh, info=info)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "%.3f s" % dt, len(lam), "pairs, p =", info.get("p"), "max |lam| diff vs first:", None if mode == "dev" and "ref" not in globals() else float(np.abs(np.sort_complex(lam) - ref).max()))
    if "ref" not in globals(): ref = np.sort_complex(lam)
print("plans", [(p["state"], p["uses"], p["fails"]) for p in _DeviceRefactor.plans.values()])
