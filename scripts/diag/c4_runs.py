"""N calls of config C4 (contour_beyn on gun, 64 nodes, k = 32) and nothing else -- for kernel statistics / traces"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
import torch
import nep_amd as na
import baseline_configs as bc
nep = na.nep_gallery("gun_spmf"); nep.dev
Vh = na.probe_block(nep.n, 32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(N):
    t0 = time.perf_counter()
    lam, V = bc.c4_device(na, nep, Vh=Vh)
    torch.cuda.synchronize()
    print("call %d: %.2f ms, %d pairs" % (i, (time.perf_counter() - t0) * 1e3, len(lam)), flush=True)
    if i == 0:
        from nep_amd.linsolvers import _DeviceRefactor
        _DeviceRefactor.wait()
# phase split of one more call (each phase closed by a device synchronisation), both tails
for mode in ("1", "0"):
    os.environ["NEP_BEYN_DEVICE_TAIL"] = mode
    bc.c4_device(na, nep, Vh=Vh, info={})
    pinfo = {"phases_s": {}}
    bc.c4_device(na, nep, Vh=Vh, info=pinfo)
    ph = pinfo["phases_s"]
    shard = ph.get("factorise_nodes", 0.0) + ph.get("solve_nodes_and_accumulate", 0.0)
    tot = sum(ph.values())
    print("device tail" if mode == "1" else "host tail", {k_: round(v * 1e3, 3) for k_, v in ph.items()}, "replicated %.1f %%" % (100 * (tot - shard) / tot), "p =", pinfo.get("p"))
