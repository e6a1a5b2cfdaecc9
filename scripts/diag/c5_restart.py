"""C5 at full size: GMRES restart length and preconditioner region count"""
import os, sys, time, json
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
bc.c5_device(na, 303, 299, solver="gmres", N=13)
for (N, restart) in ((37, 60), (37, 100), (37, 150), (27, 60), (27, 100), (111, 60)):
    tm = {}
    t = time.perf_counter()
    try:
        lam, Q, res, info = bc.c5_device(na, 1003, 999, solver="gmres", N=N, restart=restart, timers=tm)
        print(json.dumps(dict(N=N, restart=restart, total_s=time.perf_counter() - t, eigenpairs=len(lam), max_res=max(res + [0]), solve_s=info["solve_s"],
                              prec_s=info.get("preconditioner_setup_s"), solve_phase=round(tm.get("solve", 0), 3), orth=round(tm.get("orth", 0), 3))), flush=True)
    except Exception as e:
        print("N", N, "restart", restart, "failed:", repr(e)[:200], flush=True)
