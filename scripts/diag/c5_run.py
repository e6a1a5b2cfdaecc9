import os, sys, time, json
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1003
for rep in range(2):
    tm = {}
    t = time.perf_counter()
    lam, Q, res, info = bc.c5_device(na, nx, nx - 4, solver="gmres", timers=tm)
    print(json.dumps(dict(total_s=time.perf_counter() - t, eigenpairs=len(lam), max_res=max(res), solve_s=info["solve_s"],
                          prec_setup_s=info.get("preconditioner_setup_s"), phases={k: round(v, 3) for k, v in tm.items()})), flush=True)
