import os, sys, time, json
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
bc.c5_device(na, 303, 299, solver="gmres", N=13)
for (reltol, refine) in ((1e-6, 10), (1e-8, 10), (1e-10, 0), (1e-11, 0), (1e-9, 1), (1e-7, 2)):
    tm = {}
    t = time.perf_counter()
    lam, Q, res, info = bc.c5_device(na, 1003, 999, solver="gmres", reltol=reltol, refine=refine, timers=tm)
    print(json.dumps(dict(reltol=reltol, refine=refine, total_s=time.perf_counter() - t, eigenpairs=len(lam), max_res=max(res + [0]), solve_s=info["solve_s"],
                          solve_phase=round(tm.get("solve", 0), 3))), flush=True)
