import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) if "__file__" in globals() else "/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8"); os.environ.setdefault("OMP_NUM_THREADS", "8")
def stat():
    d = {}
    for ln in open("/sys/fs/cgroup/cpu.stat"):
        k, v = ln.split(); d[k] = int(v)
    return d
import numpy as np, torch
import nep_amd as na
import baseline_configs as bc
bc.c5_device(na, 1003, 999, solver="gmres")
for rep in range(3):
    a = stat(); t = time.perf_counter()
    lam, Q, res, info = bc.c5_device(na, 1003, 999, solver="gmres")
    dt = time.perf_counter() - t; b = stat()
    print("C5 %.2f s (solver %.2f), cpu used %.2f s, throttled periods %d of %d, throttled %.0f ms" % (dt, info["solve_s"], (b["usage_usec"] - a["usage_usec"]) / 1e6, b["nr_throttled"] - a["nr_throttled"], b["nr_periods"] - a["nr_periods"], (b["throttled_usec"] - a["throttled_usec"]) / 1e3), flush=True)
