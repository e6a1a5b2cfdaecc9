"""One run of the CPU oracle of config C5 at FULL size (nx = 1003, nz = 999, n = 1 003 995) by the reference's own route
for this problem, next to the device run of the same call: the cpu_baseline of C5 and SURVEY.md section 8d rules (i)/(iii)
at full size.  Tens of minutes of host time; run through gpurun, the record goes to gpurun_out/ and is committed as
profiles/r4_c5_oracle_full.json (+ the eigenvalues as tests/golden/c5_full_oracle_eigs.json).

    python scripts/c5_oracle_full.py [nx nz N]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import baseline_configs as bc
import bench

nx, nz, N = (int(a) for a in (sys.argv[1:4] + ["1003", "999", "37"][len(sys.argv) - 1:]))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
lam_dev = None; dev = {}
import torch
if torch.cuda.is_available():
    import nep_amd as na
    bc.c5_device(na, nx, nz, solver="gmres", N=N)                 # warm-up (module loads, plans)
    lam_dev, Q, res, info = bc.c5_device(na, nx, nz, solver="gmres", N=N)
    dev = {"device_eigenpairs": int(len(lam_dev)), "device_seconds_solver": info["solve_s"], "device_max_residual": max(res),
           "device_eigenvalues": [[float(l.real), float(l.imag)] for l in lam_dev]}
    print("device:", dev, flush=True)
t0 = time.perf_counter()


def progress(nsolves, its):
    print("  %7.1f s: GMRES solves %d, iterations of the last %s" % (time.perf_counter() - t0, nsolves, its), flush=True)


rec = bench.c5_oracle_full_record(bc, nx, nz, lam_dev, N=N, progress=progress)
rec.update(dev)
if lam_dev is not None:
    rec["speedup_device_over_cpu"] = rec["seconds_solver"] / dev["device_seconds_solver"]
with open(os.path.join(out, "r4_c5_oracle_full.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
