"""Per-dispatch analysis of a rocprofv3 --kernel-trace CSV of iar runs: K6 (k_orth_dots / k_orth_update) time against the
algorithmic bytes of every Arnoldi step (run-weighted roofline fraction), and the per-solve launch chain.
usage: python scripts/trace_k6.py <kernel_trace.csv> [n] [maxit]"""
import csv
import json
import sys
from collections import defaultdict

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 9956
m = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
    names[nm][0] += 1; names[nm][1] += d
kname = lambda r: r["Kernel_Name"].replace("void ", "")
dots = [r for r in rows if kname(r).startswith("k_orth_dots")]
upd = [r for r in rows if kname(r).startswith("k_orth_update")]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# group the dots launches into runs of 2*m (two passes per step)
per_run = 2 * m
nrun = len(dots) // per_run
out = {"dots_launches": len(dots), "update_launches": len(upd), "runs": nrun}
if nrun:
    d = dots[-per_run:]; u = upd[-per_run:]
    tot = sum(map(dur, d)) + sum(map(dur, u))
    pass_bytes = lambda k: 2 * 16 * n * (k * (k + 1) // 2) + 3 * 16 * n * (k + 1)
    byts = sum(pass_bytes(k) for k in range(1, m + 1))
    # second passes that actually ran (DGKS criterion met on the device): their bytes are algorithmic too; a gated-off second
    # pass is pure overhead (its kernels exit after reading the flag)
    real2 = [k for k in range(1, m + 1) if dur(d[2 * (k - 1) + 1]) + dur(u[2 * (k - 1) + 1]) > 0.5 * (dur(d[2 * (k - 1)]) + dur(u[2 * (k - 1)]))]
    byts2 = byts + sum(pass_bytes(k) for k in real2)
    ghost = sum(dur(d[2 * (k - 1) + 1]) + dur(u[2 * (k - 1) + 1]) for k in range(1, m + 1) if k not in real2)
    out["last_run"] = {"k6_us": tot, "algorithmic_bytes_first_passes": byts, "run_weighted_frac_first_passes_only": byts / (tot * 1e-6) / 8e12,
                       "second_passes_that_ran": len(real2), "algorithmic_bytes_incl_real_second_passes": byts2,
                       "run_weighted_frac": byts2 / (tot * 1e-6) / 8e12, "gated_off_second_pass_us": ghost}
    steps = []
    for k in range(1, m + 1):
        p1 = dur(d[2 * (k - 1)]) + dur(u[2 * (k - 1)]); p2 = dur(d[2 * (k - 1) + 1]) + dur(u[2 * (k - 1) + 1])
        b = 2 * 16 * n * (k * (k + 1) // 2) + 3 * 16 * n * (k + 1)
        steps.append((k, round(p1, 1), round(p2, 1), round(b / (p1 * 1e-6) / 8e12, 3)))
    out["steps(k, pass1_us, pass2_us, pass1_frac)"] = steps[::5] + [steps[-1]]
out["top"] = sorted(((k, v[0], round(v[1], 1)) for k, v in names.items()), key=lambda t: -t[2])[:30]
print(json.dumps(out))
