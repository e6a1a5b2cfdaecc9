"""device idle time in a rocprofv3 kernel trace: union of the kernel intervals (all queues), the gaps between them, and which
kernel FOLLOWS the long gaps (what the device was waiting for).  python scripts/diag/trace_idle.py <kernel_trace.csv> [t0_ms t1_ms]"""
import csv, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows)
T0 = iv[0][0]
lo = float(sys.argv[2]) * 1e6 + T0 if len(sys.argv) > 2 else iv[0][0]
hi = float(sys.argv[3]) * 1e6 + T0 if len(sys.argv) > 3 else max(e for _, e, _ in iv)
iv = [x for x in iv if x[0] >= lo and x[1] <= hi]
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]
gaps = []
for s, e, nm in iv[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, nm, (cur_e - T0) / 1e6)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0] if len(iv) else 1
print("span %.1f ms, busy %.1f ms (%.0f %%), %d kernels, %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(iv), len(gaps)))
for thr in (5e3, 20e3, 100e3, 1e6):
    g = [x for x in gaps if x[0] >= thr]
    print("gaps >= %5.0f us: %5d, total %.1f ms" % (thr / 1e3, len(g), sum(x[0] for x in g) / 1e6))
c = Counter(); t = Counter()
for d, nm, _ in gaps:
    if d >= 20e3:
        c[nm] += 1; t[nm] += d
print("kernel that follows a gap >= 20 us: count, total ms")
for nm, _ in t.most_common(14):
    print("  %-40s %5d %8.1f" % (nm, c[nm], t[nm] / 1e6))
