"""idle gaps of the device timeline from a rocprofv3 kernel trace: python scripts/diag/trace_gaps.py kernel_trace.csv [t0_frac t1_frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_first = int(rows[0]["Start_Timestamp"]); t_last = int(rows[-1]["End_Timestamp"])
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0; f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lo = t_first + f0 * (t_last - t_first); hi = t_first + f1 * (t_last - t_first)
rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) <= hi]
busy = 0; gaps = collections.Counter(); gapn = collections.Counter(); prev_end = None; prev_name = None
big = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:40]
    if prev_end is not None:
        g = s - prev_end
        if g > 0:
            gaps[(prev_name, name)] += g; gapn[(prev_name, name)] += 1
            if g > 200000: big.append((g, prev_name, name))
        busy += max(0, e - max(s, prev_end))
    else:
        busy += e - s
    prev_end = max(prev_end or e, e); prev_name = name
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels %d span %.2f ms busy %.2f ms idle %.2f ms" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
for (a, b), g in gaps.most_common(18):
    print("  %-42s -> %-42s total %7.2f ms  n %5d  avg %6.2f us" % (a, b, g / 1e6, gapn[(a, b)], g / gapn[(a, b)] / 1e3))
print("gaps > 0.2 ms:", [(round(g / 1e6, 2), a, b) for g, a, b in sorted(big, reverse=True)[:12]])
