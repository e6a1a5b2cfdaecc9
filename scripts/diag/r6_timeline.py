"""from a rocprofv3 --kernel-trace csv of repeated headline iar calls (nep_iar_run): for the LAST call, per hardware queue the span
and busy time, the recurrence's per-step duration profile, and what happens after the last Arnoldi step (the tail)"""
import sys, csv, collections
rows = []
with open(sys.argv[1]) as f:
    for x in csv.DictReader(f):
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:48], x.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, x in enumerate(rows) if x[2].startswith("k_iar_start")]
print("kernels", len(rows), "calls", len(starts))
a = starts[-2]; b = starts[-1]          # the second-to-last call: complete, followed by another one
# the call begins with its factorisation, which precedes k_iar_start: back up to the previous call's end
seg = rows[a:b]
t0 = seg[0][0]
fin = [i for i, x in enumerate(seg) if x[2].startswith("k_orth_finish")]
t_rec_end = seg[fin[-1]][1]
mainq = seg[fin[-1]][3]
print("recurrence: first kernel -> last k_orth_finish end: %.2f ms (queue %s)" % ((t_rec_end - t0) / 1e6, mainq))
mq = [x for x in seg if x[3] == mainq and x[0] <= t_rec_end]
busy = sum(x[1] - x[0] for x in mq)
print("  main queue: %d kernels, busy %.2f ms, idle between kernels %.2f ms" % (len(mq), busy / 1e6, (t_rec_end - t0 - busy) / 1e6))
c = collections.Counter(); d = collections.Counter()
for x in mq: c[x[2]] += 1; d[x[2]] += x[1] - x[0]
for name, t in d.most_common(14):
    print("    %-48s x%-5d %.3f ms  avg %.1f us" % (name, c[name], t / 1e6, t / c[name] / 1e3))
# step durations: between consecutive k_orth_finish ends
ends = [seg[i][1] for i in fin]
steps = [(ends[i] - ends[i - 1]) / 1e3 for i in range(1, len(ends))]
for lo in (0, 20, 40, 60, 80, 90):
    s_ = steps[lo:lo + 10]
    if s_: print("  steps %3d..%3d: avg %.0f us per step" % (lo + 2, lo + 1 + len(s_), sum(s_) / len(s_)))
# the call ends where the NEXT call's factorisation begins (its first k_lu_init*): kernels from there on belong to that call
nxt = [x[0] for x in seg if x[0] > t_rec_end and x[2].startswith("k_lu_init")]
t_cut = min(nxt) if nxt else max(x[1] for x in seg) + 1
tail = [x for x in seg if x[1] > t_rec_end and x[0] < t_cut and not x[2].startswith(("k_ml_inverse", "k_ml_gather", "k_gemm_general", "k_apex"))]
t_end = max(x[1] for x in tail) if tail else t_rec_end
print("tail: %.2f ms from the end of the recurrence to the last kernel of the call" % ((t_end - t_rec_end) / 1e6))
byq = collections.defaultdict(list)
for x in tail: byq[x[3]].append(x)
for q, xs in byq.items():
    print("  queue %s: %d kernels, %.2f .. %.2f ms after the recurrence, busy %.2f ms" % (q, len(xs), (min(x[0] for x in xs) - t_rec_end) / 1e6, (max(x[1] for x in xs) - t_rec_end) / 1e6, sum(x[1] - x[0] for x in xs) / 1e6))
    c = collections.Counter(); d = collections.Counter()
    for x in xs: c[x[2]] += 1; d[x[2]] += x[1] - x[0]
    for name, t in d.most_common(6):
        print("      %-48s x%-4d %.3f ms" % (name, c[name], t / 1e6))
# before the recurrence: factorisation etc. of THIS call = kernels between the previous call's last kernel and k_iar_start
prev = rows[starts[-3]:a] if len(starts) >= 3 else []
if prev:
    pf = [i for i, x in enumerate(prev) if x[2].startswith("k_orth_finish")]
    pend = prev[pf[-1]][1]
    pre = [x for x in prev if x[0] > pend and (x[2].startswith("k_lu") or x[2].startswith("k_ml") or x[2].startswith("k_apex") or x[2].startswith("k_fuse") or "gemm" in x[2])]
    if pre:
        print("set-up kernels of the call (factorisation + schedule): %d kernels, span %.2f ms, busy %.2f ms" % (len(pre), (max(x[1] for x in pre) - min(x[0] for x in pre)) / 1e6, sum(x[1] - x[0] for x in pre) / 1e6))
