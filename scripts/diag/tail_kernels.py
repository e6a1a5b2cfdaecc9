"""from a rocprofv3 --kernel-trace csv of repeated headline iar calls: what the device executes between the last Arnoldi step of
a call (last k_orth_finish) and the first kernel of the next call's factorisation (k_lu_init), for the calls where that takes long"""
import sys, csv, collections
rows = []
with open(sys.argv[1]) as f:
    r = csv.DictReader(f)
    for x in r:
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:60], x.get("Queue_Id", "?"), x.get("Stream_Id", "?")))
rows.sort()
starts = [i for i, x in enumerate(rows) if x[2].startswith("k_lu_init")]
print("kernels", len(rows), "calls", len(starts))
gaps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    fin = [i for i, x in enumerate(seg) if x[2].startswith("k_orth_finish")]
    if not fin: continue
    last = fin[-1]
    t_fin = seg[last][1]
    tail = seg[last + 1:]
    t_next = rows[b][0]
    gaps.append(((t_next - t_fin) / 1e6, a, tail, (t_fin - seg[0][0]) / 1e6))
gaps.sort(key=lambda g: -g[0])
print("median tail %.2f ms" % sorted(g[0] for g in gaps)[len(gaps) // 2])
for g in gaps[:4]:
    print("--- tail %.1f ms after a %.1f ms recurrence; kernels in the tail:" % (g[0], g[3]))
    c = collections.Counter(); d = collections.Counter(); q = collections.defaultdict(set)
    for x in g[2]:
        c[x[2]] += 1; d[x[2]] += (x[1] - x[0]) / 1e6; q[x[2]].add(x[3])
    for name, n in c.most_common(8):
        print("    %-60s x%-5d %.2f ms  queues %s" % (name, n, d[name], sorted(q[name])))
    if g[2]:
        print("    first tail kernel starts %.2f ms after the last k_orth_finish, last ends %.2f ms after it" % ((g[2][0][0] - (g[2][0][0] if False else rows[g[1]][0])) / 1e6 * 0 + (g[2][0][0] - min(x[0] for x in g[2])) / 1e6, (max(x[1] for x in g[2]) - g[2][0][0]) / 1e6))
