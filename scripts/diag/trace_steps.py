"""kernel sequence of single Arnoldi steps from a rocprofv3 kernel trace of repeated headline calls:
python scripts/diag/trace_steps.py kernel_trace.csv [step ...]   (steps of the LAST run; default 10 50 90)
A step is what the recurrence's queue runs between two consecutive k_orth_finish kernels; kernels of other queues that overlap are listed
with a leading '|'."""
import csv, sys, os
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [int(a) for a in sys.argv[2:] if not a.startswith("-")] or [10, 50, 90]
fin = [i for i, r in enumerate(rows) if r["Kernel_Name"].replace("void ", "").startswith("k_orth_finish")]
# runs: groups of 100 finishes
nrun = len(fin) // 100
base = (nrun - 1) * 100
qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
mainq = rows[fin[base]][qkey] if qkey else None
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]
for s in steps:
    i0, i1 = fin[base + s - 2], fin[base + s - 1]
    t0 = int(rows[i0]["End_Timestamp"])
    print("== step %d: %.1f us from the end of the previous k_orth_finish to the end of this one" % (s, (int(rows[i1]["End_Timestamp"]) - t0) / 1e3))
    prev_end = t0; busy = 0; n = 0
    for r in rows[i0 + 1:i1 + 1]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if qkey and r[qkey] != mainq:
            print("   | %-44s start %7.1f dur %6.1f  (queue %s)" % (short(r["Kernel_Name"]), (st - t0) / 1e3, (en - st) / 1e3, r[qkey]))
            continue
        print("  %-46s start %7.1f gap %5.1f dur %6.1f" % (short(r["Kernel_Name"]), (st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3))
        busy += en - st; prev_end = en; n += 1
    print("   launches %d, kernel time %.1f us, gaps %.1f us" % (n, busy / 1e3, (prev_end - t0 - busy) / 1e3))

if "--setup" in sys.argv:
    # what the device runs between the last kernel of the previous call and the first Arnoldi step of the last call, run-length grouped
    i_first = fin[base]
    i_prev = fin[base - 1]
    seg = rows[i_prev + 1:i_first + 1]
    # the call starts after the longest idle gap in that segment (host work between calls)
    gaps = [(int(seg[j + 1]["Start_Timestamp"]) - int(seg[j]["End_Timestamp"]), j) for j in range(len(seg) - 1)]
    j0 = max(gaps)[1] + 1 if gaps else 0
    seg = seg[j0:]
    t0 = int(seg[0]["Start_Timestamp"])
    print("== set-up of the last call: %.2f ms from its first kernel to the end of the first k_orth_finish, %d launches" %
          ((int(seg[-1]["End_Timestamp"]) - t0) / 1e6, len(seg)))
    cur = None; n = 0; tsum = 0; tstart = 0; prev_end = t0; gapsum = 0
    def flush():
        if cur is not None:
            print("  %-46s x%-4d start %8.1f us  kernel %8.1f us  gaps %7.1f us" % (cur, n, tstart, tsum / 1e3, gapsum / 1e3))
    for r in seg:
        nm = short(r["Kernel_Name"]); st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if nm != cur:
            flush(); cur = nm; n = 0; tsum = 0; gapsum = 0; tstart = (st - t0) / 1e3
        n += 1; tsum += en - st; gapsum += max(0, st - prev_end); prev_end = max(prev_end, en)
    flush()
