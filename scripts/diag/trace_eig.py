"""timeline of the last iar call in a rocprofv3 kernel trace: Arnoldi steps (k_orth_finish), eig batches (k_hess_qr / k_hess_invit)
    python scripts/diag/trace_eig.py gpurun_out/<dir>/kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = "Kernel_Name"; st = "Start_Timestamp"; en = "End_Timestamp"
rows.sort(key=lambda r: int(r[st]))
fin = [r for r in rows if "k_orth_finish" in r[name]]
# last call = last 100 finishes
fin = fin[-100:]
t0 = int(fin[0][st])
print("steps: first finish at 0, step 25 %.2f, 50 %.2f, 75 %.2f, 100 %.2f ms" % tuple((int(fin[i][en]) - t0) / 1e6 for i in (24, 49, 74, 99)))
qr = [r for r in rows if "k_hess_qr" in r[name] and int(r[st]) >= t0]
iv = [r for r in rows if "k_hess_invit" in r[name] and int(r[st]) >= t0]
for r in qr:
    print("qr  grid %5s start %7.2f end %7.2f (%.2f ms) queue %s" % (r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r[st]) - t0) / 1e6, (int(r[en]) - t0) / 1e6, (int(r[en]) - int(r[st])) / 1e6, r.get("Queue_Id", "?")))
last = max(int(r[en]) for r in rows)
print("last kernel of the trace ends at %.2f ms" % ((last - t0) / 1e6))
qs = {}
for r in rows:
    if int(r[st]) >= t0:
        qs.setdefault(r.get("Queue_Id", "?"), set()).add(r[name].split("(")[0][:40])
for q, v in qs.items():
    print("queue", q, sorted(v)[:12])
