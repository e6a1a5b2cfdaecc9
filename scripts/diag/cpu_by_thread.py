"""CPU time per host thread (by thread name) over N headline iar calls: who burns the cgroup's CPU quota"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
TCK = os.sysconf("SC_CLK_TCK")
def snap():
    d = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            d[tid] = (name, (int(rest[11]) + int(rest[12])) / TCK)
        except Exception:
            pass
    return d
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
for _ in range(4): na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# threads of a call die with it: sample inside the calls as well (every 5 ms from a sampler thread)
import threading
acc = collections.Counter(); last = {}; stop = [False]; first_seen = {}; last_seen = {}
def sampler():
    while not stop[0]:
        s = snap(); now = time.perf_counter()
        for tid, (name, t) in s.items():
            if tid in last:
                acc[tid] += t - last[tid][1]
            else:
                first_seen[tid] = now
            last[tid] = (name, t); last_seen[tid] = now
        time.sleep(0.004)
th = threading.Thread(target=sampler, name="sampler", daemon=True); th.start()
time.sleep(0.05); acc.clear()
t0 = time.perf_counter()
for _ in range(N): na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
wall = time.perf_counter() - t0
stop[0] = True; th.join()
t_end = time.perf_counter()
print("wall %.1f ms per call; CPU ms per call:" % (wall / N * 1e3))
pers = {tid: t for tid, t in acc.items() if last_seen[tid] - first_seen.get(tid, t0) > 0.8 * wall}
trans = {tid: t for tid, t in acc.items() if tid not in pers}
print("  long-lived threads (%d): %.1f in total; the busiest:" % (len(pers), sum(pers.values()) / N * 1e3))
for tid, t in sorted(pers.items(), key=lambda kv: -kv[1])[:10]:
    print("    tid %s (%s): %.1f" % (tid, last[tid][0], t / N * 1e3))
print("  per-call threads (%d seen): %.1f in total (sampled every 4 ms: undercounts short ones)" % (len(trans), sum(trans.values()) / N * 1e3))
print("  main thread: %.1f" % (acc.get(str(os.getpid()), 0.0) / N * 1e3))
