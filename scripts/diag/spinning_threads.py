"""which long-lived host threads spin: thread ids by creation stage, then their CPU use over one idle second and over 20 iar calls"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
TCK = os.sysconf("SC_CLK_TCK")
def tids(): return set(os.listdir("/proc/self/task"))
def cpu(tid):
    try:
        f = open("/proc/self/task/%s/stat" % tid).read(); r = f[f.rindex(")") + 2:].split(); return (int(r[11]) + int(r[12])) / TCK
    except Exception:
        return 0.0
stage = {}; seen = tids()
def mark(name):
    global seen
    now = tids()
    for t in now - seen: stage[t] = name
    seen = now
import numpy as np; mark("numpy")
import scipy.linalg; mark("scipy")
import torch; mark("torch")
torch.zeros(1, device="cuda"); torch.cuda.synchronize(); mark("cuda init")
import nep_amd as na; mark("nep_amd")
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev; mark("nep.dev")
na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10); mark("first iar")
for _ in range(3): na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
mark("more iar")
live = sorted(tids(), key=int)
a = {t: cpu(t) for t in live}; time.sleep(1.0); b = {t: cpu(t) for t in live}
t0 = time.perf_counter()
for _ in range(20): na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
wall = time.perf_counter() - t0
c = {t: cpu(t) for t in live}
print("threads alive: %d; by creation stage:" % len(live), {s: sum(1 for t in live if stage.get(t) == s) for s in dict.fromkeys(stage.values())})
print("threads using > 5%% of a CPU while the process sleeps, or > 30%% during the calls (wall %.0f ms):" % (wall * 1e3))
for t in live:
    idle = b[t] - a[t]; busy = (c[t] - b[t]) / wall
    if idle > 0.05 or busy > 0.3:
        print("  tid %s created at %-10s idle-second CPU %.2f s, during calls %.0f %%" % (t, stage.get(t, "start"), idle, 100 * busy))
