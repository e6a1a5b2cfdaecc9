import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def stat():
    d = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split(); d[k] = int(v)
    except Exception as e:
        d["err"] = str(e)
    return d
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, "affinity", len(os.sched_getaffinity(0)))
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
for _ in range(4): na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
s0 = stat(); ts = []
for rep in range(60):
    a = stat(); torch.cuda.synchronize(); t = time.perf_counter()
    na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3; b = stat()
    ts.append((dt, b.get("nr_throttled", 0) - a.get("nr_throttled", 0), (b.get("throttled_usec", 0) - a.get("throttled_usec", 0)) / 1e3, (b.get("usage_usec", 0) - a.get("usage_usec", 0)) / 1e3))
s1 = stat()
print("total: periods", s1.get("nr_periods", 0) - s0.get("nr_periods", 0), "throttled", s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), "throttled ms %.1f" % ((s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3))
med = np.median([x[0] for x in ts])
print("median %.1f ms; cpu ms per call median %.0f" % (med, np.median([x[3] for x in ts])))
for i, x in enumerate(ts):
    if x[0] > 1.2 * med or x[1] > 0:
        print("run %d: %.1f ms, throttled periods %d, throttled %.1f ms, cpu used %.0f ms" % (i, *x))
