"""N headline iar runs in one process: wall time of each, refinement misses, the trace line of the outliers"""
import sys, os, time, io, contextlib, gc; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "notrace" not in sys.argv: os.environ["NEP_IAR_TRACE"] = "1"
import numpy as np, torch, nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    gc.disable()
iarf = sys.modules[na.iar.__module__].iar
ts = []; traces = []
for rep in range(N):
    buf = io.StringIO()
    torch.cuda.synchronize(); t = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        lam, Q = na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)[:2]
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t2 - t) * 1e3); traces.append("[returned after %.1f ms, sync %.1f ms, gc %s] " % ((t1 - t) * 1e3, (t2 - t1) * 1e3, gc.get_count()) + buf.getvalue().strip())
    assert len(lam) == 46
ts = np.array(ts)
med = np.median(ts[3:])
print("runs", N, "misses", iarf.refinement_misses, "median %.1f ms" % med, "mean %.1f" % ts[3:].mean(), "max %.1f" % ts[3:].max())
print("typical:", (traces[-1] if ts[-1] < 1.1 * med else traces[-2])[:400])
for i, x in enumerate(ts):
    if i >= 3 and x > 1.2 * med:
        print("outlier run %d %.1f ms:" % (i, x), traces[i])
