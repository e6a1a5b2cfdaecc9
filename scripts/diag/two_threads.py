"""two host threads, each running the headline iar configuration on its OWN NEP object at the same time (shared device, shared
pool, shared check stream; thread-local scratch): results must equal the single-threaded ones"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
neps = [na.nep_gallery("gun_spmf_scaled") for _ in range(2)]
for n_ in neps: n_.dev
ref = np.sort_complex(na.iar(neps[0], maxit=100, neigs=np.inf, v=np.ones(neps[0].n), tol=1e-10)[0])
bad = [0]; runs = [0]
def work(nep, reps):
    torch.cuda.set_device(0)
    for _ in range(reps):
        try:
            lam = np.sort_complex(na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)[0])
            ok = len(lam) == len(ref) and np.abs(lam - ref).max() <= 1e-8 * np.abs(ref).max()
        except Exception as e:
            print("EXC", repr(e)[:200], flush=True); ok = False
        runs[0] += 1
        if not ok: bad[0] += 1
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t = time.perf_counter()
th = [threading.Thread(target=work, args=(neps[i], R)) for i in range(2)]
for x in th: x.start()
for x in th: x.join()
print("two threads: %d runs, %d bad, %.1f ms per run per thread" % (runs[0], bad[0], (time.perf_counter() - t) * 1e3 / R))
