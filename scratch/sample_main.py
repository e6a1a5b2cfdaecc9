import sys, os, time, threading, collections, traceback; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
nep=na.nep_gallery("gun_spmf_scaled"); nep.dev
def step(): return na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10, return_device=True)
step(); step()
main_id=threading.get_ident(); hist=collections.Counter(); stop=[False]
def sampler():
    while not stop[0]:
        fr=sys._current_frames().get(main_id)
        if fr is not None:
            stack=traceback.extract_stack(fr)
            # innermost frame inside our package
            key=None
            for f in reversed(stack):
                if "nonlineareigenproblems" in f.filename or "concurrent" in f.filename or "threading" in f.filename:
                    key="%s:%d %s"%(os.path.basename(f.filename),f.lineno,f.name); break
            hist[key or ("%s:%d"%(os.path.basename(stack[-1].filename),stack[-1].lineno))]+=1
        time.sleep(0.0005)
th=threading.Thread(target=sampler); th.start()
t=time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt=time.perf_counter()-t
stop[0]=True; th.join()
tot=sum(hist.values())
print("5 steps %.1f ms, %d samples"%(dt*1e3,tot))
for k,v in hist.most_common(14): print("%5.1f%%  %s"%(100*v/tot,k))
