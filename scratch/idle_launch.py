import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na, threading
x=torch.zeros(10000,dtype=torch.complex128,device="cuda")
def trial(gap, busy_threads=0):
    stop=[False]
    def burn():
        H=np.triu(np.random.randn(100,100)+1j*np.random.randn(100,100),-1)
        he=sys.modules.get("nep_amd._hosteig")
        while not stop[0]: he.eig(H)
    ths=[threading.Thread(target=burn) for _ in range(busy_threads)]
    [t.start() for t in ths]
    lc=[]; tot=[]
    for _ in range(200):
        torch.cuda.synchronize()
        t=time.perf_counter()
        while time.perf_counter()-t<gap: pass
        t0=time.perf_counter(); na.dense.scal(x,1.0); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        lc.append(t1-t0); tot.append(t2-t0)
    stop[0]=True; [t.join() for t in ths]
    return np.median(lc)*1e6, np.mean(lc)*1e6, np.median(tot)*1e6
for bt in (0,4):
    for gap in (0,1e-4,1e-3,5e-3):
        print("busy eig threads %d gap %.1f ms: launch call median %.1f us mean %.1f us, launch+sync median %.1f us"%((bt,gap*1e3)+trial(gap,bt)))
