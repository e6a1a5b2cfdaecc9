import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
def main():
    import nep_amd as na
    print("start", flush=True)
    nep=na.nep_gallery("gun_spmf"); n=nep.n; nep.dev
    Vh=na.probe_block(n,32)
    t=time.perf_counter(); na.HostLUPool.warm(); print("pool warm %.2f s"%(time.perf_counter()-t), flush=True)
    for N in (8, 64):
        t=time.perf_counter()
        lam,V=na.contour_beyn(nep,Vh=Vh,sigma=250.0**2,radius=1e4,N=N,k=32,neigs=10**6,tol=1e-6)
        torch.cuda.synchronize()
        print("N=%d: %d eigs in %.2f s"%(N,len(lam),time.perf_counter()-t), flush=True)
    print("done", flush=True)
if __name__=="__main__":
    main()
