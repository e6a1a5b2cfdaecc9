"""cycle split of k_hess_qr (csrc/hesseig.hip) from an instrumented build: compile util.hip + hesseig.hip with -DHQ_PROF into
scripts/diag/_hq_prof.so (hipcc -shared), then `python scripts/diag/hq_prof_run.py`: cycles in wait / deflation scan / shift /
publish / QR half per sweep and per rotation."""
import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0,'/root/repo')
lib=C.CDLL('/root/repo/scripts/diag/_hq_prof%s.so' % (sys.argv[1] if len(sys.argv) > 1 else ''))
H=np.load('/root/repo/tests/golden/gun_iar_H100.npy')
for k in (10,50,100):
    Hk=np.ascontiguousarray(H[:k,:k].T)
    Hd=torch.from_numpy(Hk).cuda()
    nb=C.c_int64(0); lib.nep_hess_eig_worksize(k,C.byref(nb))
    work=torch.zeros(nb.value,dtype=torch.uint8,device='cuda'); w=torch.zeros(k+2,dtype=torch.complex128,device='cuda')
    lib.nep_hess_eigvals_dev.argtypes=[C.c_int32,C.c_void_p,C.c_int64,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p]
    for _ in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); rc=lib.nep_hess_eigvals_dev(k,Hd.data_ptr(),k,w.data_ptr(),work.data_ptr(),None,None); e1.record(); torch.cuda.synchronize()
    off=16*(k*k+k+2)
    dbg=work[off:off+64].cpu().numpy().view(np.float64)
    ms=e0.elapsed_time(e1)
    print('k',k,'rc',rc,'ms %.3f'%ms,'cycles total %.0f (%.2f GHz)'%(dbg[0],dbg[0]/ms/1e6),'wait %.0f scan %.0f shift %.0f pub %.0f qr %.0f'%tuple(dbg[1:6]),'steps %.0f sweeps %.0f'%(dbg[6],dbg[7]),'qr/step %.0f, per-sweep non-qr %.0f'%(dbg[5]/max(dbg[6],1),(dbg[1]+dbg[2]+dbg[3]+dbg[4])/max(dbg[7],1)))
