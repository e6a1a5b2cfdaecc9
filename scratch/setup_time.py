import sys, time; sys.path.insert(0,'.')
import numpy as np, nep_amd as na, torch
nep=na.nep_gallery("gun_spmf_scaled"); nep.dev; n=nep.n; m=100
def T(label, f):
    torch.cuda.synchronize(); t=time.perf_counter(); r=f(); torch.cuda.synchronize(); print("%-28s %.2f ms"%(label,(time.perf_counter()-t)*1e3)); return r
for rep in range(3):
    print("--- rep",rep)
    V=T("V zeros 1.6GB", lambda: torch.zeros((m+1, n*(m+1)), dtype=torch.complex128, device="cuda"))
    A=T("compute_Mder", lambda: nep.compute_Mder(0.0))
    lu=T("DeviceLU(hint 200)", lambda: na.DeviceLU(A, expected_solves=200))
    print("     t_factor %.1f t_convert %.1f t_create %.1f t_setup %.1f"%(lu.t_factor*1e3, lu.t_convert*1e3, lu.t_create*1e3, lu.t_setup*1e3))
    import _nep_hostlu, scipy.sparse as sp
    Ac=sp.csc_matrix(A,dtype=np.complex128)
    t=time.perf_counter(); F=_nep_hostlu.factor(Ac.data,Ac.indices,Ac.indptr,Ac.shape); print("     factor() alone total %.1f ms (t_factor %.1f t_total %.1f)"%((time.perf_counter()-t)*1e3,F["t_factor"]*1e3,F["t_total"]*1e3))
    tab=T("derivative_table", lambda: nep.derivative_table(0.0, m, rowscale=np.ones(m)))
    T("first solve (graph capture)", lambda: lu.solve(V[0,:n]))
    T("second solve", lambda: lu.solve(V[0,:n]))
    T("del V", lambda: None)
    del V, lu
    T("after del", lambda: None)
