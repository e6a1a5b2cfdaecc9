import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, torch, nep_amd as na
from nep_amd import nep_amd_hostlu as _nep_hostlu
nep = na.nep_gallery("gun_spmf_scaled"); nep.dev
A = nep.compute_Mder(0.0)
T=time.perf_counter
for i in range(6):
    t0=T(); Ac=sp.csc_matrix(A,dtype=np.complex128); t1=T()
    ctl=_nep_hostlu.blas_controller()
    with ctl.limit(limits=1):
        lu=spla.splu(Ac,permc_spec="MMD_AT_PLUS_A",diag_pivot_thresh=0.001,options=dict(SymmetricMode=True))
    t2=T(); L=sp.csr_matrix(lu.L); U=sp.csr_matrix(lu.U); t3=T(); L.sort_indices(); U.sort_indices(); t4=T()
    arrs=[np.ascontiguousarray(x) for x in (L.indptr,L.indices,L.data,U.indptr,U.indices,U.data)]; nrm=float(np.linalg.norm(Ac.data)); t5=T()
    print("csc %.1f splu %.1f tocsr(L,U incl lu.L/lu.U) %.1f sort %.1f arrays %.1f | total %.1f"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(t5-t0)*1e3),flush=True)
