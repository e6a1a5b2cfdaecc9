import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"): os.environ.setdefault(_v, "8")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, torch, nep_amd as na
from nep_amd import nep_amd_hostlu as _nep_hostlu
nx, nz = int(sys.argv[1]), int(sys.argv[2])
T = time.perf_counter
t0 = T(); nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); nep.dev; t1 = T()
A = nep.compute_Mder(-3 - 3.5j); t2 = T()
Ac = sp.csc_matrix(A, dtype=np.complex128); t3 = T()
F = _nep_hostlu.factor(Ac.data, Ac.indices, Ac.indptr, Ac.shape); t4 = T()
lu = na.DeviceLU(factors=F, expected_solves=200); torch.cuda.synchronize(); t5 = T()
print("generate+upload %.2f | compute_Mder %.2f | csc %.2f | factor total %.2f (splu %.2f) | DeviceLU(factors) %.2f (lib create %.2f)" %
      (t1 - t0, t2 - t1, t3 - t2, t4 - t3, F["t_factor"], t5 - t4, lu.t_create))
