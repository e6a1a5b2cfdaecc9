"""device factorisation of a batch of B gun matrices in one pass (grid.y = matrix): time per pass for the panel sizes of the wide levels
python scripts/diag/lu_batch_time.py [P ...]      (each P builds its own plan; NEP_LU_WIDE_P is read at plan time)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
import nep_amd as na
from nep_amd._lib import lib, check, hptr, c_vp
from nep_amd.nep import stream_ptr
import nep_amd_hostlu as hl
nep = na.nep_gallery("gun_spmf_scaled")
zs = [0.05 * np.exp(2j * np.pi * (i + 0.5) / 16) for i in range(16)]
mats = [sp.csc_matrix(nep.compute_Mder(z)).astype(np.complex128) for z in zs]
for A in mats: A.sort_indices()
A0 = mats[0]; n = A0.shape[0]
F = hl.factor(A0.data, A0.indices, A0.indptr, A0.shape)
ref = na.DeviceLU(factors=F)
for P in [int(a) for a in sys.argv[1:]] or [1, 4]:
    os.environ["NEP_LU_WIDE_P"] = str(P)
    h = c_vp()
    check(lib.nep_lu_refac_create(ref.h, n, hptr(F["Lp"]), hptr(F["Li"]), hptr(F["Up"]), hptr(F["Ui"]), hptr(F["perm_r"]), hptr(F["perm_c"]),
                                  hptr(np.ascontiguousarray(A0.indptr, dtype=np.int32)), hptr(np.ascontiguousarray(A0.indices, dtype=np.int32)), C.byref(h)))
    for B in (1, 2, 4, 8, 16):
        Ax = np.ascontiguousarray(np.stack([m.data for m in mats[:B]]))
        health = np.zeros((B, 3)); ts = []
        for rep in range(6):
            outs = (c_vp * B)()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib.nep_lu_factor_dev_batch(h, B, hptr(Ax), 10, 1e8, hptr(health), None, outs, stream_ptr()))
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
            for o in outs:
                if o: lib.nep_lu_destroy(o)
        print("P %d  B %2d  factor kernels + read-back %.2f ms, schedules built %.2f ms (min of 5)" % (P, B, min(t[0] for t in ts[1:]), min(t[1] for t in ts[1:])), flush=True)
    lib.nep_lu_refac_destroy(h)
