"""wide-level panels (k_lu_widep) against the host factor: per P, per matrix of a batch of two"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
import nep_amd as na
from oracle import gallery as og
from nep_amd._lib import lib, check, hptr, c_vp
from nep_amd.nep import stream_ptr
import nep_amd_hostlu as hl
nn = int(sys.argv[1]) if len(sys.argv) > 1 else 1310
onep = og.gun_spmf_scaled(nn)
mats = [sp.csc_matrix(onep.compute_Mder(z)).astype(np.complex128) for z in (0.1, 0.15 + 0.05j)]
for A in mats: A.sort_indices()
A0, A1 = mats; n = A0.shape[0]
F = hl.factor(A0.data, A0.indices, A0.indptr, A0.shape); F1 = hl.factor(A1.data, A1.indices, A1.indptr, A1.shape)
ref = na.DeviceLU(factors=F); nL = len(F["Lx"]); nU = len(F["Ux"])
order = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
for P in order:
    os.environ["NEP_LU_WIDE_P"] = str(P)
    h = c_vp()
    check(lib.nep_lu_refac_create(ref.h, n, hptr(F["Lp"]), hptr(F["Li"]), hptr(F["Up"]), hptr(F["Ui"]), hptr(F["perm_r"]), hptr(F["perm_c"]),
                                  hptr(np.ascontiguousarray(A0.indptr, dtype=np.int32)), hptr(np.ascontiguousarray(A0.indices, dtype=np.int32)), C.byref(h)))
    wi = (C.c_int64 * 5)(); check(lib.nep_lu_refac_wide_info(h, wi)); print("P", P, "wide info", list(wi))
    Ax = np.ascontiguousarray(np.stack([A0.data, A1.data]))
    LU = np.empty((2, nL + nU), dtype=np.complex128); health = np.zeros((2, 3)); outs = (c_vp * 2)()
    check(lib.nep_lu_factor_dev_batch(h, 2, hptr(Ax), 10, 1e8, hptr(health), hptr(LU), outs, stream_ptr()))
    for b_, (A, Fh) in enumerate(((A0, F), (A1, F1))):
        same = np.array_equal(Fh["perm_r"], F["perm_r"]) and np.array_equal(Fh["Lp"], F["Lp"])
        mk = lambda x, i_, p_: sp.csc_matrix((np.array(x), np.array(i_), np.array(p_)), shape=(n, n))    # copies: scipy sorts in place
        Ld = mk(LU[b_, :nL], F["Li"], F["Lp"]); Ud = mk(LU[b_, nL:], F["Ui"], F["Up"])
        Lh = mk(Fh["Lx"], Fh["Li"], Fh["Lp"]); Uh = mk(Fh["Ux"], Fh["Ui"], Fh["Up"])
        pr = F["perm_r"]; pc = F["perm_c"]
        print("  b", b_, "same pivots/pattern", same, "health", health[b_], "dL %.2e dU %.2e" % (abs(Ld - Lh).max() / abs(Lh).max(), abs(Ud - Uh).max() / abs(Uh).max()))
        if outs[b_]: lib.nep_lu_destroy(outs[b_])
    lib.nep_lu_refac_destroy(h)
