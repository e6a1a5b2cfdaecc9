"""phase cycles of k_dft_cols_rb from a -DWEP_PROF build of util.hip + wep.hip (+ gemm.o) linked into scripts/diag/_wep_prof.so:
load / stage 1 / stage 2 + store, summed over the workgroups, and the longest workgroup.  python scripts/diag/wep_prof_run.py"""
import ctypes as C, numpy as np, torch, os
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_wep_prof.so"))
nz, nx = 999, 1003
D = (np.random.default_rng(0).standard_normal(nz) + 1j * np.random.default_rng(1).standard_normal(nz)) - 50.0
h = C.c_void_p()
lib.nep_wep_sylv_create.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]
rc = lib.nep_wep_sylv_create(nz, nx, D.ctypes.data, 1.0e4, C.byref(h)); assert rc == 0, rc
X = torch.randn(nx * nz, dtype=torch.float64, device="cuda").to(torch.complex128)
lib.nep_wep_sylv_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
out = (C.c_ulonglong * 16)()
for _ in range(3):
    lib.nep_wep_sylv_solve(h, X.data_ptr(), None)
torch.cuda.synchronize(); lib.nep_wep_prof_read(out, 1)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.nep_wep_sylv_solve(h, X.data_ptr(), None)
e1.record(); torch.cuda.synchronize(); lib.nep_wep_prof_read(out, 0)
o = list(out)
print("solve %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
for name, b in (("forward", 0), ("inverse", 8)):
    n = max(o[b + 3], 1)
    print(name, "per workgroup cycles: load %.0f stage1 %.0f stage2+store %.0f | longest workgroup %d cycles (%.1f us at 2.4 GHz), %d workgroups x 20" %
          (o[b] / n, o[b + 1] / n, o[b + 2] / n, o[b + 4], o[b + 4] / 2400.0, n // 20), "| stage-1 arithmetic of wave 0 %.0f, barrier wait %.0f" % (o[b + 5] / n, o[b + 6] / n))
