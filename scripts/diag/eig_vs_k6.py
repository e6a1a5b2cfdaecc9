"""does a one-workgroup-per-matrix eig batch start promptly while the recurrence's streaming kernels (K6 at the shape of iar
step k) flood the device from another stream?  Times a QR batch on its own stream (normal / low priority) with and without
the K6 loop running next to it.   python scripts/diag/eig_vs_k6.py"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nep_amd as na
from nep_amd import dense
from nep_amd._lib import lib, check, c_vp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = np.load(os.path.join(ROOT, "tests", "golden", "gun_iar_H100.npy"))
n = 9956
def k6_setup(k):
    rows = n * (k + 1)
    active = (np.arange(1, k + 1) * n).astype(np.int64)
    V = torch.randn((k, rows), dtype=torch.float64, device="cuda").to(torch.complex128)
    w = torch.randn(rows, dtype=torch.float64, device="cuda").to(torch.complex128)
    act_d = torch.from_numpy(active).to("cuda"); out = torch.zeros(k + 2, dtype=torch.complex128, device="cuda")
    return lambda: dense.orthogonalize_and_normalize_dev(V, w, k, out, rows=rows, ldv=rows, active_dev=act_d, method=dense.CGS)
k6 = k6_setup(90)
for kq, nb in ((40, 1), (60, 4), (76, 8), (100, 8)):
    Hd = torch.from_numpy(np.ascontiguousarray(H[:100, :100].T)).to("cuda")
    wsz = (dense.hess_eig_worksize(kq) + 15) // 16 * 16
    work = torch.empty(nb * wsz, dtype=torch.uint8, device="cuda"); w = torch.zeros((nb, kq + 2), dtype=torch.complex128, device="cuda")
    for prio in (0, 1):
        st = torch.cuda.Stream(priority=prio)
        for flood in (False, True):
            torch.cuda.synchronize()
            if flood:
                for _ in range(60):
                    k6()                          # ~60 x 0.3 ms of back-to-back streaming kernels on the current stream
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record()
                check(lib.nep_hess_eigvals_batch_dev(nb, kq - nb + 1, 1, c_vp(Hd.data_ptr()), 100, c_vp(w.data_ptr()), kq + 2, c_vp(work.data_ptr()), wsz, None, 0, c_vp(st.cuda_stream)))
                e1.record()
            torch.cuda.synchronize()
            print(json.dumps({"kmax": kq, "batch": nb, "lds_KB": round((16 * kq * kq + 32 * (kq + 1) + 48) / 1024, 1), "prio": prio, "k6_flood": flood, "qr_ms": round(e0.elapsed_time(e1), 3)}))
