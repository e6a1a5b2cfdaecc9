import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, nep_amd as na
from oracle import gallery as og, solvers as osol
nep = na.nep_gallery("dep_symm_double", 10); n = nep.n
t = time.time(); out = na.ilan(nep, v=np.ones(n), tol=1e-5, neigs=12); print("device ilan %.2f s" % (time.time() - t))
lam, W = out[0], out[1]
print(np.sort(lam.real))
onep = og.dep_symm_double(10)
print("resid", max(np.linalg.norm(onep.compute_Mlincomb(lam[i], W[:, i])) / np.linalg.norm(W[:, i]) for i in range(len(lam))))
oo = osol.ilan(onep, v=np.ones(n), tol=1e-5, neigs=12)
print("H diff", np.linalg.norm(out[3] - oo[3]) / np.linalg.norm(oo[3]), "omega diff", np.linalg.norm(out[4] - oo[4]) / np.linalg.norm(oo[4]))
out2 = na.ilan(nep, v=np.ones(n), tol=1e-5, neigs=3, proj_solve=False, maxit=30)
print("ritz path", out2[0])
Hd, Ho = out[3], oo[3]
for j in range(0, Hd.shape[1], 4):
    print(j, np.linalg.norm(Hd[:, j] - Ho[:, j]) / np.linalg.norm(Ho[:, j]), abs(out[4][j] - oo[4][j]) / abs(oo[4][j]))
