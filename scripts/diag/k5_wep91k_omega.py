import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import nep_amd as na
nep = na.nep_gallery("WEP", nx=303, nz=299, benchmark_problem="JARLEBRING")
lam = -3 - 3.5j
A = sp.csc_matrix(nep.compute_Mder(lam)); n = nep.n
rng = np.random.default_rng(3)
b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
ls = na.create_linsolver(na.FactorizeLinSolverCreator(), nep, lam)
lu = ls.lu
x0 = na.to_host(lu.solve(na.to_dev(b)))[:, 0]
print("raw rel resid %.3e" % (np.linalg.norm(A @ x0 - b) / np.linalg.norm(b)))
for i in range(8):
    x = na.lin_solve(ls, b)
    print(i, "omega %.3e steps %d resid %.3e" % (ls.last_omega, ls.refine_steps_taken, np.linalg.norm(A @ x - b) / np.linalg.norm(b)))
# the same with the numeric factorisation done on the device (stored pivot sequence) -- what a second factorisation of the pattern gets
from nep_amd.linsolvers import _DeviceRefactor
_DeviceRefactor.wait()
ls2 = na.create_linsolver(na.FactorizeLinSolverCreator(), nep, lam)
print("device factorized:", getattr(ls2.lu, "device_factorized", None))
for i in range(9):
    x = na.lin_solve(ls2, b)
    print(i, "omega %.3e steps %d resid %.3e" % (ls2.last_omega, ls2.refine_steps_taken, np.linalg.norm(A @ x - b) / np.linalg.norm(b)))
