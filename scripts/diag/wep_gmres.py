"""One fixed-shift solve of the waveguide problem through the Schur complement with GMRES + Sylvester-SMW preconditioner:
iterations, true residual of the full system, time.  Usage: python scripts/diag/wep_gmres.py [nx nz N [reltol]]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nep_amd as na

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 303
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 299
N = int(sys.argv[3]) if len(sys.argv) > 3 else 23
reltol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-10
sigma = -3 - 3.5j
nep = na.nep_gallery("WEP", nx=nx, nz=nz, benchmark_problem="JARLEBRING"); nep.dev
n = nep.n
t = time.perf_counter(); P = na.wep_generate_preconditioner(nep, N, sigma); torch.cuda.synchronize(); tp = time.perf_counter() - t
s = na.create_linsolver(na.WEPLinSolverCreator(solver_type="gmres", kwargs=(("Pl", P), ("reltol", reltol), ("restart", 100), ("maxiter", 400))), nep, sigma)
rng = np.random.default_rng(0)
b = na.to_dev(rng.standard_normal(n) + 1j * rng.standard_normal(n))[0]
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    x = s.solve_dev(b)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
r = nep.compute_Mlincomb(sigma, x.reshape(1, n)).reshape(-1) - b
print("graph step:", s.gmres.fused_step is not None, getattr(s, "_graph_error", None))
print(json.dumps(dict(nx=nx, nz=nz, n=n, N=N, mm=P.mm, smw_cond=P.cond, precond_setup_s=tp, reltol=reltol, gmres_iterations=s.iterations[-1],
                      solve_s=dt, true_rel_residual=float(torch.linalg.norm(r) / torch.linalg.norm(b)))))
# cost of the pieces
ops = s.ops
v = torch.randn(nep.N, dtype=torch.complex128, device="cuda"); out = torch.empty_like(v)
cand = [("SchurMatVec", lambda: ops.matvec(v, out)), ("preconditioner", lambda: P(out)), ("Sylvester solve", lambda: P.linv(out))]
if s.gmres.fused_step is not None:
    cand.append(("graph step", lambda: s.gmres.fused_step(v, out)))
for name, f in cand:
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print("%-16s %.3f ms" % (name, 1e2 * (time.perf_counter() - t)))
