"""Oracle-side diagnostics of the dynamic NLEIGS run on the "particle in a canyon" problem
(test/nleigs/nleigs_particle_variant_r2.jl): Ritz values near the interval, their residuals, the norm of the Ritz
coefficient vector s (||H s|| = 1) and cond(H) at several subspace sizes, plus the relative size of the new direction in
every early step (beta / |w|), for the reference's start vector and for seeded random ones.
Usage: python scripts/diag/nleigs_r2_history.py [ref|normal|uniform|complex|demean] [seed]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as sla
from oracle import gallery, nleigs as onl, solvers

nep, Sigma, Xi, v, nodes, xmin, xmax = gallery.particle_init(2)
n = nep.size(1)
E = solvers.ResidualErrmeasure(nep)
which = sys.argv[1] if len(sys.argv) > 1 else "ref"
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
v0 = {"ref": lambda: v, "demean": lambda: v - v.mean(), "normal": lambda: rng.standard_normal(n) + 0j,
      "uniform": lambda: 1 - 2 * rng.random(n) + 0j, "complex": lambda: rng.standard_normal(n) + 1j * rng.standard_normal(n)}[which]()
info = {"_debug": True, "_nobreak": True}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    onl.nleigs(nep, Sigma, Xi=Xi, maxdgr=50, minit=30, maxit=100, v=v0, nodes=nodes, info=info)
V, H, K, sig = info["V"], info["H"], info["K"], info["sigma"]
print("N", info["N"], "interval", xmin, xmax)
for l in list(range(1, 13)) + [20, 30, 43]:
    h = H[:l + 1, l - 1]
    print("step %3d shift %.6f |w| %.2e beta/|w| %.1e" % (l, sig[l].real, np.linalg.norm(h), abs(h[l]) / np.linalg.norm(h)))
for l in (50, 60, 73, 78, 83, 100):
    lam_, S = sla.eig(K[:l, :l], H[:l, :l])
    out = []
    for i in [i for i in range(l) if xmin < lam_[i].real < xmax and abs(lam_[i].imag) < 1e-6]:
        s = S[:, i] / np.linalg.norm(H[:l + 1, :l] @ S[:, i])
        x = V[:n, :l + 1] @ (H[:l + 1, :l] @ s); x /= np.linalg.norm(x)
        out.append("%.11f%+.1ei res %.1e |s| %.0e" % (lam_[i].real, lam_[i].imag, E(lam_[i], x), np.linalg.norm(s)))
    print("l %3d cond(H) %.1e  " % (l, np.linalg.cond(H[:l + 1, :l])) + " | ".join(out))
