"""CPU oracle for the nep-mi355 hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the reference algorithms
(nep-pack/NonlinearEigenproblems.jl v1.1.1) for the hot path named in
BASELINE.json: compute_Mlincomb / compute_MM / compute_Mder of SPMF-type
NEPs, the fixed-shift lin_solve, DGKS orthogonalisation and the drivers
iar, tiar, resinv, quasinewton, contour_beyn, nleigs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it, and only as the checker.  The product package
(nonlineareigenproblems.jl_amd, import name ``nep_amd``) never imports it.

Parity pin: every function cites the reference file:line it follows; the
restatement is pinned against the known-answer values the reference itself
publishes in docstrings/tests in tests/test_oracle_kat.py.
The reference (Julia) can be neither compiled nor imported in this image.
"""
