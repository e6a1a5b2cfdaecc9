"""host-side profile (cProfile, cumulative) of the second of two C5 runs: python scripts/diag/c5_cprofile.py"""
import os, sys, cProfile, pstats, io
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nep_amd as na, torch
import baseline_configs as bc
bc.c5_device(na)
pr = cProfile.Profile()
pr.enable()
lam, Q, res, info = bc.c5_device(na)
pr.disable()
print("solve_s %.3f" % info["solve_s"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
