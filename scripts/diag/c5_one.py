"""two C5 runs (for kernel statistics): python scripts/diag/c5_one.py"""
import os, sys, time
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nep_amd as na, torch
import baseline_configs as bc
for _ in range(2):
    lam, Q, res, info = bc.c5_device(na)
    print("solve_s %.3f setup %.3f pairs %d maxres %.2e" % (info["solve_s"], info.get("preconditioner_setup_s", 0), len(lam), max(res)))
