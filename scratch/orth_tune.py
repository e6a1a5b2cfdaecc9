import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na, bench
print(os.environ.get("NEP_DOTS_TARGET"), bench.orth_roofline(na, 9956, 100, reps=20)["achieved"])
