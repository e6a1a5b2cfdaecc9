"""prints how many DGKS passes each iar step of the gun run takes (NEP_IAR_PASSES=1)"""
import os, sys
os.environ["NEP_IAR_PASSES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import nep_amd as na
nep = na.nep_gallery("gun_spmf_scaled")
lam, Q, V = na.iar(nep, maxit=100, neigs=np.inf, v=np.ones(nep.n), tol=1e-10)
print(len(lam))
