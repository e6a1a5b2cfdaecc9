"""nep_amd -- MI355X-native inner linear-algebra backend for NEP-PACK-style nonlinear eigensolvers.

Product package (directory `nonlineareigenproblems.jl_amd/`, import name `nep_amd`).  The compute
path is libnepmi355.so (hand-written HIP for gfx950 behind the C ABI of include/nepmi355.h); this
package is the host-side mirror of the reference's NEP / LinSolver / orthogonalisation interface.
It never imports the CPU oracle and has no CPU fallback.
"""
from . import _lib, funcs, dense
from ._lib import NepError, device_count, LIB_PATH
from .exceptions import NoConvergenceException, LostOrthogonalityException
from .nep import (NEP, AbstractSPMF, SPMF_NEP, DEP, PEP, SumNEP, DerSPMF, shift_and_scale, SPMFDevice, Mder_NEP,
                  LowRankMatrixAndFunction, LowRankFactorizedNEP,
                  to_dev, to_host)
from .linsolvers import (LinSolver, FactorizeLinSolver, BackslashLinSolver, FactorizeLinSolverCreator,
                         BackslashLinSolverCreator, DefaultLinSolverCreator, create_linsolver, lin_solve,
                         LinSolverCache, DeviceLU, HostLUPool, GMRESLinSolver, GMRESLinSolverCreator, seed_plan_from_rank0)
from .errmeasure import (Errmeasure, ResidualErrmeasure, StandardSPMFErrmeasure, DefaultErrmeasure,
                         estimate_error, estimate_errors)
from .dense import gemm_ts, orthogonalize_and_normalize, DGKS, CGS, MGS
from .iar import iar
from .tiar import tiar
from .iar_chebyshev import iar_chebyshev
from .ilan import ilan
from .nlar import nlar, residual_eigval_sorter, default_eigval_sorter
from .jd import jd_betcke, jd_eig_sorter
from .newton import resinv, quasinewton, augnewton, compute_rf, armijo_rule, ScalarNewtonInnerSolver
from .projection import (Proj_SPMF_NEP, create_proj_NEP, inner_solve, InnerSolver, DefaultInnerSolver, IARInnerSolver,
                         NewtonInnerSolver, IARChebInnerSolver, PolyeigInnerSolver, polyeig)
from .nleigs import nleigs, NleigsSolutionDetails
from . import rk_helper
from .contour import (contour_beyn, contour_block_SS, integrate_interval, MatrixIntegrator, MatrixTrapezoidal,
                      MatrixTrapezoidalSharded, probe_block)
from .comm import DeviceComm, HostStagedComm
from . import gallery
from . import wep_linsolvers
from .wep_linsolvers import (WEPLinSolverCreator, WEPFactorizedLinSolver, WEPBackslashLinSolver, WEPGMRESLinSolver,
                             wep_generate_preconditioner, construct_WEP_schur_complement)
from .gallery import nep_gallery
