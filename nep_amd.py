"""Import shim: the product package lives in the directory `nonlineareigenproblems.jl_amd/`
(the name the build contract fixes), which is not a valid Python identifier.  `import nep_amd`
loads that directory as the package `nep_amd`."""
import importlib.util
import os
import sys

# OpenBLAS' idle worker threads spin ~0.1 s after every job before they sleep; the drivers call small LAPACK routines every few
# hundred microseconds, so the workers would burn CPUs for the whole run (and a container's CPU quota with them).  Only takes
# effect when OpenBLAS has not been loaded yet (i.e. `import nep_amd` before `import numpy`); harmless otherwise.
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "12")

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nonlineareigenproblems.jl_amd")
_spec = importlib.util.spec_from_file_location("nep_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nep_amd"] = _mod
_spec.loader.exec_module(_mod)
