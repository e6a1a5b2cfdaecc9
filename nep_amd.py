"""Import shim: the product package lives in the directory `nonlineareigenproblems.jl_amd/`
(the name the build contract fixes), which is not a valid Python identifier.  `import nep_amd`
loads that directory as the package `nep_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nonlineareigenproblems.jl_amd")
_spec = importlib.util.spec_from_file_location("nep_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nep_amd"] = _mod
_spec.loader.exec_module(_mod)
