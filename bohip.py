"""`import bohip` -> the package in ./bayesianoptimization.jl_amd/ (whose directory name, fixed by the
project layout, is not a valid Python identifier).  This shim loads it under the module name `bohip`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bayesianoptimization.jl_amd")
_spec = importlib.util.spec_from_file_location("bohip", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bohip"] = _mod
_spec.loader.exec_module(_mod)
