"""Import alias for the package directory `algebraicmultigrid.jl_amd/`.

The directory name required by the project layout contains a dot, which Python's
import statement cannot spell; this shim loads it under the module name
`amg_amd` (`import amg_amd as AMG`).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "algebraicmultigrid.jl_amd")
_spec = importlib.util.spec_from_file_location("amg_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["amg_amd"] = _mod
_spec.loader.exec_module(_mod)
