"""Import shim: exposes the package directory `jets.jl_amd/` (its name has a dot, so Python cannot
import it by name) as the module `jets_jl_amd`."""
import importlib.util as _u
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "jets.jl_amd")
_spec = _u.spec_from_file_location("jets_jl_amd", _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["jets_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
