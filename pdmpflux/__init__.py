"""Import shim: the product package lives in the directory ``pdmpflux.jl_amd/`` (a name Python's
import statement cannot spell).  ``import pdmpflux.jl_amd`` resolves through this shim."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pdmpflux.jl_amd")
if "pdmpflux.jl_amd" not in _sys.modules:
    _spec = _u.spec_from_file_location("pdmpflux.jl_amd", _os.path.join(_dir, "__init__.py"),
                                       submodule_search_locations=[_dir])
    _mod = _u.module_from_spec(_spec)
    _sys.modules["pdmpflux.jl_amd"] = _mod
    _spec.loader.exec_module(_mod)
jl_amd = _sys.modules["pdmpflux.jl_amd"]
