"""Importable alias for the package directory `off-policy_amd/` (a hyphen is not a valid identifier).

`import offpolicy_amd` loads `off-policy_amd/__init__.py` AS the package `offpolicy_amd` through importlib (a module spec whose
submodule search path is that directory) and puts it in `sys.modules` in place of this stub, so `offpolicy_amd.utils.rec_buffer`
etc. resolve to files under `off-policy_amd/` and `offpolicy_amd.__file__` / `__path__` are the real package's.
"""
import importlib.util as _util
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "off-policy_amd")
_spec = _util.spec_from_file_location(__name__, _os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
_mod = _util.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
