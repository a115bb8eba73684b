"""Importable alias for the package directory `off-policy_amd/` (a hyphen is not a valid identifier).

`import offpolicy_amd` executes `off-policy_amd/__init__.py` with `__path__` pointing at that directory, so
`offpolicy_amd.utils.rec_buffer` etc. resolve to files under `off-policy_amd/`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "off-policy_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
