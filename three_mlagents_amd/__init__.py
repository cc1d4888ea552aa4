"""Importable alias of the `three-mlagents_amd/` package directory (a hyphen is not a valid module name).

`import three_mlagents_amd` executes three-mlagents_amd/__init__.py with this module's __path__ pointing there,
so `three_mlagents_amd.harness`, `.ppo`, `.vec_env`, ... resolve to the files in that directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "three-mlagents_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
