"""Importable alias for the on-disk package directory ``ae-wavenet_amd/``.

The repo layout mandates a hyphenated package directory, which Python cannot
import by name.  This stub redirects the package search path to that directory
and executes its ``__init__`` so that ``import ae_wavenet_amd.engine`` etc.
resolve to ``ae-wavenet_amd/engine.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "ae-wavenet_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _fh
