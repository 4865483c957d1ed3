"""Importable alias for the ``tg-pose_amd/`` package directory.

The product lives in ``tg-pose_amd/`` (the layout the build contract names); a hyphen is not a
valid Python identifier, so this stub points the ``tgpose_amd`` package at that directory and
executes its ``__init__``.  ``import tgpose_amd`` is therefore ``tg-pose_amd``.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "tg-pose_amd")
if not _os.path.isdir(_real):  # pragma: no cover
    raise ImportError("tgpose_amd: package directory %r is missing" % _real)
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
