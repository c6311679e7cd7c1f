"""Importable alias of the product package.

The product lives in ``accelerating-t2i-ar-with-sjd_amd/`` (the name the build contract fixes); a
hyphenated directory cannot be imported by name, so this alias points its ``__path__`` there:
``import sjd_amd.engine`` loads ``accelerating-t2i-ar-with-sjd_amd/engine.py``.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "accelerating-t2i-ar-with-sjd_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
