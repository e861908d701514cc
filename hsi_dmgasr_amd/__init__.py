"""Importable alias of the product package.

The product lives in ``hsi-dmgasr_amd/`` (the project's name); a hyphen is not a valid Python identifier,
so this stub only points the import system at that directory.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "hsi-dmgasr_amd")
__path__.insert(0, _real)

from .precision import get_default_precision, set_default_precision  # noqa: E402,F401
