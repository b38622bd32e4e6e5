"""Importable alias of the package directory `unbiased-inverse-volume-rendering_amd/`
(a hyphen is not a valid Python identifier): `import uivr_amd` returns that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("unbiased-inverse-volume-rendering_amd")
sys.modules[__name__] = _pkg
