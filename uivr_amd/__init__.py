"""Importable alias of the package directory `unbiased-inverse-volume-rendering_amd/`
(a hyphen is not a valid Python identifier): `import uivr_amd` returns that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_long = "unbiased-inverse-volume-rendering_amd"
_pkg = importlib.import_module(_long)
sys.modules[__name__] = _pkg
# the submodules too: `from uivr_amd.optimize import x` must find the module that is loaded already - importing it a
# second time under the alias would re-bind the package's attributes (e.g. `render`, the function, to `render`, the module)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_long + "."):
        sys.modules.setdefault(__name__ + _name[len(_long):], _mod)
