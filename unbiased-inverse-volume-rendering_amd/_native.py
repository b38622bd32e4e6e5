"""Loader of the native extension.  There is no Python/CPU fallback: if the
in-tree build of libdrt_hip.so / _drt_pybind is missing this raises, loudly."""
from __future__ import annotations

import importlib
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_mod = None


class NativeExtensionMissing(ImportError):
    pass


def native():
    """The pybind11 module `_drt_pybind` (thin shim over include/drt_hip.h)."""
    global _mod
    if _mod is None:
        try:
            _mod = importlib.import_module(__package__ + "._drt_pybind")
        except ImportError as e:  # pragma: no cover - exercised only on broken installs
            raise NativeExtensionMissing(
                "the HIP extension of the DRT integrator is not built: run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, gfx950). "
                f"Original error: {e}") from e
    return _mod


def library_path() -> str:
    return os.path.join(_PKG, "csrc", "libdrt_hip.so")
