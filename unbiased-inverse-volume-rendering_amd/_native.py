"""Loader of the native extension.  There is no Python/CPU fallback: if the
in-tree build of libdrt_hip.so / _drt_pybind is missing this raises, loudly."""
from __future__ import annotations

import importlib
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_mods = {}


class NativeExtensionMissing(ImportError):
    pass


def native(hooks: bool = False):
    """The pybind11 module over include/drt_hip.h: `_drt_pybind` (production library) or, with `hooks`,
    `_drt_pybind_hooks` (the build with test hooks: drt_set_debug_flags accepts non-zero flags)."""
    name = "._drt_pybind_hooks" if hooks else "._drt_pybind"
    if name not in _mods:
        try:
            _mods[name] = importlib.import_module(__package__ + name)
        except ImportError as e:  # pragma: no cover - exercised only on broken installs
            raise NativeExtensionMissing(
                "the HIP extension of the DRT integrator is not built: run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, gfx950). "
                f"Original error: {e}") from e
    return _mods[name]


def library_path(hooks: bool = False) -> str:
    return os.path.join(_PKG, "csrc", "libdrt_hip_hooks.so" if hooks else "libdrt_hip.so")
