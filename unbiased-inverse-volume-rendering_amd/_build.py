"""In-tree build of the native pieces (gfx950 only):

  csrc/libdrt_hip.so                  hipcc: kernels + C ABI (include/drt_hip.h)
  _drt_pybind.<abi>.so                g++:   pybind11 shim linked against it

The arithmetic specification requires -ffp-contract=off; hardware fp32
atomics need -munsafe-fp-atomics.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_ROOT = os.path.dirname(_PKG)

HIP_SOURCES = ["drt_kernels.hip", "drt_wavefront.hip", "drt_deferred.hip", "drt_coop.hip", "drt_fused.hip", "drt_capi.cpp"]
HIP_HEADERS = ["drt_device.h", "drt_launch.h", "drt_coop_tracer.h", os.path.join(_ROOT, "include", "drt_hip.h")]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall"]

LIB_PATH = os.path.join(_CSRC, "libdrt_hip.so")
PYBIND_PATH = os.path.join(_PKG, "_drt_pybind" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libdrt_hip.so)")


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Every translation unit is compiled on its own (in parallel: the tracing kernels take about a minute
    each) into csrc/_obj/, then linked; only units whose sources changed are recompiled."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(_CSRC, s) for s in HIP_SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(_CSRC, h) for h in HIP_HEADERS]
    extra = [f"-D{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("DRT_") and v.lstrip("-").isdigit()]
    objdir = os.path.join(_CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")
    flags_now = " ".join(HIP_FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    compile_flags = [f for f in HIP_FLAGS if f != "-shared"]

    def obj_of(src):
        return os.path.join(objdir, os.path.basename(src) + ".o")

    def compile_one(src):
        cmd = [_hipcc()] + compile_flags + extra + ["-c", src, "-o", obj_of(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=_CSRC)

    todo = [s for s in srcs if force or _newer(obj_of(s), [s] + hdrs)]
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(compile_one, todo))
        with open(stamp, "w") as f:
            f.write(flags_now)
    objs = [obj_of(s) for s in srcs]
    if todo or _newer(LIB_PATH, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=_CSRC)
    return LIB_PATH


def build_pybind(force: bool = False, verbose: bool = False) -> str:
    import pybind11
    src = os.path.join(_CSRC, "drt_pybind.cpp")
    deps = [src, os.path.join(_ROOT, "include", "drt_hip.h"), LIB_PATH]
    if force or _newer(PYBIND_PATH, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
               src, "-o", PYBIND_PATH,
               "-L" + _CSRC, "-ldrt_hip", "-Wl,-rpath,$ORIGIN/csrc"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=_CSRC)
    return PYBIND_PATH


def build_all(force: bool = False, verbose: bool = False):
    return build_hip(force, verbose), build_pybind(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
