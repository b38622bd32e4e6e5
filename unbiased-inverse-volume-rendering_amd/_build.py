"""In-tree build of the native pieces (gfx950 only):

  csrc/libdrt_hip.so                  hipcc: kernels + C ABI (include/drt_hip.h) - the production library
  _drt_pybind.<abi>.so                g++:   pybind11 shim linked against it
  csrc/libdrt_hip_hooks.so            the same sources with -DDRT_TEST_HOOKS: kernel-variant selection, ablations,
  _drt_pybind_hooks.<abi>.so          simulated out-of-memory (drt_set_debug_flags) for the tests / profiling

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

HIP_SOURCES = ["drt_kernels.hip", "drt_deferred.hip", "drt_coop.hip", "drt_coop_super.hip", "drt_order.hip", "drt_sq.hip", "drt_nerf_tile.hip", "drt_own.hip", "drt_capi.cpp"]
# older generations of the tracer (round 1/2 state machine of whole flights, round 3 lane state machines with posted flights): no production call
# reaches them ("which call reaches which kernel"); the flavour with test hooks keeps them in lock-step with the oracle
HOOKS_ONLY_SOURCES = ["drt_wavefront.hip", "drt_super.hip"]
HIP_HEADERS = ["drt_device.h", "drt_launch.h", "drt_coop_tracer.h", "drt_coop_kernel.h", "drt_nerf_kernel.h", os.path.join(_ROOT, "include", "drt_hip.h")]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall"]

# per-unit flags behind HIP_FLAGS (measured, profiles/r05_sq_experiments.txt): the queued tracer's envmap instantiations come out 12 % smaller at -O2
# (56.8 instead of 64.7 KB for the adjoint kernel) and 1.5 % faster (headline + envmap + factor 8: 755-757 -> 766-768 Msamples/s); the others the same
UNIT_FLAGS = {"drt_sq.hip": ["-O2"]}

_EXT = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
LIB_PATH = os.path.join(_CSRC, "libdrt_hip.so")                 # production library: no test hooks
PYBIND_PATH = os.path.join(_PKG, "_drt_pybind" + _EXT)
HOOKS_LIB_PATH = os.path.join(_CSRC, "libdrt_hip_hooks.so")     # the same sources with -DDRT_TEST_HOOKS (variant tests, ablations)
HOOKS_PYBIND_PATH = os.path.join(_PKG, "_drt_pybind_hooks" + _EXT)


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


def _flavours(hooks):
    """(library path, object directory, extra -D flags) of the flavours to build."""
    out = [(LIB_PATH, os.path.join(_CSRC, "_obj"), [])]
    if hooks:
        out.append((HOOKS_LIB_PATH, os.path.join(_CSRC, "_obj_hooks"), ["-DDRT_TEST_HOOKS=1"]))
    return out


def build_hip(force: bool = False, verbose: bool = False, hooks: bool = True):
    """Every translation unit of every flavour is compiled on its own (in parallel: the tracing kernels take about a
    minute each) into csrc/_obj*/, then linked; only units whose sources changed are recompiled."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [h if os.path.isabs(h) else os.path.join(_CSRC, h) for h in HIP_HEADERS]
    env_defs = [f"-D{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("DRT_") and v.lstrip("-").isdigit()]
    compile_flags = [f for f in HIP_FLAGS if f != "-shared"]
    jobs, links = [], []
    for lib, objdir, defs in _flavours(hooks):
        os.makedirs(objdir, exist_ok=True)
        stamp = os.path.join(objdir, "flags.txt")
        flags_now = " ".join(HIP_FLAGS + env_defs + defs + [f"{k}:{' '.join(v)}" for k, v in sorted(UNIT_FLAGS.items())])
        stale = force or not os.path.exists(stamp) or open(stamp).read() != flags_now
        srcs = [os.path.join(_CSRC, s) for s in HIP_SOURCES + (HOOKS_ONLY_SOURCES if defs else [])]
        objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]
        todo = [(s, o) for s, o in zip(srcs, objs) if stale or _newer(o, [s] + hdrs)]
        jobs += [(s, o, defs) for s, o in todo]
        links.append((lib, objs, bool(todo), stamp, flags_now))

    def compile_one(job):
        src, obj, defs = job
        cmd = [_hipcc()] + compile_flags + UNIT_FLAGS.get(os.path.basename(src), []) + env_defs + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=_CSRC)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(compile_one, jobs))
    for lib, objs, changed, stamp, flags_now in links:
        if changed or _newer(lib, objs):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=_CSRC)
        with open(stamp, "w") as f:
            f.write(flags_now)
    return [l[0] for l in links]


def build_pybind(force: bool = False, verbose: bool = False, hooks: bool = True):
    import pybind11
    src = os.path.join(_CSRC, "drt_pybind.cpp")
    out = []
    for lib, target, name in [(LIB_PATH, PYBIND_PATH, "_drt_pybind")] + ([(HOOKS_LIB_PATH, HOOKS_PYBIND_PATH, "_drt_pybind_hooks")] if hooks else []):
        deps = [src, os.path.join(_ROOT, "include", "drt_hip.h"), lib]
        if force or _newer(target, deps):
            libname = os.path.basename(lib)[3:-3]
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", f"-DDRT_PYBIND_NAME={name}",
                   "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
                   src, "-o", target, "-L" + _CSRC, "-l" + libname, "-Wl,-rpath,$ORIGIN/csrc"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=_CSRC)
        out.append(target)
    return out


def build_all(force: bool = False, verbose: bool = False, hooks: bool = True):
    return build_hip(force, verbose, hooks), build_pybind(force, verbose, hooks)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
