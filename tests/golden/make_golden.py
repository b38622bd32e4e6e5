#!/usr/bin/env python3
"""Generates tests/golden/cube_golden.npz with the CPU oracle.

The reference stores no golden vectors for this path and cannot be imported here
(Mitsuba 3 / Dr.Jit absent), so these fixtures are produced by the
build's own oracle (oracle/drt_oracle.c) on the reference's fully specified 3^3 cube
fixture (tests/test_integrators.py:19-116) and pin it against regressions:
tests/test_oracle_kat.py (CPU) checks the oracle against the committed file, and
tests/test_gpu_film_shapes.py::test_hip_path_equals_the_committed_golden_* compare the
HIP path with the committed file directly (without calling the oracle).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import uivr_amd as u                      # noqa: E402  (scene dataclasses only)
from conftest import VARIANTS, props_for  # noqa: E402
from oracle import binding as ob          # noqa: E402

RES, SPP, SEED, SCALE = 16, 8, 12345, 2.0


def main():
    scene = u.cube_test_scene(RES, RES, density_scale=SCALE)
    out = dict(res=RES, spp=SPP, seed=SEED, density_scale=SCALE)
    for name in VARIANTS:
        r = ob.h1_step(ob.OracleScene(scene), props_for(name), SPP, SEED, n_threads=1)
        out[f"{name}/image"] = r["image"]
        out[f"{name}/L"] = r["L"]
        out[f"{name}/loss"] = np.float64(r["loss"])
        out[f"{name}/grad_sigma_t"] = r["grad_sigma_t"]
        out[f"{name}/grad_albedo"] = r["grad_albedo"]
        out[f"{name}/counters"] = np.array([r["counters"][k] for k in sorted(r["counters"])], dtype=np.int64)
    out["counter_names"] = np.array(sorted(r["counters"]))
    # explicit-ray (batched flow) vector: 64 rays through the box
    rng = np.random.default_rng(2024)
    o = (rng.normal(size=(64, 3)) * 0.2 + np.array([3.0, 2.0, -3.0])).astype(np.float32)
    t = rng.random((64, 3)).astype(np.float32) * 2.0 - 0.5
    d = t - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    osc = ob.OracleScene(scene, sensor_index=None)
    L, _ = ob.render_primal(osc, props_for("drt"), 4, 99, rays_o=o, rays_d=d, n_threads=1)
    out["rays/o"], out["rays/d"], out["rays/L"] = o, d, L
    path = os.path.join(HERE, "cube_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
