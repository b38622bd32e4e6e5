"""Colour grids (albedo; the nerf integrator's emission) on their OWN lattice: Mitsuba's GridVolume::eval interpolates every grid on
its own resolution, and the reference's janga-smoke pairs a 264 x 136 x 136 density with 256 x 128 x 128 albedo / emission grids
(python/scene_config.py:108-110).  CPU checks of the oracle's restatement (drto_medium::res_colour):
  * the albedo lookup equals a numpy restatement of the cell-centred, clamped trilinear lookup on the colour lattice;
  * equal lattices given explicitly change nothing, bit for bit;
  * the free-flight estimator's albedo gradient on the colour lattice equals central finite differences of the primal at the same
    seed (path replay: the paths do not depend on the albedo), and the gradient buffer has the colour grid's shape.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import props_for


def _scene(uivr, colour_shape=(4, 5, 6), film=12):
    rng = np.random.default_rng(3)
    st = (rng.random((7, 5, 9, 1), dtype=np.float32) * 2.5).astype(np.float32)
    st[rng.random(st.shape) < 0.3] = 0.0
    al = (rng.random(tuple(colour_shape) + (3,), dtype=np.float32) * 0.8 + 0.1).astype(np.float32)
    medium = uivr.GridMedium(sigma_t=st, albedo=al, emission=al.copy(), bbox_min=(-1.0, -0.6, -0.8), bbox_max=(1.2, 0.9, 0.7), scale=1.4)
    sensor = uivr.PerspectiveSensor(origin=(3.0, 2.0, 3.5), target=(0.1, 0.1, 0.0), fov=35.0, width=film, height=film - 3)
    return uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter((0.9, 1.0, 0.7)), sensors=[sensor])


def _trilerp_np(grid, bmin, bmax, p):
    """GridVolume::eval restated in float64: q = (p - bmin) / ext * res - 0.5, clamped corners."""
    z, y, x = grid.shape[:3]
    out = np.zeros(3)
    q = [(p[a] - bmin[a]) / (bmax[a] - bmin[a]) * r - 0.5 for a, r in zip(range(3), (x, y, z))]
    i0 = [int(np.floor(v)) for v in q]
    w = [v - np.floor(v) for v in q]
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                ix = min(max(i0[0] + dx, 0), x - 1); iy = min(max(i0[1] + dy, 0), y - 1); iz = min(max(i0[2] + dz, 0), z - 1)
                wt = (w[0] if dx else 1 - w[0]) * (w[1] if dy else 1 - w[1]) * (w[2] if dz else 1 - w[2])
                out += wt * grid[iz, iy, ix].astype(np.float64)
    return out


def test_albedo_lookup_on_its_own_lattice(oracle, uivr):
    scene = _scene(uivr)
    osc = oracle.OracleScene(scene)
    assert tuple(osc.medium.res_colour) == (6, 5, 4) and tuple(osc.medium.res) == (9, 5, 7)
    rng = np.random.default_rng(1)
    bmin, bmax = np.array(scene.medium.bbox_min), np.array(scene.medium.bbox_max)
    for _ in range(300):
        p = (bmin + rng.random(3) * (bmax - bmin)).astype(np.float32)
        out = (C.c_float * 3)()
        oracle.lib().drto_eval_albedo(C.byref(osc.medium), (C.c_float * 3)(*p), out)
        np.testing.assert_allclose(np.array(out[:]), _trilerp_np(scene.medium.albedo, bmin, bmax, p.astype(np.float64)), atol=2e-6)


def test_equal_lattices_are_the_default_bit_for_bit(oracle, uivr):
    scene = _scene(uivr, colour_shape=(7, 5, 9))
    osc = oracle.OracleScene(scene)
    assert tuple(osc.medium.res_colour) == (0, 0, 0)
    a = oracle.h1_step(osc, props_for("drt"), 4, 11)
    osc2 = oracle.OracleScene(scene)
    osc2.medium.res_colour = (C.c_int32 * 3)(9, 5, 7)                       # stated explicitly: the same lattice
    b = oracle.h1_step(osc2, props_for("drt"), 4, 11)
    np.testing.assert_array_equal(a["L"].view(np.uint32), b["L"].view(np.uint32))
    np.testing.assert_array_equal(a["grad_albedo"], b["grad_albedo"])
    np.testing.assert_array_equal(a["grad_sigma_t"], b["grad_sigma_t"])


def test_free_flight_albedo_gradient_on_the_colour_lattice_equals_fd(oracle, uivr):
    scene = _scene(uivr)
    props, spp, seed = props_for("basic"), 64, 5
    ref = oracle.h1_step(oracle.OracleScene(scene), props, spp, seed)
    assert ref["grad_albedo"].shape == (4, 5, 6, 3) and ref["grad_sigma_t"].shape == (7, 5, 9, 1)
    g = ref["grad_albedo"]
    idx = np.argsort(-np.abs(g).reshape(-1))[:6]                            # the six largest entries
    eps = 2e-3
    for flat in idx:
        ijk = np.unravel_index(flat, g.shape)
        vals = []
        for sgn in (+1, -1):
            sc = _scene(uivr)
            sc.medium.albedo[ijk] += sgn * eps
            vals.append(oracle.h1_step(oracle.OracleScene(sc), props, spp, seed)["loss"])
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert fd == pytest.approx(g[ijk], rel=2e-2, abs=1e-7), (ijk, fd, g[ijk])
