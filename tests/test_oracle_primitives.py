"""Known-answer tests that pin the oracle's building blocks (CPU only).

The reference holds no golden vectors for this path, so the
[M3-ext] primitives are pinned against published / analytic answers:
  * PCG32: the reference output of pcg32-demo (pcg-random.org, seed 42 / stream 54)
  * sample_tea_32: the Python restatement in the host package (two independent
    implementations of the published TEA round function)
  * log / sincos polynomials: against numpy in float64
  * trilinear lookup: against a numpy restatement of Mitsuba's cell-centred,
    clamped GridVolume lookup
"""
import ctypes as C

import numpy as np
import pytest


def test_pcg32_reference_vector(oracle):
    out = (C.c_uint32 * 6)()
    oracle.lib().drto_pcg32_raw(42, 54, 6, out)
    assert [hex(v) for v in out] == ['0xa15c02b7', '0x7b47f409', '0xba1d3330',
                                     '0x83d2f293', '0xbfa4784b', '0xcbed606e']


def test_tea32_matches_host_restatement(oracle, uivr):
    rng = np.random.default_rng(0)
    for v0, v1 in rng.integers(0, 2 ** 32, size=(200, 2), dtype=np.uint64):
        o1 = C.c_uint32()
        o0 = oracle.lib().drto_tea32(int(v0), int(v1), C.byref(o1))
        assert (o0, o1.value) == uivr.sample_tea_32(int(v0), int(v1))
    # seeds of iteration 0 of the optimisation loop (optimize.py:327-328, base_seed 988378)
    assert uivr.sample_tea_32(0, 988378)[0] != uivr.sample_tea_32(1, 988378)[0]


def test_sampler_floats_in_unit_interval_and_uniform(oracle):
    buf = np.zeros(100000, dtype=np.float32)
    oracle.lib().drto_pcg32_floats(1234, 7, buf.size, buf.ctypes.data_as(C.POINTER(C.c_float)))
    assert buf.min() >= 0.0 and buf.max() < 1.0
    assert abs(buf.mean() - 0.5) < 5e-3 and abs(buf.var() - 1 / 12) < 2e-3
    # 23 mantissa bits: every value is a multiple of 2^-23
    assert np.all(buf * 2.0 ** 23 == np.round(buf * 2.0 ** 23))


def test_logf_accuracy(oracle):
    k = np.arange(1, 1 << 23, 1013, dtype=np.int64)
    x = (k.astype(np.float64) / float(1 << 23)).astype(np.float32)
    got = np.array([oracle.lib().drto_logf(float(v)) for v in x], dtype=np.float64)
    ref = np.log(x.astype(np.float64))
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
    assert err[x < 0.999].max() < 4e-7
    assert np.abs(got - ref).max() < 2e-7 * 16    # absolute near x = 1
    assert oracle.lib().drto_logf(1.0) == 0.0


def test_sincos_and_sphere(oracle):
    L = oracle.lib()
    s, c = C.c_float(), C.c_float()
    u = np.random.default_rng(1).random(5000, dtype=np.float32)
    for v in u:
        L.drto_sincos_2pi(float(v), C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(2 * np.pi * float(v))) < 4e-7
        assert abs(c.value - np.cos(2 * np.pi * float(v))) < 4e-7
    out = (C.c_float * 3)()
    acc = np.zeros(3)
    for a, b in np.random.default_rng(2).random((20000, 2), dtype=np.float32):
        L.drto_uniform_sphere(float(a), float(b), out)
        d = np.array(out[:], dtype=np.float64)
        assert abs(np.linalg.norm(d) - 1.0) < 1e-6
        acc += d
    assert np.abs(acc / 20000).max() < 0.02     # uniform: zero mean direction


def _numpy_trilerp(grid, bmin, bmax, p):
    """Mitsuba GridVolume / Dr.Jit texture lookup restated in numpy (float64)."""
    z, y, x = grid.shape[:3]
    res = np.array([x, y, z], dtype=np.float64)
    q = (np.asarray(p, np.float64) - bmin) / (bmax - bmin) * res - 0.5
    i0 = np.floor(q).astype(int)
    w1 = q - i0
    w0 = 1 - w1
    a = np.clip(i0, 0, res.astype(int) - 1)
    b = np.clip(i0 + 1, 0, res.astype(int) - 1)
    out = 0
    for dz, wz in ((a[2], w0[2]), (b[2], w1[2])):
        for dy, wy in ((a[1], w0[1]), (b[1], w1[1])):
            for dx, wx in ((a[0], w0[0]), (b[0], w1[0])):
                out = out + wz * wy * wx * grid[dz, dy, dx].astype(np.float64)
    return out


def test_grid_lookup_matches_numpy(oracle, uivr):
    rng = np.random.default_rng(3)
    sigma_t = rng.random((6, 7, 5, 1), dtype=np.float32) * 3
    albedo = rng.random((6, 7, 5, 3), dtype=np.float32)
    bmin, bmax = np.array([-1.0, 0.0, 2.0]), np.array([1.5, 3.0, 2.5])
    medium = uivr.GridMedium(sigma_t=sigma_t, albedo=albedo, bbox_min=tuple(bmin), bbox_max=tuple(bmax), scale=1.7)
    osc = oracle.OracleScene(uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter(), sensors=[]), sensor_index=None)
    L = oracle.lib()
    out = (C.c_float * 3)()
    for p in bmin + (bmax - bmin) * (rng.random((500, 3)) * 1.1 - 0.05):
        pc = (C.c_float * 3)(*p)
        got = L.drto_eval_sigma_t(C.byref(osc.medium), pc)
        assert abs(got - 1.7 * _numpy_trilerp(sigma_t, bmin, bmax, np.float32(p))[0]) < 2e-5
        L.drto_eval_albedo(C.byref(osc.medium), pc, out)
        np.testing.assert_allclose(out[:], _numpy_trilerp(albedo, bmin, bmax, np.float32(p)), atol=2e-6)
    # voxel centres reproduce the data exactly; the majorant is scale * max
    c = bmin + (bmax - bmin) * (np.array([2, 3, 4]) + 0.5) / np.array([5, 7, 6])
    assert abs(L.drto_eval_sigma_t(C.byref(osc.medium), (C.c_float * 3)(*c)) - 1.7 * sigma_t[4, 3, 2, 0]) < 1e-5
    assert L.drto_majorant(C.byref(osc.medium)) == pytest.approx(1.7 * float(sigma_t.max()), rel=1e-6)


def test_box_hit_cases(oracle, uivr):
    scene = uivr.cube_test_scene(8, 8)     # box [-0.5, 1.5]^3
    osc = oracle.OracleScene(scene)
    L = oracle.lib()
    t, n = C.c_float(), (C.c_float * 3)()

    def hit(o, d):
        v = L.drto_box_hit(C.byref(osc.medium), (C.c_float * 3)(*o), (C.c_float * 3)(*d), C.byref(t), n)
        return v, t.value, tuple(n[:])

    assert hit((-2, 0.5, 0.5), (1, 0, 0)) == (1, 1.5, (-1.0, 0.0, 0.0))      # entry face from outside
    assert hit((0.5, 0.5, 0.5), (0, 0, 1)) == (1, 1.0, (0.0, 0.0, 1.0))      # exit face from inside
    assert hit((0.5, 0.5, 0.5), (0, -1, 0)) == (1, 1.0, (0.0, -1.0, 0.0))
    assert hit((-2, 0.5, 0.5), (-1, 0, 0))[0] == 0                           # pointing away
    assert hit((-2, 5.0, 0.5), (1, 0, 0))[0] == 0                            # parallel, outside the slab
    assert hit((3, 3, 3), (1, 1, 1))[0] == 0                                 # behind the origin


def test_alt_seed_is_lane0_draw(oracle, uivr):
    """volpathsimple.py:99-107: alt seed = tea32(bits of lane 0's 2nd (4th with film
    position draws) float, 1)[0]."""
    L = oracle.lib()
    buf = np.zeros(4, dtype=np.float32)
    L.drto_pcg32_floats(777, 0, 4, buf.ctypes.data_as(C.POINTER(C.c_float)))
    assert L.drto_alt_seed(777, 0) == uivr.sample_tea_32(int(buf[1:2].view(np.uint32)[0]), 1)[0]
    assert L.drto_alt_seed(777, 1) == uivr.sample_tea_32(int(buf[3:4].view(np.uint32)[0]), 1)[0]
