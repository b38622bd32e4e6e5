"""`envmap` emitter (python/scene_config.py:102,152 use it in every paper scene;
call sites volpathsimple.py:273 pdf_direction, :284 eval, :419 sample_emitter_direction).

Mitsuba's envmap plugin is absent from /root/reference (parity unpinned), so the oracle is pinned
by what the domain offers:
  1. the sampling density integrates to 1 over the sphere and sampling is consistent with it
     (importance-sampled and uniformly-sampled estimates of the integral of Le agree);
  2. radiance / pdf times pdf reproduces eval at the sampled direction;
  3. a constant-valued map equals the `constant` emitter: white furnace L == Le;
  4. NEE + MIS against a strongly non-uniform map agrees with an independent textbook path tracer
     that only ever evaluates the map on escape (the role of Mitsuba's `volpath`,
     tests/test_integrators.py:222-257);
  5. hide_emitters removes exactly the directly visible background (volpathsimple.py:268)."""
import numpy as np
import pytest

from conftest import props_for


def _blob_map(h=16, w=32, seed=0):
    rng = np.random.default_rng(seed)
    pix = rng.uniform(0.05, 1.0, size=(h, w, 3)).astype(np.float32)
    pix[3:6, 4:9] += 20.0          # a sun
    pix[h - 5:, :] = 0.0           # black ground: zero-probability region
    return pix


def _sphere_dirs(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def test_atan2_accuracy(oracle):
    rng = np.random.default_rng(1)
    y, x = rng.normal(size=5000).astype(np.float32), rng.normal(size=5000).astype(np.float32)
    got = np.array([oracle.lib().drto_atan2f(float(a), float(b)) for a, b in zip(y, x)])
    assert np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 5e-7
    L = oracle.lib()
    assert L.drto_atan2f(0.0, 0.0) == 0.0 and L.drto_atan2f(0.0, 1.0) == 0.0
    assert abs(L.drto_atan2f(0.0, -1.0) - np.pi) < 1e-6 and abs(L.drto_atan2f(1.0, 0.0) - np.pi / 2) < 1e-6
    assert abs(L.drto_atan2f(-1.0, 0.0) + np.pi / 2) < 1e-6


def test_tables_and_density(oracle, uivr):
    em = uivr.EnvmapEmitter(pixels=_blob_map(), scale=2.0, to_world=uivr.EnvmapEmitter.rotation_y(30.0))
    marg, cond = oracle.envmap_tables(em)
    assert marg[0] == 0 and marg[-1] == 1 and np.all(np.diff(marg) >= 0)
    assert np.all(cond[:, 0] == 0) and np.all(cond[:, -1] == 1) and np.all(np.diff(cond, axis=1) >= 0)
    # black rows further than one texel from a lit one get probability zero
    assert np.all(np.diff(marg)[-3:] == 0)
    dirs = _sphere_dirs(40000, 2)
    pdf = np.array([oracle.envmap_pdf(em, d) for d in dirs])
    assert abs(pdf.mean() * 4 * np.pi - 1.0) < 0.05          # MC, sun-dominated variance
    Le = np.array([oracle.envmap_eval(em, d) for d in dirs])
    assert np.all(Le[pdf == 0] == 0)                         # pdf > 0 wherever the lookup is non-zero
    rng = np.random.default_rng(3)
    w = []
    for u1, u2 in rng.uniform(size=(8000, 2)).astype(np.float32):
        d, p, wt = oracle.envmap_sample(em, u1, u2)
        assert p > 0 and abs(np.linalg.norm(d) - 1) < 1e-6
        assert abs(p - oracle.envmap_pdf(em, d)) <= 1e-6 * p
        np.testing.assert_allclose(wt * p, oracle.envmap_eval(em, d), rtol=1e-5, atol=1e-6)
        w.append(wt)
    uniform_estimate = Le.mean(axis=0) * 4 * np.pi
    np.testing.assert_allclose(np.mean(w, axis=0), uniform_estimate, rtol=0.08)


def test_lookup_orientation(oracle, uivr):
    """Row 0 is the +Y pole, u = 0 faces -Z and increases towards +X (Mitsuba's convention:
    d = (sin phi sin theta, cos theta, -cos phi sin theta))."""
    h, w = 8, 16
    pix = np.zeros((h, w, 3), np.float32)
    pix[0, :, 0] = 1.0                      # top row red
    pix[h // 2, w // 4, 1] = 1.0            # a green texel on the equator at u = 0.25 + half a texel
    em = uivr.EnvmapEmitter(pixels=pix)
    assert oracle.envmap_eval(em, [0, 1, 0])[0] > 0.9
    assert oracle.envmap_eval(em, [0, -1, 0]).max() == 0
    u, v = (w // 4 + 0.5) / w, (h // 2 + 0.5) / h
    phi, theta = 2 * np.pi * u, np.pi * v
    d = [np.sin(phi) * np.sin(theta), np.cos(theta), -np.cos(phi) * np.sin(theta)]
    np.testing.assert_allclose(oracle.envmap_eval(em, d), [0, 1, 0], atol=1e-5)
    assert d[0] > 0.9                       # u = 0.25 is +X
    # to_world rotates the map with the frame: 90 degrees about Y moves +X content to -Z
    em_r = uivr.EnvmapEmitter(pixels=pix, to_world=uivr.EnvmapEmitter.rotation_y(90.0))
    R = np.asarray(em_r.to_world, np.float64)
    np.testing.assert_allclose(oracle.envmap_eval(em_r, (R @ np.asarray(d)).astype(np.float32)), [0, 1, 0], atol=1e-4)


def test_constant_map_is_the_constant_emitter(oracle, uivr):
    scene = uivr.cube_test_scene(16, 16, density_scale=2.0)
    scene.medium.albedo[...] = 1.0
    rgb = np.float32([1.0, 0.8, 0.2])
    scene.emitter = uivr.EnvmapEmitter(pixels=np.broadcast_to(rgb, (8, 16, 3)).copy())
    spp = 256
    for variant in ("basic", "drt"):
        img = oracle.develop(oracle.render_primal(oracle.OracleScene(scene), props_for(variant), spp, 5)[0], spp)
        assert np.abs(img / rgb - 1).max() < 0.25            # white furnace, per pixel
        np.testing.assert_allclose(img.mean(axis=0), rgb, rtol=0.02)


def test_envmap_nee_matches_textbook_path_tracer(oracle, uivr):
    scene = uivr.cube_test_scene(16, 16, density_scale=2.0)
    scene.emitter = uivr.EnvmapEmitter(pixels=_blob_map(), scale=0.5, to_world=uivr.EnvmapEmitter.rotation_y(-40.0))
    osc = oracle.OracleScene(scene)
    spp = 4096
    a = oracle.develop(oracle.render_primal(osc, props_for("drt", rr_depth=999), spp, 1)[0], spp)
    b = oracle.develop(oracle.render_textbook(osc, props_for("drt"), 4 * spp, 2), 4 * spp)
    np.testing.assert_allclose(a.mean(axis=0), b.mean(axis=0), rtol=2e-2)
    no_nee = oracle.develop(oracle.render_primal(osc, props_for("drt", use_nee=False, rr_depth=999), spp, 3)[0], spp)
    np.testing.assert_allclose(no_nee.mean(axis=0), b.mean(axis=0), rtol=4e-2)
    # importance sampling pays: NEE + MIS is the lower-variance estimator under a sun
    assert np.abs(a - b).mean() < np.abs(no_nee - b).mean()


def test_hide_emitters_with_envmap(oracle, uivr):
    scene = uivr.cube_test_scene(16, 16, density_scale=2.0)
    scene.emitter = uivr.EnvmapEmitter(pixels=_blob_map(), scale=0.5)
    osc = oracle.OracleScene(scene)
    spp = 64
    shown = oracle.develop(oracle.render_primal(osc, props_for("drt"), spp, 9)[0], spp)
    hidden = oracle.develop(oracle.render_primal(osc, props_for("drt", hide_emitters=True), spp, 9)[0], spp)
    assert np.all(hidden <= shown + 1e-6) and hidden.sum() < shown.sum()
    # rays that miss the box see only the background: hidden there is exactly zero
    corner = hidden.reshape(16, 16, 3)[0, 0]
    assert np.all(corner == 0)
