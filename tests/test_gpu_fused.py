"""BASELINE config 5: nerf emission-absorption (python/integrators/nerf.py:47-165, 128 queries) FUSED with DRT
scattering (python/integrators/volpathsimple.py) in one pass over one interleaved four-channel [sigma_t, r, g, b] grid
(drt_fused_render_* in csrc/drt_capi.cpp - since round 5 two dense passes: the nerf march, adjoint in csrc/drt_nerf_tile.hip, beside the volpathsimple half in the production tracers; the reference's scenes bind ONE asset as albedo and emission, python/scene_config.py:109-110).

Oracle: the two restated integrators run on the same rays / streams (oracle.binding.fused_render_*).
  * radiance of both halves BIT-EXACT per ray, event counters equal, gradients within 2e-4 max|oracle|;
  * the fused pass equals the two stand-alone HIP integrators (same grids);
  * at the registered size (256^3, 512^2 x 32 spp, 128 queries): determinism, linearity of the adjoint, a window
    of rays against the oracle.
"""
import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _fused(uivr, variant="drt", **nerf):
    d = {"type": "nerf+volpathsimple"}
    d.update(props_for(variant))
    d.update(nerf)
    return uivr.load_dict(d)


def _close(g_hip, g_ref, what):
    ref = torch.from_numpy(g_ref).to(g_hip.device)
    tol = GRAD_RTOL * float(ref.abs().max()) + 1e-9
    err = float((g_hip.double() - ref).abs().max())
    assert float(ref.abs().max()) > 0, what
    assert err <= tol, f"{what}: max abs err {err:.3e} > tol {tol:.3e}"


@pytest.mark.parametrize("variant,nerf", [("drt", dict(queries_per_ray=64)),
                                          ("drt", dict(queries_per_ray=16, activation="relu", jittering_enabled=False)),
                                          ("basic", dict(queries_per_ray=32, nerf_hide_emitters=True)),
                                          ("quadratic", dict(queries_per_ray=8))])
def test_fused_matches_oracle_on_the_fixture(uivr, oracle, gpu, variant, nerf):
    scene = uivr.cube_test_scene(32, 32, density_scale=2.0)
    scene.medium.sigma_t[1, 1, 1, 0] = 0.0                    # a voxel the relu / empty-space logic sees
    props = props_for(variant)
    nerf_props = dict(queries_per_ray=nerf.get("queries_per_ray", 128), activation=nerf.get("activation", "identity"),
                      jittering_enabled=nerf.get("jittering_enabled", True), hide_emitters=nerf.get("nerf_hide_emitters", False))
    spp, seed = 8, 4711
    osc = oracle.OracleScene(scene)
    Lr, cp = oracle.fused_render_primal(osc, props, nerf_props, spp, seed)
    rng = np.random.default_rng(1)
    n = Lr.shape[0]
    dL = ((rng.random((n, 6), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs, grgb, ca = oracle.fused_render_backward(osc, props, nerf_props, spp, seed, dL, Lr)

    sg = uivr.scene_to(scene, gpu)
    integ = _fused(uivr, variant, **nerf)
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    for counting in (True, False):                             # counting build and production (specialised) build
        h.enable_counters(counting)
        h.reset_counters()
        L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
        np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
        if counting:
            assert {k: int(v) for k, v in h.get_counters().items()} == cp
        h.reset_counters()
        grads = uivr.alloc_grads(sg, integ.param_keys)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
        if counting:
            assert {k: int(v) for k, v in h.get_counters().items()} == ca
        _close(grads[uivr.SIGMA_T_KEY], gs, f"{variant} grad sigma_t")
        _close(grads[uivr.ALBEDO_KEY], grgb, f"{variant} grad colour")
    h.enable_counters(False)


@pytest.mark.parametrize("flags", [128, 16384 | 524288], ids=["atomic-path", "record-memory-runs-out"])
def test_fused_backward_when_the_volpathsimple_half_leaves_the_deferred_path(uivr, oracle, gpu, flags):
    """The nerf half's window flush adds to the gradient grids on its own stream WHILE the volpathsimple half runs.  When that half
    takes the atomic gradient path (test hook 128; or record memory that runs out after the first ray sub-batch: 16384 | 524288) its
    last kernel is untile_gradients_kernel, whose flush must then be atomic too - a plain `+=` would lose window flushes that land
    between its load and its store.  Repeated: a lost update is a race, not a certainty."""
    scene = uivr.cube_test_scene(48, 48, density_scale=2.0)
    props, nerf_props = props_for("drt"), dict(queries_per_ray=64, activation="identity", jittering_enabled=True, hide_emitters=False)
    spp, seed = 16, 99
    osc = oracle.OracleScene(scene)
    Lr, _ = oracle.fused_render_primal(osc, props, nerf_props, spp, seed)
    n = Lr.shape[0]
    dL = ((np.random.default_rng(3).random((n, 6), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs, grgb, _ = oracle.fused_render_backward(osc, props, nerf_props, spp, seed, dL, Lr)
    sg = uivr.scene_to(scene, gpu)
    d = {"type": "nerf+volpathsimple", "test_hooks": True, "queries_per_ray": 64}
    d.update(props)
    integ = uivr.load_dict(d)
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    h.set_debug_flags(flags)
    try:
        for rep in range(4):
            L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
            np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
            grads = uivr.alloc_grads(sg, integ.param_keys)
            integ.sample(uivr.ADMode.Backward, sg, samp.clone(), batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
            _close(grads[uivr.SIGMA_T_KEY], gs, f"flags {flags} rep {rep} grad sigma_t")
            _close(grads[uivr.ALBEDO_KEY], grgb, f"flags {flags} rep {rep} grad colour")
    finally:
        h.set_debug_flags(0)


@pytest.mark.parametrize("env,factor", [(True, 0), (False, 3), (True, 3)])
def test_fused_with_envmap_and_supergrid_matches_oracle(uivr, oracle, gpu, env, factor):
    """the reference's nerf scenes run an environment map AND majorant_resolution_factor 8
    (python/scene_config.py:36,102-141) - the fused pass under either emitter and either kind of majorant
    (drt_fused_env.hip, drt_fused_super.hip, drt_fused_env_super.hip): radiance of both halves bit-exact, counters
    equal, gradients within tolerance, for the counting and the specialised kernels."""
    from test_gpu_envmap import _env_scene
    scene = _env_scene(uivr, film=24, factor=factor)
    scene.medium.emission = np.asarray(scene.medium.albedo).copy()      # one asset for both (scene_config.py:109-110)
    if not env:
        scene.emitter = uivr.cube_test_scene(4, 4).emitter
    props = props_for("drt")
    nerf = dict(queries_per_ray=24)
    nerf_props = dict(queries_per_ray=24, activation="identity", jittering_enabled=True, hide_emitters=False)
    spp, seed = 8, 515
    osc = oracle.OracleScene(scene)
    Lr, cp = oracle.fused_render_primal(osc, props, nerf_props, spp, seed)
    rng = np.random.default_rng(3)
    n = Lr.shape[0]
    dL = ((rng.random((n, 6), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs, grgb, ca = oracle.fused_render_backward(osc, props, nerf_props, spp, seed, dL, Lr)
    sg = uivr.scene_to(scene, gpu)
    integ = _fused(uivr, "drt", **nerf)
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    for counting in (True, False):
        h.enable_counters(counting)
        h.reset_counters()
        L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
        np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
        if counting:
            assert {k: int(v) for k, v in h.get_counters().items()} == cp
        h.reset_counters()
        grads = uivr.alloc_grads(sg, integ.param_keys)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
        if counting:
            assert {k: int(v) for k, v in h.get_counters().items()} == ca
        _close(grads[uivr.SIGMA_T_KEY], gs, "grad sigma_t")
        _close(grads[uivr.ALBEDO_KEY], grgb, "grad colour")
    h.enable_counters(False)
    # ... and equals the two stand-alone HIP integrators on the same scene
    drt = uivr.load_dict(dict(type="volpathsimple", **props))
    nrf = uivr.load_dict(dict(type="nerf", queries_per_ray=24))
    Ld, _, _ = drt.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    Ln, _, _ = nrf.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    assert torch.equal(L[:, :3], Ln) and torch.equal(L[:, 3:], Ld)


def test_fused_equals_the_two_standalone_integrators(uivr, gpu):
    from uivr_amd import synthetic
    sg = synthetic.smoke_scene(res=48, film=64, device=gpu, optical_side=12.0)
    sg.medium.albedo = (torch.rand_like(sg.medium.albedo) * 0.8 + 0.1).contiguous()
    sg.medium.emission = sg.medium.albedo
    spp, seed = 4, 99
    fused = _fused(uivr, "drt", queries_per_ray=48)
    drt = uivr.load_dict(dict(type="volpathsimple", **props_for("drt")))
    nerf = uivr.load_dict(dict(type="nerf", queries_per_ray=48))
    n = 64 * 64 * spp
    batch = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    L, _, st = fused.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    Ld, _, std = drt.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    Ln, _, stn = nerf.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    assert torch.equal(L[:, :3], Ln) and torch.equal(L[:, 3:], Ld)
    dL = (torch.rand((n, 6), device=gpu) - 0.5) * 1e-2
    g = uivr.alloc_grads(sg, fused.param_keys)
    fused.sample(uivr.ADMode.Backward, sg, samp, batch, δL=dL, state_in=st, grads=g)
    gd = uivr.alloc_grads(sg, drt.param_keys)
    drt.sample(uivr.ADMode.Backward, sg, samp, batch, δL=dL[:, 3:].contiguous(), state_in=std, grads=gd)
    gn = uivr.alloc_grads(sg, nerf.param_keys)
    nerf.sample(uivr.ADMode.Backward, sg, samp, batch, δL=dL[:, :3].contiguous(), state_in=stn, grads=gn)
    for key, ref in ((uivr.SIGMA_T_KEY, gd[uivr.SIGMA_T_KEY] + gn[uivr.SIGMA_T_KEY]), (uivr.ALBEDO_KEY, gd[uivr.ALBEDO_KEY] + gn[uivr.EMISSION_KEY])):
        tol = GRAD_RTOL * float(ref.abs().max())
        assert float((g[key] - ref).abs().max()) <= tol, key
    # the four-channel copy follows parameter updates (drt_params_changed / a new tensor)
    sg.medium.sigma_t.mul_(0.5)
    L2, _, _ = fused.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    Ld2, _, _ = drt.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    assert torch.equal(L2[:, 3:], Ld2) and not torch.equal(L2, L)
    # the autograd op: image [pixels, 6]
    params = {uivr.SIGMA_T_KEY: sg.medium.sigma_t.clone().requires_grad_(True), uivr.ALBEDO_KEY: sg.medium.albedo.clone().requires_grad_(True)}
    img = uivr.render(sg, params=params, integrator=fused, spp=4, spp_grad=2, seed=3, seed_grad=4)
    assert img.shape == (64 * 64, 6)
    ((img - 0.5) ** 2).mean().backward()
    assert float(params[uivr.SIGMA_T_KEY].grad.abs().max()) > 0 and float(params[uivr.ALBEDO_KEY].grad.abs().max()) > 0


def test_config5_fused_256_512x32_128_queries(uivr, oracle, gpu):
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=256, film=512, device=gpu)
    integ = uivr.get_int_config("nerf-drt-fused").create(max_depth=64)
    assert integ.queries_per_ray == 128
    props, nerf_props = props_for("drt"), dict(queries_per_ray=128)
    spp, seed = 32, 2007
    s = sg.sensors[0]
    batch = uivr.RayBatch(n_rays=512 * 512 * spp, spp=spp, sensor=s)
    samp = uivr.IndependentSampler(seed, spp)
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    L2, _, _ = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    assert torch.equal(L, L2) and torch.isfinite(L).all()
    gi = torch.randn((512 * 512, 6), device=gpu) * 1e-6
    g = []
    for scale in (1.0, 2.0):
        L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
        grads = uivr.alloc_grads(sg, integ.param_keys)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=integ.film_backward(sg, (scale * gi).contiguous(), spp), state_in=st, grads=grads)
        g.append(grads["_flat"])
    sc = float(g[0].abs().max())
    assert sc > 0 and torch.isfinite(g[0]).all() and float((g[1] - 2.0 * g[0]).abs().max()) <= 1e-3 * sc
    del g, grads
    # window of 128 pixels of row 380 against the oracle
    first, n = (380 * 512 + 200) * spp, 128 * spp
    m = sg.medium
    cpu = uivr.Scene(medium=uivr.GridMedium(sigma_t=m.sigma_t.cpu().numpy(), albedo=m.albedo.cpu().numpy(), bbox_min=m.bbox_min,
                                            bbox_max=m.bbox_max, scale=m.scale), emitter=sg.emitter, sensors=sg.sensors)
    osc = oracle.OracleScene(cpu)
    Lr, cp = oracle.fused_render_primal(osc, props, nerf_props, spp, seed, n_rays=n, ray_offset=first)
    np.testing.assert_array_equal(L[first:first + n].cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    rng = np.random.default_rng(8)
    dL = ((rng.random((n, 6), dtype=np.float32) - 0.5) * 1e-3).astype(np.float32)
    gs, grgb, ca = oracle.fused_render_backward(osc, props, nerf_props, spp, seed, dL, Lr, n_rays=n, ray_offset=first)
    wb = uivr.RayBatch(n_rays=n, spp=spp, sensor=s, ray_offset=first)
    h = integ.native_handle(sg)
    h.enable_counters(True)
    h.reset_counters()
    Lw, _, stw = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), wb)
    assert {k: int(v) for k, v in h.get_counters().items()} == cp
    np.testing.assert_array_equal(Lw.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    h.reset_counters()
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp, wb, δL=torch.from_numpy(dL).to(gpu), state_in=stw, grads=grads)
    assert {k: int(v) for k, v in h.get_counters().items()} == ca
    h.enable_counters(False)
    _close(grads[uivr.SIGMA_T_KEY], gs, "fused window grad sigma_t")
    _close(grads[uivr.ALBEDO_KEY], grgb, "fused window grad colour")


def test_nerf_tile_adjoint_and_fused_pass_repeat_at_size(gpu):
    """tools/stress_nerf_tile.py: config 5 (nerf and nerf + DRT, 256^3, 512^2 x 32 spp) and two ragged shapes, the same seeds over and over - the LDS
    window protocol of drt_nerf_tile.hip (rays waiting for the window, flushes, moves) and the two streams of the fused pass must not change a result:
    radiance bitwise the same every time, gradients equal up to summation order and finite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_nerf_tile.py"), "--reps", "6"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "STRESS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
