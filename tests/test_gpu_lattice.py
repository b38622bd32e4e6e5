"""Colour grids on their OWN lattice on the GPU (drt_set_colour_resolution, csrc/drt_own.hip) against the oracle: the reference's
janga-smoke pairs a 264 x 136 x 136 density with 256 x 128 x 128 albedo / emission grids (python/scene_config.py:108-110) and Mitsuba
interpolates every grid on its own resolution.  A 33 x 17 x 17 density with a 32 x 16 x 16 colour grid (the same ratio, an eighth of the
size): radiance BIT-EXACT, counters equal, gradients within 2e-4 max|oracle| - the colour gradient on the colour grid's lattice - for
volpathsimple (global majorant and supergrid, either emitter, sensor rays and explicit ray batches), nerf and the fused pass."""
import numpy as np
import pytest
import torch

from conftest import props_for
from test_oracle_envmap import _blob_map

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _close(g_hip, g_ref, what):
    g = g_hip.detach().double().cpu().numpy()
    assert g.shape == g_ref.shape, (what, g.shape, g_ref.shape)
    tol = GRAD_RTOL * np.abs(g_ref).max() + 1e-9
    assert np.abs(g_ref).max() > 0, what
    assert np.abs(g - g_ref).max() <= tol, f"{what}: {np.abs(g - g_ref).max():.3e} > {tol:.3e}"


def _scene(uivr, factor=0, env=False, film=(40, 28), colour=(16, 16, 32)):
    rng = np.random.default_rng(8)
    st = (rng.random((17, 17, 33, 1), dtype=np.float32) ** 2 * 5.0).astype(np.float32)
    st[:, :, 12:18] = 0.0                                                  # an empty slab (occupancy mask, empty supergrid cells)
    al = (rng.random(tuple(colour) + (3,), dtype=np.float32) * 0.85 + 0.1).astype(np.float32)
    medium = uivr.GridMedium(sigma_t=st, albedo=al, emission=al.copy(), bbox_min=(-1.94, -1.0, -1.0), bbox_max=(1.94, 1.0, 1.0),
                             scale=1.2, majorant_resolution_factor=factor)
    sensor = uivr.PerspectiveSensor(origin=(2.0, 1.5, 6.5), target=(0.0, 0.0, 0.0), fov=35.0, width=film[0], height=film[1])
    em = uivr.EnvmapEmitter(pixels=_blob_map(), scale=0.6, to_world=uivr.EnvmapEmitter.rotation_y(25.0)) if env else uivr.ConstantEmitter((0.9, 1.0, 0.6))
    return uivr.Scene(medium=medium, emitter=em, sensors=[sensor])


@pytest.mark.parametrize("variant,factor,env", [("drt", 0, False), ("drt", 4, False), ("basic", 0, True), ("quadratic", 4, True),
                                                ("drt-nomis", 4, False)])
def test_volpathsimple_with_the_albedo_on_its_own_lattice(uivr, oracle, gpu, variant, factor, env):
    props = props_for(variant)
    scene = _scene(uivr, factor, env)
    s = scene.sensors[0]
    spp, seed = 8, 6601
    osc = oracle.OracleScene(scene)
    ref = oracle.h1_step(osc, props, spp, seed)
    _, c_primal = oracle.render_primal(osc, props, spp, seed)
    assert ref["grad_albedo"].shape == (16, 16, 32, 3)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    h = integ.native_handle(sg)
    h.enable_counters(True)
    h.reset_counters()
    n_pix = s.width * s.height
    batch = uivr.RayBatch(n_rays=n_pix * spp, spp=spp, sensor=sg.sensors[0])
    L, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(seed, spp), batch)
    np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(ref["L"]))
    img = uivr.render_primal(sg, integ, 0, spp, seed)
    grads = uivr.render_backward(sg, integ, ((2.0 / (n_pix * 3)) * (img - 0.5)).contiguous(), 0, spp, seed)
    cnt = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    assert cnt == {k: ref["counters"][k] + 2 * c_primal[k] for k in ref["counters"]}
    assert tuple(grads[uivr.ALBEDO_KEY].shape) == (16, 16, 32, 3) and tuple(grads[uivr.SIGMA_T_KEY].shape) == (17, 17, 33, 1)
    _close(grads[uivr.SIGMA_T_KEY], ref["grad_sigma_t"], f"{variant} grad sigma_t")
    _close(grads[uivr.ALBEDO_KEY], ref["grad_albedo"], f"{variant} grad albedo")


def test_explicit_ray_batch_with_the_albedo_on_its_own_lattice(uivr, oracle, gpu):
    scene = _scene(uivr, factor=4)
    rng = np.random.default_rng(2)
    n, spp, seed = 3000, 4, 6602
    o = (rng.normal(size=(n, 3)) * 0.3 + np.array([0.5, 0.8, 5.0])).astype(np.float32)
    tgt = (rng.random((n, 3)) * np.array([3.88, 2.0, 2.0]) + np.array([-1.94, -1.0, -1.0])).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    props = props_for("drt")
    osc = oracle.OracleScene(scene, sensor_index=None)
    Lr, _ = oracle.render_primal(osc, props, spp, seed, rays_o=o, rays_d=d)
    dL = ((rng.random((n, 3), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs, ga, _ = oracle.render_backward(osc, props, spp, seed, dL, Lr, rays_o=o, rays_d=d)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    batch = uivr.RayBatch(n_rays=n, spp=spp, o=torch.from_numpy(o).to(gpu), d=torch.from_numpy(d).to(gpu))
    samp = uivr.IndependentSampler(seed, spp)
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(Lr))
    grads = uivr.alloc_grads(sg)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
    _close(grads[uivr.SIGMA_T_KEY], gs, "grad sigma_t")
    _close(grads[uivr.ALBEDO_KEY], ga, "grad albedo")


@pytest.mark.parametrize("props", [dict(queries_per_ray=48), dict(queries_per_ray=24, activation="relu", jittering_enabled=False)])
def test_nerf_with_the_emission_on_its_own_lattice(uivr, oracle, gpu, props):
    scene = _scene(uivr)
    s = scene.sensors[0]
    spp, seed = 4, 6603
    n_pix = s.width * s.height
    osc = oracle.OracleScene(scene)
    Lr, cr = oracle.nerf_render(osc, scene.medium.emission, props, spp, seed)
    dL = np.repeat((2.0 / (n_pix * 3)) * (oracle.develop(Lr, spp) - 0.5) / spp, spp, axis=0).astype(np.float32)
    gs, ge, ca = oracle.nerf_render(osc, scene.medium.emission, props, spp, seed, dL=dL, L_in=Lr)
    assert ge.shape == (16, 16, 32, 3)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="nerf", **props))
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n_pix * spp, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    h.enable_counters(True)
    h.reset_counters()
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(Lr))
    assert {k: int(v) for k, v in h.get_counters().items()} == cr
    h.reset_counters()
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
    assert {k: int(v) for k, v in h.get_counters().items()} == ca
    h.enable_counters(False)
    _close(grads[uivr.SIGMA_T_KEY], gs, "nerf grad sigma_t")
    _close(grads[uivr.EMISSION_KEY], ge, "nerf grad emission")


def test_fused_pass_with_the_colour_grid_on_its_own_lattice(uivr, oracle, gpu):
    scene = _scene(uivr, factor=4, env=True)
    props = props_for("drt")
    nerf_props = dict(queries_per_ray=32, activation="identity", jittering_enabled=True, hide_emitters=False)
    spp, seed = 4, 6604
    osc = oracle.OracleScene(scene)
    Lr, cp = oracle.fused_render_primal(osc, props, nerf_props, spp, seed)
    n = Lr.shape[0]
    dL = ((np.random.default_rng(4).random((n, 6), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs, grgb, ca = oracle.fused_render_backward(osc, props, nerf_props, spp, seed, dL, Lr)
    sg = uivr.scene_to(scene, gpu)
    d = {"type": "nerf+volpathsimple", "queries_per_ray": 32}
    d.update(props)
    integ = uivr.load_dict(d)
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    h.enable_counters(True)
    h.reset_counters()
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(Lr))
    assert {k: int(v) for k, v in h.get_counters().items()} == cp
    h.reset_counters()
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
    assert {k: int(v) for k, v in h.get_counters().items()} == ca
    h.enable_counters(False)
    _close(grads[uivr.SIGMA_T_KEY], gs, "fused grad sigma_t")
    _close(grads[uivr.ALBEDO_KEY], grgb, "fused grad colour")


def test_equal_lattices_keep_the_production_kernels(uivr, oracle, gpu):
    """A colour grid on sigma_t's lattice never reaches drt_own.hip: same handle, lattices switched own -> equal -> own; each result the
    oracle's."""
    props = props_for("drt")
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    spp, seed = 4, 6605
    for colour in ((16, 16, 32), (17, 17, 33), (16, 16, 32)):
        scene = _scene(uivr, factor=4, colour=colour)
        ref = oracle.h1_step(oracle.OracleScene(scene), props, spp, seed)
        sg = uivr.scene_to(scene, gpu)
        s = sg.sensors[0]
        img = uivr.render_primal(sg, integ, 0, spp, seed)
        np.testing.assert_allclose(img.cpu().numpy(), ref["image"], rtol=0, atol=1e-6)
        grads = uivr.render_backward(sg, integ, ((2.0 / (s.width * s.height * 3)) * (img - 0.5)).contiguous(), 0, spp, seed)
        _close(grads[uivr.ALBEDO_KEY], ref["grad_albedo"], f"colour {colour} grad albedo")
        _close(grads[uivr.SIGMA_T_KEY], ref["grad_sigma_t"], f"colour {colour} grad sigma_t")
