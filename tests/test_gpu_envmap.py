"""GPU parity of the `envmap` emitter (volpathsimple.py:273,284,419): device
primitives bit for bit against the oracle, then the integrators under an environment map -
primal radiance BIT-EXACT, counters equal, gradients within 2e-4 * max|g| - on both tracing
kernels, and the nerf background composite (nerf.py:131-146)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import props_for
from test_oracle_envmap import _blob_map, _sphere_dirs

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _env_scene(uivr, film=32, factor=0, **kw):
    scene = uivr.cube_test_scene(film, film, density_scale=2.0)
    if factor:                                              # 3^3 -> 9^3 voxels with a 3^3 majorant supergrid
        scene.medium.sigma_t = np.repeat(np.repeat(np.repeat(np.asarray(scene.medium.sigma_t), 3, 0), 3, 1), 3, 2).copy()
        scene.medium.albedo = np.repeat(np.repeat(np.repeat(np.asarray(scene.medium.albedo), 3, 0), 3, 1), 3, 2).copy()
        scene.medium.majorant_resolution_factor = factor
    scene.emitter = uivr.EnvmapEmitter(pixels=_blob_map(**kw), scale=0.5, to_world=uivr.EnvmapEmitter.rotation_y(-40.0))
    return scene


def _debug(uivr, gpu, sg, op, inp):
    integ = uivr.load_dict(dict(type="volpathsimple", **props_for("drt")))
    h = integ.native_handle(sg)
    n = inp.shape[0]
    buf = np.zeros((n, 6), dtype=np.float32)
    buf[:, :inp.shape[1]] = inp
    tin = torch.from_numpy(buf).to(gpu)
    tout = torch.empty_like(tin)
    h.debug_eval(op, tin.data_ptr(), n, tout.data_ptr())
    torch.cuda.synchronize()
    return tout.cpu().numpy()


def test_envmap_primitives_bit_exact(uivr, oracle, gpu):
    scene = _env_scene(uivr)
    sg = uivr.scene_to(scene, gpu)
    L = oracle.lib()
    rng = np.random.default_rng(5)
    yx = rng.normal(size=(20000, 2)).astype(np.float32)
    yx = np.concatenate([yx, np.float32([[0, 0], [0, 1], [0, -1], [1, 0], [-1, 0], [1, 1], [-1, -1], [1e-30, 1], [1, 1e-30]])])
    got = _debug(uivr, gpu, sg, 11, yx)[:, 0]
    ref = np.array([L.drto_atan2f(float(a), float(b)) for a, b in yx], dtype=np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(ref))

    dirs = np.concatenate([_sphere_dirs(4000, 6), np.float32([[0, 1, 0], [0, -1, 0], [0, 0, -1], [1, 0, 0]])])
    got = _debug(uivr, gpu, sg, 12, dirs)[:, :4]
    ref = np.array([np.append(oracle.envmap_eval(scene.emitter, d), np.float32(oracle.envmap_pdf(scene.emitter, d)))
                    for d in dirs], dtype=np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(ref))

    uv = np.concatenate([rng.random((4000, 2), dtype=np.float32), np.float32([[0, 0], [1 - 2.0 ** -24, 1 - 2.0 ** -24], [0.5, 0]])])
    got = _debug(uivr, gpu, sg, 13, uv)[:, :4]
    ref = np.zeros_like(got)
    for i, (a, b) in enumerate(uv):
        d, p, _ = oracle.envmap_sample(scene.emitter, a, b)
        ref[i, :3], ref[i, 3] = d, p
    np.testing.assert_array_equal(_bits(got), _bits(ref))


@pytest.mark.parametrize("flags,variant", [(0, "drt"), (0, "basic"), (0, "quadratic"), (8, "drt"), (32, "drt"), (32, "basic"),
                                           (-3, "drt"), (-3, "basic"), (-3, "quadratic"), (-3 - 1073741824, "drt"), (-3 - 1073741824, "quadratic")])
def test_envmap_render_matches_oracle(uivr, oracle, gpu, flags, variant):
    """(flags -3: a majorant supergrid of factor 3 - the envmap instantiations of the queued supergrid tracer, drt_sq.hip; -3 - f: with test
    hooks f - 1073741824: a launch this small scheduled like the large ones, i.e. with the adjoint's tail pool and tail launch)"""
    props = props_for(variant)
    scene = _env_scene(uivr, factor=3 if flags < 0 else 0)
    flags = -flags - 3 if flags < 0 else flags
    spp, seed = 8, 4242
    ref = oracle.h1_step(oracle.OracleScene(scene), props, spp, seed)
    _, c_primal = oracle.render_primal(oracle.OracleScene(scene), props, spp, seed)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", test_hooks=flags != 0, **props))
    h = integ.native_handle(sg)
    h.set_debug_flags(flags)
    h.enable_counters(True)
    h.reset_counters()
    batch = uivr.RayBatch(n_rays=32 * 32 * spp, spp=spp, sensor=sg.sensors[0])
    L, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(seed, spp), batch)
    np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(ref["L"]))
    img = uivr.render_primal(sg, integ, 0, spp, seed)
    grads = uivr.render_backward(sg, integ, ((2.0 / (32 * 32 * 3)) * (img - 0.5)).contiguous(), 0, spp, seed)
    cnt = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    h.set_debug_flags(0)
    assert cnt == {k: ref["counters"][k] + 2 * c_primal[k] for k in ref["counters"]}
    for key, name in ((uivr.SIGMA_T_KEY, "grad_sigma_t"), (uivr.ALBEDO_KEY, "grad_albedo")):
        g = grads[key].cpu().numpy().astype(np.float64)
        tol = GRAD_RTOL * np.abs(ref[name]).max() + 1e-9
        assert np.abs(g - ref[name]).max() <= tol, (variant, name)


def test_envmap_hide_emitters_and_switching(uivr, oracle, gpu):
    """hide_emitters (volpathsimple.py:268) and emitter replacement on a live handle."""
    scene = _env_scene(uivr, film=16)
    sg = uivr.scene_to(scene, gpu)
    spp, seed = 8, 99
    for hide in (False, True):
        props = props_for("drt", hide_emitters=hide)
        integ = uivr.load_dict(dict(type="volpathsimple", **props))
        L, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(seed, spp),
                               uivr.RayBatch(n_rays=16 * 16 * spp, spp=spp, sensor=sg.sensors[0]))
        Lr, _ = oracle.render_primal(oracle.OracleScene(scene), props, spp, seed)
        np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(Lr))
    # same integrator object, emitter switched envmap -> constant -> another envmap
    integ = uivr.load_dict(dict(type="volpathsimple", **props_for("drt")))
    a = uivr.render_primal(sg, integ, 0, spp, seed).cpu().numpy()
    const_scene = uivr.Scene(medium=sg.medium, emitter=uivr.ConstantEmitter((1.0, 0.8, 0.2)), sensors=sg.sensors)
    b = uivr.render_primal(const_scene, integ, 0, spp, seed).cpu().numpy()
    cube = uivr.cube_test_scene(16, 16, density_scale=2.0)
    cube.emitter = uivr.ConstantEmitter((1.0, 0.8, 0.2))
    Lr, _ = oracle.render_primal(oracle.OracleScene(cube), props_for("drt"), spp, seed)
    np.testing.assert_allclose(b, oracle.develop(Lr, spp), rtol=0, atol=1e-6)
    a2 = uivr.render_primal(sg, integ, 0, spp, seed).cpu().numpy()
    np.testing.assert_array_equal(a, a2)
    assert np.abs(a - b).max() > 1e-3
    with pytest.raises(TypeError):
        bad = uivr.Scene(medium=sg.medium, emitter=uivr.EnvmapEmitter(pixels=_blob_map()), sensors=sg.sensors)
        uivr.render_primal(bad, integ, 0, 1, 1)            # host pixels: no CPU path


def test_envmap_nerf_background(uivr, oracle, gpu):
    scene = uivr.cube_test_scene(32, 32, density_scale=1.5)
    scene.emitter = uivr.EnvmapEmitter(pixels=_blob_map(), scale=0.5, to_world=uivr.EnvmapEmitter.rotation_y(15.0))
    spp, seed = 4, 1234
    props = dict(queries_per_ray=32)
    Lr, _ = oracle.nerf_render(oracle.OracleScene(scene), scene.medium.emission, props, spp, seed)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="nerf", **props))
    L, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(seed, spp),
                           uivr.RayBatch(n_rays=32 * 32 * spp, spp=spp, sensor=sg.sensors[0]))
    np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(Lr))


def test_envmap_full_size_map(uivr, gpu):
    """A 2k x 1k map (the paper scenes' size): upload + table build, white furnace under a
    constant-valued map at the BASELINE grid size stays L == Le."""
    from uivr_amd import synthetic
    scene = synthetic.constant_cube_scene(res=64, film=64, device=gpu, sigma_t=3.0)
    scene.medium.albedo.fill_(1.0)
    rgb = torch.tensor([0.7, 0.9, 1.1], device=gpu)
    scene.emitter = uivr.EnvmapEmitter(pixels=rgb.expand(1024, 2048, 3).contiguous())
    integ = uivr.get_int_config("volpathsimple-drt").create(max_depth=64)
    img = uivr.render_primal(scene, integ, 0, 64, 3)
    ratio = (img / rgb).cpu().numpy()
    assert abs(ratio.mean() - 1.0) < 0.01 and np.abs(ratio - 1).max() < 0.5
