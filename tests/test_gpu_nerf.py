"""GPU parity tests of the `nerf` integrator (python/integrators/nerf.py) - BASELINE config 5's
emissive RGB + sigma path.  Same bar as the DRT path: primal radiance bit-exact against the
oracle, gradients within 2e-4 * max|g|; plus the reference's own gradient check
(tests/test_integrators.py:158-218: radiative-backprop gradients vs forward finite differences,
eps = 5e-3, at most 3 entries per channel outside rtol = 3e-2, all within rtol = 0.75)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _close(g_hip, g_ref, what):
    g_hip = g_hip.detach().cpu().numpy().astype(np.float64)
    tol = GRAD_RTOL * np.abs(g_ref).max() + 1e-9
    err = np.abs(g_hip - g_ref).max()
    assert err <= tol, f"{what}: max abs err {err:.3e} > tol {tol:.3e}"


def _h1(uivr, sg, integ, spp, seed):
    n = sg.sensors[0].width * sg.sensors[0].height
    img = uivr.render_primal(sg, integ, 0, spp, seed)
    grads = uivr.render_backward(sg, integ, ((2.0 / (n * 3)) * (img - 0.5)).contiguous(), 0, spp, seed)
    return img, grads


@pytest.mark.parametrize("flags", [0, 512, 512 | 128, 512 | 256])
@pytest.mark.parametrize("props", [dict(), dict(queries_per_ray=64, activation="relu"),
                                   dict(queries_per_ray=17, jittering_enabled=False, hide_emitters=True)])
def test_nerf_matches_oracle(uivr, oracle, gpu, props, flags):
    """flags: 0 = production: the adjoint of sensor rays pre-reduces its splats in an LDS window (drt_nerf_tile.hip, round 5); 512 = the record
    path that explicit ray batches take (nerf_kernel + deferred tile-binned splatting), with 128 = atomics into the apron scratch,
    256 = two-chunk record streams (out-of-chunks fallback)."""
    scene = uivr.cube_test_scene(32, 32, density_scale=1.5)
    if props.get("activation") == "relu":
        scene.medium.sigma_t[1, 1, 1, 0] = -0.3          # exercise the clamped branch
    spp, seed = 4, 1234
    osc = oracle.OracleScene(scene)
    Lr, cr = oracle.nerf_render(osc, scene.medium.emission, props, spp, seed)
    img_r = oracle.develop(Lr, spp)
    dL = np.repeat((2.0 / (32 * 32 * 3)) * (img_r - 0.5) / spp, spp, axis=0).astype(np.float32)
    gs, ge, _ = oracle.nerf_render(osc, scene.medium.emission, props, spp, seed, dL=dL, L_in=Lr)

    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="nerf", test_hooks=flags != 0, **props))      # flags: the flavour with test hooks
    assert isinstance(integ, uivr.NeRFIntegrator)
    batch = uivr.RayBatch(n_rays=32 * 32 * spp, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    h = integ.native_handle(sg)
    h.set_debug_flags(flags)
    h.enable_counters(True)
    h.reset_counters()
    L, _, state = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    cnt = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    assert cnt == cr
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=state, grads=grads)
    h.set_debug_flags(0)
    _close(grads[uivr.SIGMA_T_KEY], gs, "grad sigma_t")
    _close(grads[uivr.EMISSION_KEY], ge, "grad emission")


def test_nerf_rb_gradients_vs_finite_differences(uivr, gpu):
    """tests/test_integrators.py:158-218 on the HIP path (same seed on both sides, fd.py:12,45)."""
    scene = uivr.cube_test_scene(32, 32, density_scale=1.0)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.get_int_config("nerf").create(max_depth=64)
    spp, seed, eps = 4, 1234, 5e-3

    def loss():
        img = uivr.render_primal(sg, integ, 0, spp, seed)
        return float(((img.double() - 0.5) ** 2).mean())

    _, grads = _h1(uivr, sg, integ, spp, seed)
    l0 = loss()
    for key, grid in ((uivr.SIGMA_T_KEY, sg.medium.sigma_t), (uivr.EMISSION_KEY, sg.medium.emission)):
        fd = torch.zeros_like(grid, dtype=torch.float64)
        flat = grid.view(-1)
        for i in range(flat.numel()):
            orig = float(flat[i])
            flat[i] = orig + eps                       # in-place: bumps the tensor version -> params_changed
            fd.view(-1)[i] = (loss() - l0) / eps
            flat[i] = orig
        a = grads[key].double().cpu().numpy()
        b = fd.cpu().numpy()
        for c in range(a.shape[-1]):
            bad = np.sum(np.abs(a[..., c] - b[..., c]) >= 3e-2 * np.abs(b[..., c]))
            assert bad <= 3, (key, c, bad)
            # atol: O(eps) curvature of the forward difference of the quadratic loss + fp32 noise floor
            assert np.allclose(a[..., c], b[..., c], rtol=0.75, atol=1e-5), (key, c)


def test_nerf_autograd_and_errors(uivr, gpu):
    scene = uivr.cube_test_scene(16, 16)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict({"type": "nerf", "queries_per_ray": 32})
    params = {uivr.SIGMA_T_KEY: sg.medium.sigma_t.clone().requires_grad_(True),
              uivr.EMISSION_KEY: sg.medium.emission.clone().requires_grad_(True)}
    img = uivr.render(sg, params=params, integrator=integ, spp=4, seed=3)
    ((img - 0.5) ** 2).mean().backward()
    assert params[uivr.SIGMA_T_KEY].grad.abs().max() > 0 and params[uivr.EMISSION_KEY].grad.abs().max() > 0
    with pytest.raises(ValueError):
        uivr.load_dict({"type": "nerf", "activation": "tanh"})                # nerf.py:44
    with pytest.raises(NotImplementedError):
        uivr.load_dict({"type": "nerf", "density_noise_std": 0.1})
    no_em = uivr.Scene(medium=uivr.GridMedium(sigma_t=sg.medium.sigma_t, albedo=None, bbox_min=sg.medium.bbox_min,
                                              bbox_max=sg.medium.bbox_max), emitter=sg.emitter, sensors=sg.sensors)
    with pytest.raises(TypeError):
        uivr.render_primal(no_em, integ, 0, 1, 1)


@pytest.mark.parametrize("film,chunk,spp,props", [((33, 26), 33, 5, dict(queries_per_ray=40)), ((16, 40), 40, 19, dict(queries_per_ray=24, activation="relu")),
                                                  ((24, 24), 48, 1, dict())])
def test_nerf_tile_adjoint_equals_the_record_path(uivr, oracle, gpu, film, chunk, spp, props):
    """drt_nerf_tile.hip against the ORACLE (round 6: directly, not only through the record path) and against nerf_kernel + drt_deferred.hip
    (test hook 512) where the tile mapping is ragged: films that are no multiple of the 8 x 8 tile and not square, spp that is no multiple of a
    workgroup's 16 samples, a launch over a window of the film (ray_offset: most tiles hold none of its rays) and the interleaved chunks of a
    sharded render (ShardSpec).  Same statements per ray: gradients up to summation order."""
    rng = np.random.default_rng(11)
    st = (rng.random((20, 18, 22, 1), dtype=np.float32) * 3.0).astype(np.float32)
    st[rng.random(st.shape) < 0.5] = 0.0
    if props.get("activation") == "relu":
        st[3:6, 3:6, 3:6] = -0.4
    elif spp == 1:
        st[8:11, 8:11, 8:11] = -0.8          # identity activation with negative densities: a > 1, growing throughput (the fixed-point bound follows)
    em = (rng.random((20, 18, 22, 3), dtype=np.float32) * 0.9).astype(np.float32)
    medium = uivr.GridMedium(sigma_t=st, albedo=em.copy(), emission=em, bbox_min=(-1, -0.9, -1.1), bbox_max=(1, 0.9, 1.1), scale=1.3)
    sensor = uivr.PerspectiveSensor(origin=(2.5, 1.5, 3.0), target=(0, 0, 0), fov=40.0, width=film[0], height=film[1])
    sg = uivr.scene_to(uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter((0.3, 0.4, 0.5)), sensors=[sensor]), gpu)
    n_pix, seed = film[0] * film[1], 99
    out = {}
    for name, flags in (("tile", 0), ("records", 512)):
        integ = uivr.load_dict(dict(type="nerf", test_hooks=True, **props))
        integ.native_handle(sg).set_debug_flags(flags)
        cases = {}
        # the whole film
        img = uivr.render_primal(sg, integ, 0, spp, seed)
        gi = ((2.0 / (n_pix * 3)) * (img - 0.4)).contiguous()
        cases["film"] = uivr.render_backward(sg, integ, gi, 0, spp, seed)["_flat"].clone()
        # the two shards of a sharded render, interleaved chunks of pixels
        for rank in range(2):
            sh = uivr.ShardSpec(rank, 2, chunk_pixels=chunk)
            li = uivr.render_primal(sg, integ, 0, spp, seed, shard=sh)
            cases[f"shard{rank}"] = uivr.render_backward(sg, integ, gi[sh.pixel_indices(n_pix, gpu)].contiguous(), 0, spp, seed, shard=sh)["_flat"].clone()
            assert torch.equal(li, img[sh.pixel_indices(n_pix, gpu)])
        # a window of rays in the middle of the film
        first, n = (n_pix // 3) * spp + 2, (n_pix // 4) * spp
        wb = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0], ray_offset=first)
        samp = uivr.IndependentSampler(seed, spp)
        L, _, stt = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), wb)
        dL = ((torch.arange(n * 3, device=gpu, dtype=torch.float32).reshape(n, 3) % 7) - 3.0) * 1e-3
        grads = uivr.alloc_grads(sg, integ.param_keys)
        integ.sample(uivr.ADMode.Backward, sg, samp, wb, δL=dL, state_in=stt, grads=grads)
        cases["window"] = grads["_flat"].clone()
        integ.native_handle(sg).set_debug_flags(0)
        out[name] = cases
    for k in out["tile"]:
        a, b = out["tile"][k], out["records"][k]
        assert float(b.abs().max()) > 0
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), k
    # ... and each case against oracle.nerf_render (python/integrators/nerf.py:47-148 restated): radiance bit-exact, gradients within 2e-4 max
    osc = oracle.OracleScene(uivr.Scene(medium=uivr.GridMedium(sigma_t=st, albedo=em.copy(), emission=em, bbox_min=(-1, -0.9, -1.1),
                                                                bbox_max=(1, 0.9, 1.1), scale=1.3),
                                        emitter=uivr.ConstantEmitter((0.3, 0.4, 0.5)), sensors=[sensor]))
    Lr, _ = oracle.nerf_render(osc, em, props, spp, seed)
    integ = uivr.load_dict(dict(type="nerf", **props))
    L_hip, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(seed, spp), uivr.RayBatch(n_rays=n_pix * spp, spp=spp, sensor=sg.sensors[0]))
    np.testing.assert_array_equal(L_hip.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    gi_r = (2.0 / (n_pix * 3)) * (oracle.develop(Lr, spp) - 0.4)
    dL_film = np.repeat(gi_r / spp, spp, axis=0).astype(np.float32)

    def oracle_grads(first, n, dL_rays):
        gs, ge, _ = oracle.nerf_render(osc, em, props, spp, seed, dL=dL_rays, L_in=Lr[first:first + n], n_rays=n, ray_offset=first)
        return np.concatenate([gs.reshape(-1), ge.reshape(-1)])

    ref = {"film": oracle_grads(0, n_pix * spp, dL_film)}
    for rank in range(2):                                                # rank r of 2 takes the pixel chunks c with c % 2 == r
        acc = np.zeros_like(ref["film"])
        for c in range(rank, (n_pix + chunk - 1) // chunk, 2):
            p0, p1 = c * chunk, min((c + 1) * chunk, n_pix)
            acc += oracle_grads(p0 * spp, (p1 - p0) * spp, dL_film[p0 * spp:p1 * spp])
        ref[f"shard{rank}"] = acc
    first, n = (n_pix // 3) * spp + 2, (n_pix // 4) * spp
    ref["window"] = oracle_grads(first, n, (((np.arange(n * 3, dtype=np.float32).reshape(n, 3) % 7) - 3.0) * 1e-3).astype(np.float32))
    for k, r in ref.items():
        a = out["tile"][k].double().cpu().numpy()
        assert np.abs(r).max() > 0
        assert np.abs(a - r).max() <= 2e-4 * np.abs(r).max() + 1e-12, ("tile kernel vs oracle", k)
    total = out["tile"]["shard0"] + out["tile"]["shard1"]
    assert float((total - out["tile"]["film"]).abs().max()) <= 2e-5 * float(total.abs().max())


def test_nerf_tile_adjoint_propagates_non_finite_inputs(uivr, gpu):
    """The LDS window accumulates in fixed point, which cannot carry a NaN: a non-finite dL (a diverged optimisation) must not come out as a
    finite, wrong gradient - the pass marks the gradient grids NaN, as the record path's float sums would be."""
    scene = uivr.cube_test_scene(16, 16, density_scale=1.5)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.get_int_config("nerf").create(max_depth=64)
    spp, seed = 4, 5
    batch = uivr.RayBatch(n_rays=16 * 16 * spp, spp=spp, sensor=sg.sensors[0])
    samp = uivr.IndependentSampler(seed, spp)
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    dL = torch.full_like(L, 1e-3)
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp.clone(), batch, δL=dL, state_in=st, grads=grads)
    assert bool(torch.isfinite(grads["_flat"]).all()) and float(grads["_flat"].abs().max()) > 0
    dL[37, 1] = float("nan")
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp.clone(), batch, δL=dL, state_in=st, grads=grads)
    # EVERY voxel: a caller (or a masked all-reduce) that looks at part of the grids must not find plausible finite numbers there
    assert bool(torch.isnan(grads[uivr.SIGMA_T_KEY]).all()) and bool(torch.isnan(grads[uivr.EMISSION_KEY]).all())
    # ... and a dL that is finite but overflows the bound the fixed-point units follow from
    dL = torch.full_like(L, 1e-3)
    dL[5, 0] = 3.0e38
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp.clone(), batch, δL=dL, state_in=st, grads=grads)
    assert not bool(torch.isfinite(grads[uivr.SIGMA_T_KEY]).any()) and not bool(torch.isfinite(grads[uivr.EMISSION_KEY]).any())
