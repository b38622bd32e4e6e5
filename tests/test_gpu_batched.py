"""N1: the batched multi-sensor render op (python/batched.py) and the losses (python/losses.py)."""
import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu


def _scene(uivr, n_sensors=5, film=24):
    from uivr_amd import synthetic
    scene = uivr.cube_test_scene(film, film, density_scale=2.0)
    scene.sensors = synthetic.ring_sensors(n_sensors, radius=6.0, height=2.0, target=(0.5, 0.5, 0.5), fov=30.0,
                                           width=film, film_height=film)
    return scene


def test_batch_sampling_matches_oracle(uivr, oracle, gpu):
    """sample_batch_pixels + sample_batch_rays (batched.py:397-467), bit for bit."""
    scene = _scene(uivr)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props_for("drt")))
    table = uivr.sensors_to_device(sg.sensors, gpu)
    B, spp, seed = 257, 3, 4242
    for which in (1, 2):
        ro, rd, sidx, pix = uivr.sample_batch(integ, sg, table, B, spp, seed, which)
        sub0 = uivr.sample_tea_32(seed, 5)[0]
        subk = uivr.sample_tea_32(seed, 17 * which + 5)[0]
        ro_r, rd_r, si_r, px_r = oracle.batch_sample_rays(scene.sensors, B, spp, sub0, subk)
        np.testing.assert_array_equal(sidx.cpu().numpy().astype(np.uint32), si_r)
        np.testing.assert_array_equal(pix.cpu().numpy().astype(np.uint32), px_r)
        np.testing.assert_array_equal(ro.cpu().numpy().view(np.uint32), ro_r.view(np.uint32))
        np.testing.assert_array_equal(rd.cpu().numpy().view(np.uint32), rd_r.view(np.uint32))
    assert 0 <= int(sidx.min()) and int(sidx.max()) < 5 and int(pix.max()) < 24
    assert len(torch.unique(sidx)) == 5           # every sensor is hit


def test_render_batch_forward_backward(uivr, oracle, gpu):
    """render_batch == oracle H1 over the same explicit rays (primal seed / adjoint seed_grad)."""
    scene = _scene(uivr)
    sg = uivr.scene_to(scene, gpu)
    props = props_for("drt")
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    B, spp, spp_grad, seed, seed_grad = 300, 4, 2, 100, 200
    params = {k: v.clone().requires_grad_(True) for k, v in sg.params().items() if k in integ.param_keys}
    image, _, _, sidx, pix = uivr.render_batch(B, sg, params=params, integrator=integ, seed=seed, seed_grad=seed_grad,
                                               spp=spp, spp_grad=spp_grad)
    assert image.shape == (B, 3)
    ref = torch.rand((5, 24, 24, 3), device=gpu)
    ref_values = uivr.gather_ref_values(ref, sidx, pix)
    loss = uivr.losses.l1(image, ref_values)                       # optimize.py:349
    loss.backward()

    osc = oracle.OracleScene(scene, sensor_index=None)
    ro, rd, _, _ = oracle.batch_sample_rays(scene.sensors, B, spp, uivr.sample_tea_32(seed, 5)[0], uivr.sample_tea_32(seed, 22)[0])
    L, _ = oracle.render_primal(osc, props, spp, seed, rays_o=ro, rays_d=rd)
    np.testing.assert_allclose(image.detach().cpu().numpy(), oracle.develop(L, spp), rtol=0, atol=1e-6)
    # adjoint: rays of sampler 2 at spp_grad, seed_grad; dL = sign(image - ref) / numel / spp_grad
    ro2, rd2, _, _ = oracle.batch_sample_rays(scene.sensors, B, spp_grad, uivr.sample_tea_32(seed, 5)[0], uivr.sample_tea_32(seed, 39)[0])
    L2, _ = oracle.render_primal(osc, props, spp_grad, seed_grad, rays_o=ro2, rays_d=rd2)
    g_img = (torch.sign(image.detach() - ref_values) / image.numel()).cpu().numpy()
    dL = np.repeat(g_img / spp_grad, spp_grad, axis=0).astype(np.float32)
    gs, ga, _ = oracle.render_backward(osc, props, spp_grad, seed_grad, dL, L2, rays_o=ro2, rays_d=rd2)
    for key, g in ((uivr.SIGMA_T_KEY, gs), (uivr.ALBEDO_KEY, ga)):
        err = np.abs(params[key].grad.double().cpu().numpy() - g).max()
        assert err <= 2e-4 * np.abs(g).max() + 1e-9, key
    with pytest.raises(Exception, match="seed"):
        uivr.render_batch(B, sg, integrator=integ, seed=7, seed_grad=7, spp=1)


def test_losses(uivr, gpu):
    a = torch.rand((64, 3), device=gpu)
    b = torch.rand((64, 3), device=gpu)
    L = uivr.losses
    assert float(L.l1(a, b)) == pytest.approx(float((a - b).abs().mean()), rel=1e-6)
    assert float(L.l2(a, b)) == pytest.approx(float(((a - b) ** 2).mean()), rel=1e-6)
    assert float(L.root_mean_squared_error(a, b)) == pytest.approx(float(((a - b) ** 2).mean().sqrt()), rel=1e-6)
    assert float(L.psnr(a, b)) == pytest.approx(-10 * np.log10(float(((a - b) ** 2).mean())), rel=1e-5)
    assert float(L.mean_relative_absolute_error(a, b)) > 0 and float(L.huber(a, b)) > 0


@pytest.mark.parametrize("factor", [0, 6])
def test_render_batch_logical_shards_equal_unsharded(uivr, gpu, factor):
    """Sharded render_batch without a process group (SURVEY.md 8e "G logical shards on one device"): the ranks'
    shares, rendered one after the other, tile the unsharded batch bit for bit; gradient shares sum to the
    unsharded gradient.  Global majorant (cooperative kernels) and supergrid (state machine + per-lane kernels)."""
    from uivr_amd import synthetic
    scene = synthetic.smoke_scene(res=24, film=32, device=gpu, optical_side=10.0)
    scene.sensors = synthetic.ring_sensors(5, radius=5.0, height=0.8, fov=30.0, width=32, film_height=32)
    scene.medium.majorant_resolution_factor = factor
    integ = uivr.get_int_config("volpathsimple-drt").create(max_depth=32)
    B, spp, spp_grad, seed, seed_grad = 515, 4, 2, 21, 22

    def run(shard):
        params = {k: v.clone().requires_grad_(True) for k, v in scene.params().items() if k in integ.param_keys}
        image, _, _, sidx, pix = uivr.render_batch(B, scene, params=params, integrator=integ, seed=seed, seed_grad=seed_grad,
                                                   spp=spp, spp_grad=spp_grad, shard=shard)
        n_local = image.shape[0]
        (uivr.losses.l1(image, torch.full_like(image, 0.3)) * uivr.local_loss_scale(n_local, B)).backward()
        return image.detach(), sidx, pix, torch.cat([params[k].grad.reshape(-1) for k in integ.param_keys])

    img_u, sidx_u, pix_u, g_u = run(None)
    for world in (2, 3):
        acc = torch.zeros_like(g_u)
        for rank in range(world):
            sh = uivr.ShardSpec(rank, world)
            first, count = sh.batch_range(B)
            img, sidx, pix, g = run(sh)
            assert torch.equal(img, img_u[first:first + count]), (world, rank)
            assert torch.equal(sidx, sidx_u[first:first + count]) and torch.equal(pix, pix_u[first:first + count])
            acc += g
        tol = 2e-4 * float(g_u.abs().max())
        assert float((acc - g_u).abs().max()) <= tol, (world, float((acc - g_u).abs().max()), tol)
