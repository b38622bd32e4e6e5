"""N2: the optimisation loop (python/optimize.py:275-365) end to end on a small synthetic target -
BASELINE config 3's shape (multi-sensor, batched rays, DRT, Adam, x2 upsampling, clipping)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _target_scene(uivr, gpu, res=16, film=32, n_sensors=4):
    from uivr_amd import synthetic
    scene = synthetic.smoke_scene(res=res, film=film, device=gpu, optical_side=8.0)
    scene.sensors = synthetic.ring_sensors(n_sensors, radius=5.0, height=0.8, fov=30.0, width=film, film_height=film)
    return scene


@pytest.mark.parametrize("batched", [True, False])
def test_run_optimization_reduces_loss(uivr, gpu, batched, tmp_path):
    scene = _target_scene(uivr, gpu)
    sc = uivr.SceneConfig(name="smoke16", scene=scene, param_keys=[uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY],
                          sensors=list(range(4)), start_from_value={uivr.SIGMA_T_KEY: 0.4, uivr.ALBEDO_KEY: 0.6},
                          max_depth=16, ref_spp=2048, max_density=20.0)
    oc = uivr.OptimizationConfig("test", spp=4, n_iter=40, lr=5e-2, primal_spp_factor=4,
                                 batch_size=1024 if batched else None, lr_schedule=uivr.Schedule.Last25,
                                 upsample=[0.5], checkpoint_stride=20)
    out = str(tmp_path)
    _, params, opt, hist = uivr.run_optimization(out, oc, sc, "volpathsimple-drt")
    assert len(hist) == 40 and np.isfinite(hist).all()
    assert np.mean(hist[-8:]) < 0.8 * np.mean(hist[:8]), (hist[:8], hist[-8:])
    # started at res/2 and was upsampled once at iteration 20 (optimize.py:146-155, 228-252)
    assert tuple(params[uivr.SIGMA_T_KEY].shape) == (16, 16, 16, 1) and tuple(params[uivr.ALBEDO_KEY].shape) == (16, 16, 16, 3)
    st, al = params[uivr.SIGMA_T_KEY], params[uivr.ALBEDO_KEY]
    assert float(st.min()) >= 0 and float(st.max()) <= 20.0 and float(al.min()) >= 0 and float(al.max()) <= 1
    # the optimised density correlates with the target
    tgt = scene.medium.sigma_t.flatten()
    corr = torch.corrcoef(torch.stack([st.flatten(), tgt]))[0, 1]
    assert float(corr) > 0.2
    files = sorted(os.listdir(os.path.join(out, "params")))
    assert "initial-medium1_sigma_t.vol" in files and "final-medium1_albedo.vol" in files and "00000020-medium1_sigma_t.vol" in files
    d, _, _ = uivr.read_vol(os.path.join(out, "params", "final-medium1_sigma_t.vol"))
    np.testing.assert_array_equal(d, st.cpu().numpy())
    assert opt.state[uivr.SIGMA_T_KEY][0] == 20          # Adam state restarted at the upsampling step


def test_multiresolution_run_keeps_the_fixed_albedo_on_its_own_lattice(uivr, gpu):
    """Only sigma_t is optimised, on an 8^3 grid that is upsampled once; the scene's 16^3 albedo is NOT a parameter: the reference upsamples what
    it optimises and leaves the other grids alone (python/optimize.py:228-252) - Mitsuba interpolates each grid on its own lattice.  Round 6: so does
    this build (drt_set_colour_resolution / drt_own.hip for the first half of the run, the production kernels after the upsampling step)."""
    scene = _target_scene(uivr, gpu)
    sc = uivr.SceneConfig(name="smoke16", scene=scene, param_keys=[uivr.SIGMA_T_KEY], sensors=list(range(4)),
                          start_from_value={uivr.SIGMA_T_KEY: 0.4}, max_depth=16, ref_spp=1024, max_density=20.0)
    oc = uivr.OptimizationConfig("own", spp=4, n_iter=30, lr=5e-2, primal_spp_factor=4, batch_size=1024, upsample=[0.5])
    seen = []
    final_scene, params, _, hist = uivr.run_optimization(None, oc, sc, "volpathsimple-drt",
                                                         progress=lambda i, l: seen.append(i))
    assert len(hist) == 30 and np.isfinite(hist).all() and seen == list(range(30))
    assert tuple(params[uivr.SIGMA_T_KEY].shape) == (16, 16, 16, 1) and list(params) == [uivr.SIGMA_T_KEY]
    assert final_scene.medium.albedo is scene.medium.albedo                  # the scene's grid as it is: never resampled
    assert np.mean(hist[-6:]) < np.mean(hist[:6])
    # one step on the coarse grid against the oracle: sigma_t 8^3, albedo 16^3
    from oracle import binding as ob
    from conftest import props_for
    m = scene.medium
    coarse = uivr.Scene(medium=uivr.GridMedium(sigma_t=torch.full((8, 8, 8, 1), 0.4, device=gpu), albedo=m.albedo, bbox_min=m.bbox_min,
                                               bbox_max=m.bbox_max, scale=m.scale), emitter=scene.emitter, sensors=scene.sensors)
    integ = uivr.get_int_config("volpathsimple-drt").create(max_depth=16)
    img = uivr.render_primal(coarse, integ, 0, 8, 77)
    cpu = uivr.Scene(medium=uivr.GridMedium(sigma_t=np.full((8, 8, 8, 1), 0.4, np.float32), albedo=m.albedo.cpu().numpy(), bbox_min=m.bbox_min,
                                            bbox_max=m.bbox_max, scale=m.scale), emitter=scene.emitter, sensors=scene.sensors)
    Lr, _ = ob.render_primal(ob.OracleScene(cpu), props_for("drt", max_depth=16), 8, 77)
    np.testing.assert_allclose(img.cpu().numpy(), ob.develop(Lr, 8), rtol=0, atol=1e-6)


def test_reference_cache_previews_and_checkpoints(uivr, gpu, tmp_path):
    """N3 around the loop (python/optimize.py:24-131, 255-272, 318-324, 355-362): reference renderings cached on
    disk (only the missing ones are rendered; multi-pass mean with seeds 1234 + pass), previews at the start, at
    `preview_stride` and at the end, the references of the preview sensors next to them, `.vol` checkpoints."""
    from uivr_amd import optimize as O
    scene = _target_scene(uivr, gpu, n_sensors=3)
    refs, out = str(tmp_path / "refs"), str(tmp_path / "out")
    sc = uivr.SceneConfig(name="smoke16", scene=scene, param_keys=[uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY], sensors=[0, 2],
                          start_from_value={uivr.SIGMA_T_KEY: 0.4, uivr.ALBEDO_KEY: 0.6}, max_depth=16, ref_spp=64,
                          max_density=20.0, references=refs, preview_sensors=[2])
    paths = uivr.get_reference_image_paths(sc)
    assert sorted(os.listdir(refs)) == ["ref_000000.pfm", "ref_000002.pfm"] and list(paths) == [0, 2]
    # what is in the file is the rendering (float32, lossless)
    integ = uivr.get_int_config(sc.ref_integrator).create(max_depth=16)
    want = uivr.render_primal(scene, integ, 2, 64, 1234).view(32, 32, 3)
    np.testing.assert_array_equal(uivr.read_image(paths[2]), want.cpu().numpy())
    # cached: nothing is rendered again (a sentinel survives), `overwrite` renders again
    sentinel = np.full((32, 32, 3), 0.25, np.float32)
    uivr.write_image(paths[0], sentinel)
    uivr.get_reference_image_paths(sc)
    np.testing.assert_array_equal(uivr.read_image(paths[0]), sentinel)
    batch = uivr.load_reference_images(paths, batchify=True, device=gpu)
    assert tuple(batch.shape) == (2, 32, 32, 3) and float(batch[0].mean()) == 0.25
    single = uivr.load_reference_images(paths)
    assert sorted(single) == [0, 2] and torch.equal(single[2].to(gpu), batch[1])
    uivr.get_reference_image_paths(sc, overwrite=True)
    assert not np.array_equal(uivr.read_image(paths[0]), sentinel)
    # passes: 32*32*64 rays with at most 32*32*24 per pass -> 3 passes of 22 spp, seeds 1234..1236, averaged
    got = uivr.render_reference_image(sc, {0: None}, max_rays_per_pass=32 * 32 * 24)[0]
    mean = sum(uivr.render_primal(scene, integ, 0, 22, 1234 + i) / 3 for i in range(3)).view(32, 32, 3)
    assert torch.allclose(got, mean, rtol=0, atol=1e-6)
    # the loop reads the cached references and writes previews + checkpoints
    oc = uivr.OptimizationConfig("t", spp=2, n_iter=5, lr=2e-2, primal_spp_factor=2, batch_size=256, preview_stride=2,
                                 preview_spp=8, checkpoint_stride=3)
    scene_o, params, _, hist = uivr.run_optimization(out, oc, sc, "volpathsimple-drt")
    assert len(hist) == 5
    names = sorted(f for f in os.listdir(out) if f.endswith(".pfm"))
    assert names == ["opt_00000002_0002.pfm", "opt_00000004_0002.pfm", "opt_final_0002.pfm", "opt_init_0002.pfm", "ref_0002.pfm"]
    np.testing.assert_array_equal(uivr.read_image(os.path.join(out, "ref_0002.pfm")), uivr.read_image(paths[2]))
    final = uivr.read_image(os.path.join(out, "opt_final_0002.pfm"))
    full = uivr.Scene(medium=scene_o.medium, emitter=scene_o.emitter, sensors=scene.sensors)
    opt_integ = uivr.get_int_config("volpathsimple-drt").create(max_depth=16)
    np.testing.assert_array_equal(final, uivr.render_primal(full, opt_integ, 2, 8, 1234).view(32, 32, 3).cpu().numpy())
    init = uivr.read_image(os.path.join(out, "opt_init_0002.pfm"))
    assert np.isfinite(init).all() and not np.array_equal(init, final)
    assert sorted(os.listdir(os.path.join(out, "params"))) == [
        "00000003-medium1_albedo.vol", "00000003-medium1_sigma_t.vol", "final-medium1_albedo.vol", "final-medium1_sigma_t.vol",
        "initial-medium1_albedo.vol", "initial-medium1_sigma_t.vol"]
    # previews can be switched off (opt_config.py:26-27)
    oc2 = uivr.OptimizationConfig("t", spp=2, n_iter=2, lr=2e-2, primal_spp_factor=2, batch_size=256, render_initial=False,
                                  render_final=False, checkpoint_initial=False, checkpoint_final=False)
    out2 = str(tmp_path / "out2")
    uivr.run_optimization(out2, oc2, sc, "volpathsimple-drt")
    assert sorted(os.listdir(out2)) == ["params", "ref_0002.pfm"] and os.listdir(os.path.join(out2, "params")) == []


def test_warm_start_from_nerf_checkpoint(uivr, gpu, tmp_path):
    """The reference's `*-from-nerf` flow (python/scene_config.py:123-141, python/reproduce.py:66-77): optimise with the
    `nerf` integrator, write `.vol` checkpoints, build the next scene's medium from those files, keep its sigma_t
    (`start_from_value` None) and restart the albedo from a constant, optimise with `volpathsimple-drt`."""
    scene = _target_scene(uivr, gpu, n_sensors=3)
    scene.medium.emission = (scene.medium.albedo * 0.5).contiguous()
    refs = torch.stack([uivr.render_primal(scene, uivr.get_int_config("volpathsimple-drt").create(max_depth=16), s, 256, 1234).view(32, 32, 3)
                        for s in range(3)])
    keys3 = [uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY, uivr.EMISSION_KEY]
    sc1 = uivr.SceneConfig(name="stage1", scene=scene, param_keys=keys3, sensors=[0, 1, 2],
                           start_from_value={uivr.SIGMA_T_KEY: 0.4, uivr.ALBEDO_KEY: 0.6, uivr.EMISSION_KEY: 0.1}, max_depth=16,
                           max_density=20.0)
    oc1 = uivr.OptimizationConfig("nerf", spp=4, n_iter=6, lr=1e-2, primal_spp_factor=1, batch_size=512, render_initial=False,
                                  render_final=False)
    out1 = str(tmp_path / "stage1" / "nerf")
    _, p1, _, _ = uivr.run_optimization(out1, oc1, sc1, "nerf", ref_images=refs)
    pdir = os.path.join(out1, "params")
    medium = uivr.medium_from_vol(os.path.join(pdir, "final-medium1_sigma_t.vol"), os.path.join(pdir, "final-medium1_albedo.vol"),
                                  os.path.join(pdir, "final-medium1_emission.vol"), scale=scene.medium.scale,
                                  majorant_resolution_factor=8, device=gpu)
    assert torch.equal(medium.sigma_t, p1[uivr.SIGMA_T_KEY]) and torch.equal(medium.emission, p1[uivr.EMISSION_KEY])
    assert tuple(medium.bbox_min) == tuple(scene.medium.bbox_min)
    scene2 = uivr.Scene(medium=medium, emitter=scene.emitter, sensors=scene.sensors)
    sc2 = uivr.SceneConfig(name="stage2", scene=scene2, param_keys=keys3, sensors=[0, 1, 2],
                           start_from_value={uivr.SIGMA_T_KEY: None, uivr.ALBEDO_KEY: 0.6, uivr.EMISSION_KEY: None}, max_depth=16,
                           max_density=20.0)
    seen = {}
    oc2 = uivr.OptimizationConfig("drt", spp=4, n_iter=3, lr=1e-3, primal_spp_factor=2, batch_size=512, render_initial=False,
                                  render_final=False)
    out2 = str(tmp_path / "stage2")
    _, p2, _, hist = uivr.run_optimization(out2, oc2, sc2, "volpathsimple-drt", ref_images=refs)
    assert len(hist) == 3 and np.isfinite(hist).all()
    start, _, _ = uivr.read_vol(os.path.join(out2, "params", "initial-medium1_sigma_t.vol"))
    np.testing.assert_array_equal(start, p1[uivr.SIGMA_T_KEY].cpu().numpy())           # kept from the checkpoint
    a0, _, _ = uivr.read_vol(os.path.join(out2, "params", "initial-medium1_albedo.vol"))
    assert float(a0.min()) == float(a0.max()) == np.float32(0.6)                        # restarted from the constant
    # the emission is carried along untouched (volpathsimple does not read it: no gradient), sigma_t moved
    assert torch.equal(p2[uivr.EMISSION_KEY], p1[uivr.EMISSION_KEY])
    assert not torch.equal(p2[uivr.SIGMA_T_KEY], p1[uivr.SIGMA_T_KEY])
