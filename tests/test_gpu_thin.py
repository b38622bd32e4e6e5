"""Round 6: the primal kernels of the queued supergrid tracer finish a ray in the regeneration block when it is over before it begins (box
miss, flagged-empty pixel, a first flight whose optical-depth target exceeds largest majorant x segment), and launches in index order over a
medium the HOST has seen to be thin run the ROUNDS instantiation (drt_sq.hip; the hint is a majorant read-back nothing waits for, drt_capi.cpp).
Both against the oracle: radiance bit-exact, counters equal, the adjoint pass that follows (path cache entries written by the finished rays)
within 2e-4 - before the hint can have arrived and after a synchronisation has made sure it has."""
import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _thin_scene(uivr, sigma, film=(48, 40)):
    rng = np.random.default_rng(12)
    st = (rng.random((24, 24, 24, 1), dtype=np.float32) * sigma).astype(np.float32)
    st[:, :, 8:12] = 0.0
    al = (rng.random((24, 24, 24, 3), dtype=np.float32) * 0.8 + 0.1).astype(np.float32)
    medium = uivr.GridMedium(sigma_t=st, albedo=al, bbox_min=(-1.0, -1.0, -1.0), bbox_max=(1.0, 1.0, 1.0), scale=1.0, majorant_resolution_factor=4)
    sensor = uivr.PerspectiveSensor(origin=(0.5, 1.0, 5.0), target=(0.0, 0.0, 0.0), fov=32.0, width=film[0], height=film[1])
    return uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter((0.8, 1.0, 0.9)), sensors=[sensor])


@pytest.mark.parametrize("sigma", [0.05, 0.6, 6.0], ids=["thin", "borderline", "thick"])
def test_explicit_ray_launches_in_index_order_before_and_after_the_majorant_hint(uivr, oracle, gpu, sigma):
    """sigma 0.05: majorant x box diagonal = 0.17 - nearly every ray is over at once, the ROUNDS kernels once the hint is there; 0.6: 2.1, thin by the
    criterion with half of the first flights colliding; 6.0: thick, the plain kernels whatever the hint says."""
    scene = _thin_scene(uivr, sigma)
    rng = np.random.default_rng(4)
    n, spp, seed = 40000, 8, 8801
    o = (rng.normal(size=(n, 3)) * 0.2 + np.array([0.3, 0.6, 4.5])).astype(np.float32)
    tgt = (rng.random((n, 3)) * 2.6 - 1.3).astype(np.float32)              # some miss the box
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    props = props_for("drt")
    osc = oracle.OracleScene(scene, sensor_index=None)
    Lr, cp = oracle.render_primal(osc, props, spp, seed, rays_o=o, rays_d=d)
    dL = ((rng.random((n, 3), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs, ga, ca = oracle.render_backward(osc, props, spp, seed, dL, Lr, rays_o=o, rays_d=d)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n, spp=spp, o=torch.from_numpy(o).to(gpu), d=torch.from_numpy(d).to(gpu))
    for rep in range(3):
        for counting in (False, True):
            h.enable_counters(counting)
            h.reset_counters()
            samp = uivr.IndependentSampler(seed, spp)
            L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
            np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
            if counting:
                assert {k: int(v) for k, v in h.get_counters().items()} == cp
            h.reset_counters()
            grads = uivr.alloc_grads(sg)
            integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
            if counting:
                assert {k: int(v) for k, v in h.get_counters().items()} == ca
            for key, ref in ((uivr.SIGMA_T_KEY, gs), (uivr.ALBEDO_KEY, ga)):
                g = grads[key].double().cpu().numpy()
                assert np.abs(g - ref).max() <= GRAD_RTOL * np.abs(ref).max() + 1e-12, (sigma, rep, key)
        h.enable_counters(False)
        torch.cuda.synchronize()                                             # (the majorant read-back has landed: later launches know)


def test_sensor_launch_over_a_thin_medium_has_no_tail_launch_and_matches(uivr, oracle, gpu):
    scene = _thin_scene(uivr, 0.05, film=(96, 80))
    props = props_for("drt")
    spp, seed = 16, 8802
    ref = oracle.h1_step(oracle.OracleScene(scene), props, spp, seed)
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    n_pix = 96 * 80
    for rep in range(2):
        img = uivr.render_primal(sg, integ, 0, spp, seed)
        np.testing.assert_allclose(img.cpu().numpy(), ref["image"], rtol=0, atol=1e-6)
        L, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(seed, spp), uivr.RayBatch(n_rays=n_pix * spp, spp=spp, sensor=sg.sensors[0]))
        np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), ref["L"].view(np.uint32))
        grads = uivr.render_backward(sg, integ, ((2.0 / (n_pix * 3)) * (img - 0.5)).contiguous(), 0, spp, seed)
        for key, name in ((uivr.SIGMA_T_KEY, "grad_sigma_t"), (uivr.ALBEDO_KEY, "grad_albedo")):
            g = grads[key].double().cpu().numpy()
            assert np.abs(g - ref[name]).max() <= GRAD_RTOL * np.abs(ref[name]).max() + 1e-12, (rep, name)
        torch.cuda.synchronize()
