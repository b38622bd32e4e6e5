"""python/fd.py on the device (`uivr_amd.fd_gradients`): same protocol (every entry offset by eps, same seed, forward
difference), checked against the definition and against the integrator's own gradients."""
import os

import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu


def test_fd_gradients_protocol_and_agreement_with_ad(uivr, gpu, tmp_path):
    scene = uivr.scene_to(uivr.cube_test_scene(64, 64, density_scale=2.0), gpu)        # tests/test_integrators.py:19-116
    integ = uivr.load_dict(dict({"type": "volpathsimple"}, **props_for("quadratic-nomis")))
    loss = lambda img: ((img - 0.5) ** 2).mean()                                       # tests/test_integrators.py:119
    eps, spp = 5e-3, 2048
    params = {uivr.SIGMA_T_KEY: scene.medium.sigma_t}
    before = scene.medium.sigma_t.clone()
    out = str(tmp_path)
    fd = uivr.fd_gradients(out, scene, params, loss, eps, spp=spp, integrator=integ, seed=1234, write_images=True)
    assert list(fd) == [uivr.SIGMA_T_KEY] and fd[uivr.SIGMA_T_KEY].shape == (3, 3, 3, 1) and np.isfinite(fd[uivr.SIGMA_T_KEY]).all()
    assert torch.equal(scene.medium.sigma_t, before)                                   # the scene's grids are left alone
    files = sorted(os.listdir(out))
    assert "fd_center.pfm" in files and "fd_0_1_2_0_0.pfm" in files and len(files) == 28
    # definition: one entry by hand, same seed
    def loss_at(delta):
        st = before.clone(); st[1, 2, 0, 0] += delta
        sc = uivr.Scene(medium=uivr.GridMedium(sigma_t=st, albedo=scene.medium.albedo, bbox_min=scene.medium.bbox_min,
                                               bbox_max=scene.medium.bbox_max, scale=scene.medium.scale), emitter=scene.emitter,
                        sensors=scene.sensors)
        return float(loss(uivr.render_primal(sc, integ, 0, spp, 1234).view(64, 64, 3)))
    l0, lp, lm = loss_at(0.0), loss_at(eps), loss_at(-eps)
    assert fd[uivr.SIGMA_T_KEY][1, 2, 0, 0] == pytest.approx((lp - l0) / eps, rel=1e-9)
    np.testing.assert_array_equal(uivr.read_image(os.path.join(out, "fd_center.pfm")),
                                  uivr.render_primal(scene, integ, 0, spp, 1234).view(64, 64, 3).cpu().numpy())
    fdc = uivr.fd_gradients(None, scene, params, loss, eps, spp=spp, integrator=integ, seed=1234, central=True)
    assert fdc[uivr.SIGMA_T_KEY][1, 2, 0, 0] == pytest.approx((lp - lm) / (2 * eps), rel=1e-9)
    # ... and the integrator's gradient (mean of 8 runs at 512 spp) is what the differences measure
    runs = []
    for r in range(8):
        img = uivr.render_primal(scene, integ, 0, 512, 100 + r)
        g = uivr.render_backward(scene, integ, ((2.0 / img.numel()) * (img - 0.5)).contiguous(), 0, 512, 100 + r)
        runs.append(g[uivr.SIGMA_T_KEY].reshape(-1).double().cpu().numpy())
    ad = np.mean(runs, axis=0)
    f = fdc[uivr.SIGMA_T_KEY].reshape(-1)
    assert np.corrcoef(ad, f)[0, 1] > 0.98
    assert np.linalg.norm(ad - f) < 0.15 * np.linalg.norm(f), (ad, f)
    with pytest.raises(ValueError):
        uivr.fd_gradients(None, scene, params, loss, eps)
