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
