"""Finite-difference gradients of a rendering loss (python/fd.py) - the reference's yardstick for its integrators'
gradients (tests/test_integrators.py:261-347) and what the `fd-forward` integrator configuration stands for
(opt_config.py:57-59: `uses_fd`, `fd_epsilon`).  Every entry of every parameter grid is offset by `eps`, the scene is
rendered again with the SAME seed (so that the difference is not drowned in Monte Carlo noise) and the loss
difference is divided by `eps`.  On the device a 128^2 x 4096 spp rendering takes ~0.1 s, so the full 27 + 81 entry
protocol of the reference's test runs in seconds.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .image_io import write_image
from .render import render_primal
from .scene import ALBEDO_KEY, EMISSION_KEY, SIGMA_T_KEY, GridMedium, Scene


def _scene_with(scene: Scene, values: Dict[str, torch.Tensor]) -> Scene:
    m = scene.medium
    medium = GridMedium(sigma_t=values.get(SIGMA_T_KEY, m.sigma_t), albedo=values.get(ALBEDO_KEY, m.albedo),
                        bbox_min=m.bbox_min, bbox_max=m.bbox_max, scale=m.scale,
                        majorant_resolution_factor=m.majorant_resolution_factor, emission=values.get(EMISSION_KEY, m.emission))
    return Scene(medium=medium, emitter=scene.emitter, sensors=scene.sensors)


def fd_gradients(output_dir: Optional[str], scene: Scene, params: Dict[str, torch.Tensor], loss_fn: Callable, eps: float,
                 spp: int = 4096, write_images: bool = False, integrator=None, seed: int = 1234, sensor: int = 0,
                 central: bool = False) -> Dict[str, np.ndarray]:
    """python/fd.py:10-77.  `params`: {key: device grid (Z, Y, X, C)} - the entries to differentiate (they replace the
    scene's grids); `loss_fn(image)` with image (H, W, 3) -> scalar tensor.  Returns {key: array of d loss / d entry}.
    `central=True` uses (loss(+eps) - loss(-eps)) / (2 eps) instead of the reference's forward difference."""
    if integrator is None:
        raise ValueError("fd_gradients needs the integrator to render with")
    s = scene.sensors[sensor]

    def loss_of(values, fname=None):
        img = render_primal(_scene_with(scene, values), integrator, sensor, spp, seed).view(s.height, s.width, 3)
        if write_images and fname:
            write_image(os.path.join(output_dir, fname), img)
        return float(loss_fn(img))

    values = {k: v.detach().clone() for k, v in params.items()}
    loss_center = loss_of(values, 'fd_center.pfm')
    results = {}
    for run_i, k in enumerate(values):
        flat = values[k].view(-1)
        grads = np.full(tuple(values[k].shape), np.nan)
        for i in range(flat.numel()):
            idx = np.unravel_index(i, grads.shape)
            orig = float(flat[i])
            flat[i] = orig + eps
            lp = loss_of(values, f"fd_{run_i}_{'_'.join(str(j) for j in idx)}.pfm")
            if central:
                flat[i] = orig - eps
                grads[idx] = (lp - loss_of(values)) / (2.0 * eps)
            else:
                grads[idx] = (lp - loss_center) / eps
            flat[i] = orig                                                     # restore before moving on
        results[k] = grads
    return results
