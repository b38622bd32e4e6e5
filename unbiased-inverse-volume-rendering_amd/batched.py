"""Batched (ray-centric) rendering: `render_batch`, the alternative to the sensor-centric `render`
used by the optimisation loop (reference: python/batched.py).

A batch is `batch_size` (sensor, pixel) pairs drawn over all sensors; the forward pass traces
`spp` rays through each pixel, the backward pass traces a separate, decorrelated set of `spp_grad`
rays through the SAME pixels (batched.py:69-82) with `seed_grad`.  Pixel / ray sampling runs on the
device (`drt_batch_sample_rays`), keyed exactly like the reference's three `independent` samplers
(sub-seed `tea32(seed, 17*i + 5)[0]`, wavefront sizes B, B*spp, B*spp_grad; batched.py:397-423).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .distributed import ShardSpec, allreduce_gradients
from .integrators import ADMode, IndependentSampler, RayBatch, sample_tea_32
from .render import _grid, _with_params, alloc_grads, sharded_support
from .scene import PerspectiveSensor, Scene


def sensors_to_device(sensors: Sequence[PerspectiveSensor], device) -> torch.Tensor:
    """[n_sensors, 16] float32 table {origin, left, up, dir, tan_x, tan_y, width, height} - the device-side
    counterpart of `dr.gather(mi.SensorPtr, scene.sensors_dr(), ...)` (optimize.py:295-296)."""
    rows = []
    w0, h0 = sensors[0].width, sensors[0].height
    for s in sensors:
        if (s.width, s.height) != (w0, h0):
            raise ValueError("all sensors must have the same film size (batched.py:427)")
        f = s.frame()
        rows.append(np.concatenate([f["origin"], f["left"], f["up"], f["dir"],
                                    [f["tan_x"], f["tan_y"], s.width, s.height]]).astype(np.float32))
    return torch.from_numpy(np.stack(rows)).to(device)


def sample_batch(integrator, scene: Scene, sensor_table: torch.Tensor, batch_size: int, spp: int, seed: int,
                 which: int, batch_first: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """sample_batch_pixels + sample_batch_rays (batched.py:397-467) -> rays_o, rays_d, sensor_idx, pixels.
    `which` = 1 for the primal rays, 2 for the adjoint rays (batch_samplers[which]).
    `batch_first` / `batch_size`: the range of GLOBAL batch entries to generate (one rank's share of a
    sharded batch; the samplers' lanes are the global entry / ray indices)."""
    h, dev = integrator._bind(scene)
    n = batch_size * spp
    ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
    sidx = torch.empty((batch_size,), dtype=torch.int32, device=dev)
    pix = torch.empty((batch_size, 2), dtype=torch.int32, device=dev)
    sub0 = sample_tea_32(seed, 17 * 0 + 5)[0]
    subk = sample_tea_32(seed, 17 * which + 5)[0]
    h.batch_sample_rays(sensor_table.data_ptr(), int(sensor_table.shape[0]), int(batch_size), int(spp), sub0, subk,
                        ro.data_ptr(), rd.data_ptr(), sidx.data_ptr(), pix.data_ptr(), int(batch_first))
    return ro, rd, sidx, pix


class _BatchedRenderOp(torch.autograd.Function):
    """python/batched.py:13-85.  With a `shard` (world > 1) the batch entries [first, first + count) of this
    rank are rendered - random streams keyed by the GLOBAL entry / ray index, so the union over ranks is
    the unsharded batch bit for bit - and the backward pass sums the gradient grids over the ranks with
    ONE all-reduce.  Unsharded calls never communicate."""

    @staticmethod
    def forward(ctx, p0, p1, scene, integrator, sensor_table, batch_size, spp, spp_grad, seed, seed_grad, shard):
        sc = _with_params(scene, integrator.param_keys, (p0.detach(), p1.detach()))
        first, count = shard.batch_range(batch_size)
        ro, rd, sidx, pix = sample_batch(integrator, sc, sensor_table, count, spp, seed, 1, first)
        batch = RayBatch(n_rays=count * spp, spp=spp, o=ro, d=rd, ray_offset=first * spp)
        L, _, _ = integrator.sample(ADMode.Primal, sc, IndependentSampler(seed, spp), batch)    # :163-173
        image = integrator.develop(sc, L, spp)                                                   # :176-197
        ctx.scene, ctx.integrator, ctx.sensor_table = sc, integrator, sensor_table
        ctx.range, ctx.spp_grad, ctx.seed, ctx.seed_grad, ctx.shard = (first, count), spp_grad, seed, seed_grad, shard
        ctx.mark_non_differentiable(sidx, pix)
        return image, sidx, pix

    @staticmethod
    def backward(ctx, grad_image, _gs, _gp):
        sc, integ = ctx.scene, ctx.integrator
        first, count = ctx.range
        # decorrelated rays through the same pixels (:69-82); same pixel sampler 0 => same pixels
        ro, rd, _, _ = sample_batch(integ, sc, ctx.sensor_table, count, ctx.spp_grad, ctx.seed, 2, first)
        batch = RayBatch(n_rays=count * ctx.spp_grad, spp=ctx.spp_grad, o=ro, d=rd, ray_offset=first * ctx.spp_grad)
        sampler = IndependentSampler(ctx.seed_grad, ctx.spp_grad)
        L, _, state = integ.sample(ADMode.Primal, sc, sampler.clone(), batch)                    # :255-264
        dL = integ.film_backward(sc, grad_image.contiguous(), ctx.spp_grad)                      # :272-306
        grads = alloc_grads(sc, integ.param_keys)
        support = sharded_support(sc, grads, ctx.shard)
        integ.sample(ADMode.Backward, sc, sampler, batch, δL=dL, state_in=state, grads=grads)    # :309-318
        allreduce_gradients(grads, shard=ctx.shard, support=support)
        k0, k1 = integ.param_keys
        return (grads[k0], grads[k1]) + (None,) * 9


def render_batch(batch_size: int, scene: Scene, sensors=None, film_size=None,
                 params: Optional[Dict[str, torch.Tensor]] = None, integrator=None, film=None,
                 pixel_format=None, sampler=None, seed: int = 0, seed_grad: int = 0, spp: int = 0,
                 spp_grad: int = 0, sensor_table: Optional[torch.Tensor] = None, shard: Optional[ShardSpec] = None):
    """Batched (ray-centric) alternative to `render` (python/batched.py:88-131).
    -> (image [batch_size, 3], film, sampler, sensor_idx [batch_size], pixel_idx [batch_size, 2])
    (`film` / `sampler` are returned as given: the device film is stateless here).
    `shard` (world > 1): the outputs hold this rank's entries `shard.batch_range(batch_size)` only and the
    backward pass all-reduces the gradient grids; scale the local loss by `local_loss_scale`."""
    if integrator is None:
        raise ValueError("render_batch: an integrator is required")
    if spp <= 0:
        raise ValueError("render_batch: spp must be > 0")
    if spp_grad == 0:
        spp_grad = spp
    if seed_grad == 0:
        seed_grad = sample_tea_32(seed, 1)[0]                                   # :117-119
    elif seed_grad == seed:
        raise Exception('The primal and differential seed should be different '
                        'to ensure unbiased gradient computation!')
    sensors = list(sensors) if sensors is not None else list(scene.sensors)
    if film_size is not None and tuple(film_size) != (sensors[0].width, sensors[0].height):
        raise ValueError("film_size does not match the sensors' film")
    keys = integrator.param_keys
    if params is None:
        params = {k: _grid(scene, k) for k in keys}
    for k in keys:
        if not isinstance(params[k], torch.Tensor):
            raise TypeError(f"render_batch: params['{k}'] must be a torch device tensor")
    if sensor_table is None:
        sensor_table = sensors_to_device(sensors, params[keys[0]].device)
    image, sidx, pix = _BatchedRenderOp.apply(params[keys[0]], params[keys[1]], scene, integrator, sensor_table,
                                              int(batch_size), int(spp), int(spp_grad), int(seed), int(seed_grad),
                                              shard or ShardSpec())
    return image, film, sampler, sidx, pix


def gather_ref_values(ref_images: torch.Tensor, sensor_idx: torch.Tensor, pixel_idx: torch.Tensor) -> torch.Tensor:
    """python/optimize.py:90-107: ref_images (n_sensors, H, W, C) -> [batch, C] at (sensor, y, x)."""
    if ref_images.dim() != 4 or ref_images.shape[-1] not in (3, 4):
        raise ValueError("ref_images must have shape (n_sensors, H, W, 3|4)")
    return ref_images[sensor_idx.long(), pixel_idx[:, 1].long(), pixel_idx[:, 0].long()]
