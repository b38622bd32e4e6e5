"""Differentiable render ops: the callers of the hot path.

`render(...)` mirrors `mi.render(scene, params, integrator, sensor, spp, spp_grad,
seed, seed_grad)` as the reference uses it (python/optimize.py:44,129,345,
python/fd.py:12,45): forward = primal pass at (seed, spp); backward = the
radiative-backprop sequence of `render_batch_backward` (python/batched.py:212-326)
at (seed_grad, spp_grad):
    (1) sample(Primal) with a clone of the sampler          -> L
    (2) film: image = mean_spp L ; dL = grad_image[pixel]/spp (box filter)
    (3) sample(Backward, same sampler, dL, state_in = L)    -> gradients
exposed to PyTorch through a `torch.autograd.Function` so that
`loss.backward()` / an optimizer work like `dr.backward(loss)` / `opt.step()`.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .distributed import ShardSpec, allreduce_gradients, gradient_support
from .integrators import ADMode, IndependentSampler, RayBatch, sample_tea_32
from .scene import ALBEDO_KEY, EMISSION_KEY, SIGMA_T_KEY, GridMedium, Scene


def _grid(scene: Scene, key: str):
    return {SIGMA_T_KEY: scene.medium.sigma_t, ALBEDO_KEY: scene.medium.albedo,
            EMISSION_KEY: scene.medium.emission}[key]


def alloc_grads(scene: Scene, keys=(SIGMA_T_KEY, ALBEDO_KEY)) -> Dict[str, torch.Tensor]:
    """Zeroed gradient grids shaped like the parameters `keys` (an integrator's `param_keys`),
    carved out of ONE flat buffer (`_flat`) so that the multi-GPU all-reduce is a single
    collective.  (A fresh buffer per backward: autograd hands these tensors out as `.grad`, so they cannot
    be pooled; the cost is one caching-allocator hit plus a 256 MiB memset = 0.05 ms at 256^3, and the
    benchmark's step pays the same memset.)"""
    grids = [_grid(scene, k) for k in keys]
    # every grid starts at a multiple of 4 floats: its view is 16-byte aligned whatever the voxel counts (the fused Adam
    # step and the block-mask kernel read float4s); the up to 3 padding floats per grid stay zero
    pad = lambda n: (n + 3) // 4 * 4
    flat = torch.zeros(sum(pad(g.numel()) for g in grids), dtype=torch.float32, device=grids[0].device)
    out, off = {"_flat": flat}, 0
    for k, g in zip(keys, grids):
        out[k] = flat[off:off + g.numel()].view(g.shape)
        off += pad(g.numel())
    return out


def sharded_support(scene: Scene, grads: Dict[str, torch.Tensor], shard: Optional[ShardSpec]):
    """The blocks of the flat gradient buffer that can be non-zero, from this step's (replicated) sigma_t - computed
    BEFORE the adjoint pass is enqueued, so that the gradient all-reduce behind it packs without a mask collective and
    without a host wait (distributed.gradient_support).  None for unsharded work (no collective) and for integrators
    other than volpathsimple's (sigma_t, albedo) pair."""
    import torch.distributed as dist
    if shard is None or not shard.partitioned or not (dist.is_available() and dist.is_initialized()):
        return None
    if SIGMA_T_KEY not in grads or ALBEDO_KEY not in grads:
        return None
    return gradient_support(scene.medium.sigma_t, grads, sparse_keys=(ALBEDO_KEY,))


def _with_params(scene: Scene, keys, tensors) -> Scene:
    m = scene.medium
    vals = {SIGMA_T_KEY: m.sigma_t, ALBEDO_KEY: m.albedo, EMISSION_KEY: m.emission}
    vals.update(dict(zip(keys, tensors)))
    medium = GridMedium(sigma_t=vals[SIGMA_T_KEY], albedo=vals[ALBEDO_KEY], bbox_min=m.bbox_min, bbox_max=m.bbox_max,
                        scale=m.scale, majorant_resolution_factor=m.majorant_resolution_factor,
                        emission=vals[EMISSION_KEY])
    return Scene(medium=medium, emitter=scene.emitter, sensors=scene.sensors)


def _sensor_batch(scene: Scene, sensor_index: int, spp: int, shard: Optional[ShardSpec]) -> RayBatch:
    sensor = scene.sensors[sensor_index]
    n_pixels = sensor.width * sensor.height
    shard = shard or ShardSpec()
    n_local = shard.n_local_pixels(n_pixels)
    off, inter = shard.ray_mapping(spp)
    return RayBatch(n_rays=n_local * spp, spp=spp, sensor=sensor, ray_offset=off, interleave=inter)


def render_primal(scene: Scene, integrator, sensor: int = 0, spp: int = 1, seed: int = 0,
                  shard: Optional[ShardSpec] = None) -> torch.Tensor:
    """Detached primal image of the local pixels, [n_local_pixels, 3]
    (render_batch_primal, batched.py:134-197)."""
    batch = _sensor_batch(scene, sensor, spp, shard)
    L, _, _ = integrator.sample(ADMode.Primal, scene, IndependentSampler(seed, spp), batch)
    return integrator.develop(scene, L, spp)


def render_backward(scene: Scene, integrator, grad_image: torch.Tensor, sensor: int = 0,
                    spp: int = 1, seed: int = 0, shard: Optional[ShardSpec] = None,
                    grads: Optional[Dict[str, torch.Tensor]] = None,
                    allreduce: bool = True, strict: Optional[bool] = True) -> Dict[str, torch.Tensor]:
    """The H1 sequence (batched.py:212-326) for the local pixels; returns the
    gradient grids - summed over all ranks when the pixels were dealt across a process group
    (`shard.world > 1`); an unsharded call never communicates.  For the sum to be the gradient of
    the GLOBAL loss, `grad_image` must be the derivative of the global loss with respect to the local
    pixels (a mean over the local pixels only over-scales it by `world`: use
    `distributed.local_loss_scale`).  `strict` (sharded calls): True - the check of the packed all-reduce is looked at before this
    call returns (a one-off call has no next call that would look); False / None - it is left to the next backward pass or
    `distributed.verify_pending()` (the autograd ops inside an optimisation loop, which ends with `verify_pending`)."""
    batch = _sensor_batch(scene, sensor, spp, shard)
    sampler = IndependentSampler(seed, spp)
    L, _, state_out = integrator.sample(ADMode.Primal, scene, sampler.clone(), batch)     # :255-264
    dL = integrator.film_backward(scene, grad_image, spp)                                  # :272-306
    if grads is None:
        grads = alloc_grads(scene, integrator.param_keys)
    support = sharded_support(scene, grads, shard) if allreduce else None
    integrator.sample(ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state_out,  # :309-318
                      grads=grads)
    if allreduce:
        allreduce_gradients(grads, shard=shard or ShardSpec(), support=support, strict=strict)
    return grads


class _RenderOp(torch.autograd.Function):
    """Counterpart of `mi._RenderOp` / `_BatchedRenderOp` (batched.py:13-85)."""

    @staticmethod
    def forward(ctx, p0, p1, scene, integrator, sensor, spp, spp_grad, seed, seed_grad, shard):
        sc = _with_params(scene, integrator.param_keys, (p0.detach(), p1.detach()))
        ctx.scene, ctx.integrator, ctx.sensor = sc, integrator, sensor
        ctx.spp_grad, ctx.seed_grad, ctx.shard = spp_grad, seed_grad, shard
        return render_primal(sc, integrator, sensor, spp, seed, shard)

    @staticmethod
    def backward(ctx, grad_image):
        g = render_backward(ctx.scene, ctx.integrator, grad_image.contiguous(), ctx.sensor,
                            ctx.spp_grad, ctx.seed_grad, ctx.shard, strict=False)   # (run_optimization ends with verify_pending)
        k0, k1 = ctx.integrator.param_keys
        return g[k0], g[k1], None, None, None, None, None, None, None, None


def render(scene: Scene, params: Optional[Dict[str, torch.Tensor]] = None, integrator=None,
           sensor: int = 0, spp: int = 1, spp_grad: int = 0, seed: int = 0, seed_grad: int = 0,
           shard: Optional[ShardSpec] = None) -> torch.Tensor:
    """`mi.render`: image of the local pixels, [n_local_pixels, 3] (the whole image,
    row-major, when unsharded - reshape to (H, W, 3)).  Differentiable with respect to
    the integrator's `param_keys` (sigma_t + albedo for `volpathsimple`, sigma_t + emission
    for `nerf`)."""
    if integrator is None:
        raise ValueError("render: an integrator is required")
    if spp_grad == 0:
        spp_grad = spp
    if seed_grad == 0:
        # de-correlate the primal and differential phases (batched.py:117-122)
        seed_grad = sample_tea_32(seed, 1)[0]
    elif seed_grad == seed:
        raise Exception('The primal and differential seed should be different '
                        'to ensure unbiased gradient computation!')
    keys = integrator.param_keys
    if params is None:
        params = {k: _grid(scene, k) for k in keys}
    for k in keys:
        if not isinstance(params[k], torch.Tensor):
            raise TypeError(f"render: params['{k}'] must be a torch device tensor "
                            "(the DRT integrator has no CPU path; use scene_to(scene, device))")
    return _RenderOp.apply(params[keys[0]], params[keys[1]], scene, integrator, sensor,
                           int(spp), int(spp_grad), int(seed), int(seed_grad), shard)
