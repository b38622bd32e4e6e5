"""The inverse-rendering loop around the hot path (N2; reference: python/optimize.py,
python/opt_config.py:11-75): Adam with per-parameter learning rates and the `Last25` schedule,
projection of the parameters onto their legal range, x2 trilinear grid upsampling, majorant
supergrid adjustment, `.vol` checkpoints - all on the device (the reference round-trips the grids
through scipy on the host for upsampling, optimize.py:217-223); reference renderings cached on disk and
preview renderings (N3; PFM instead of EXR, image_io.py).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from enum import IntEnum
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from . import losses
from .batched import gather_ref_values, render_batch, sensors_to_device
from .integrators import sample_tea_32
from .opt_config import get_int_config
from .render import render, render_primal
from .scene import ALBEDO_KEY, EMISSION_KEY, SIGMA_T_KEY, GridMedium, Scene
from .image_io import read_image, write_image
from .volume_io import write_vol


class Schedule(IntEnum):
    Constant = 0
    Last25 = 1


@dataclass
class OptimizationConfig:
    """python/opt_config.py:11-75 (same fields and defaults)."""
    name: str
    spp: int
    n_iter: int
    lr: float
    primal_spp_factor: int = 64
    batch_size: Optional[int] = None
    lr_schedule: Optional[Schedule] = None
    upsample: Optional[List[float]] = None
    base_seed: int = 988378
    render_initial: bool = True
    render_final: bool = True
    preview_stride: int = 100
    checkpoint_initial: bool = True
    checkpoint_final: bool = True
    checkpoint_stride: int = 1000
    preview_spp: Optional[int] = None
    opt_type: str = 'adam'
    opt_args: Optional[Dict] = None
    loss: Callable = losses.l1

    def __post_init__(self):
        self.upsample_at = set()
        if self.upsample:
            for t in self.upsample:
                assert t >= 0 and t <= 1
                self.upsample_at.add(int(t * self.n_iter))

    def optimizer(self, params):
        opt_type = {'sgd': SGD, 'adam': Adam}[self.opt_type]
        return opt_type(lr=self.lr, params=params, **(self.opt_args or {}))

    def learning_rates(self, scene_config, it_i):
        schedule_factor = 1.0
        if self.lr_schedule not in (None, Schedule.Constant):
            t = it_i / (self.n_iter - 1)
            if self.lr_schedule == Schedule.Last25:
                steps = [0.75, 0.85, 0.95]
            else:
                raise ValueError(f'Unsupported schedule: {self.lr_schedule}')
            for s in steps:
                if t >= s:
                    schedule_factor *= 0.5
        upsampling_factor = 1.0
        return {k: (schedule_factor * upsampling_factor * scene_config.param_lr_factors.get(k, 1.0) * self.lr)
                for k in scene_config.param_keys}

    def should_upsample(self, it_i):
        if not self.upsample_at:
            return False
        return it_i in self.upsample_at


@dataclass
class SceneConfig:
    """The data fields of python/scene_config.py:9-72 for scenes given as objects (the reference's
    registry points at XML files and assets that are not part of its repository)."""
    name: str
    scene: Scene
    param_keys: List[str]
    sensors: List[int]
    start_from_value: Dict[str, Optional[float]]
    max_depth: int = 64
    ref_spp: int = 8192
    ref_integrator: str = 'volpathsimple-drt'
    preview_sensors: Optional[List[int]] = None
    max_density: float = 250
    # scene_config.py:36.  On MI355X the global majorant (0) is ~2x faster than the supergrid DDA for
    # sparse volumes and the estimators agree in expectation: pass 0 for speed.
    majorant_resolution_factor: int = 8
    param_lr_factors: Optional[Dict[str, float]] = None
    references: Optional[str] = None       # scene_config.py:47-50: directory of the cached reference renderings

    def __post_init__(self):
        for k in self.param_keys:
            if k not in self.start_from_value:
                raise ValueError(f'Parameter "{k}" will be optimized but was not given an initial value in `start_from_value`')
        if not self.preview_sensors:
            self.preview_sensors = [self.sensors[0]]
        if not self.param_lr_factors:
            self.param_lr_factors = {k: 2.0 for k in self.param_keys if '.albedo.' in k}   # scene_config.py:67-71


def _fused_adam_ok(p, g, m, v) -> bool:
    """True iff the one-pass device kernel (drt_adam_step: float4 accesses) can take this parameter: device tensors on ONE
    device, float32, contiguous, equal sizes, every pointer 16-byte aligned.  Anything else - e.g. the gradient view of a
    grid whose voxel count is not a multiple of 4 behind another grid in `alloc_grads`' flat buffer - takes the torch ops."""
    ts = (p, g, m, v)
    return (p.is_cuda and all(t.device == p.device and t.dtype == torch.float32 and t.is_contiguous() and
                              t.numel() == p.numel() and t.data_ptr() % 16 == 0 for t in ts))


class Adam:
    """mi.ad.Adam [M3-ext] as the reference uses it (opt_config.py:46-48, optimize.py:329,352-354):
    bias-corrected Adam (beta1 0.9, beta2 0.999, epsilon 1e-8), per-parameter learning rates, state
    dropped when a parameter changes shape (upsampling)."""

    def __init__(self, lr, params: Dict[str, torch.Tensor], beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        self.lr_default = lr
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.variables: Dict[str, torch.Tensor] = dict(params)
        self.lr: Dict[str, float] = {}
        self.state: Dict[str, tuple] = {}

    def __getitem__(self, k):
        return self.variables[k]

    def __setitem__(self, k, v):
        if k in self.state and self.state[k][1].shape != v.shape:
            del self.state[k]
        self.variables[k] = v

    def items(self):
        return self.variables.items()

    def set_learning_rate(self, lr):
        if isinstance(lr, dict):
            self.lr.update(lr)
        else:
            self.lr_default = lr

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor], bounds: Optional[Dict[str, tuple]] = None):
        """`bounds` {key: (lo, hi)} (None = open): the parameter's valid range, applied to the updated value in the SAME device
        pass where the fused kernel takes the parameter (drt_adam_step_clamped) - opt.step() + enforce_valid_params
        (python/optimize.py:352-353) in one pass over p, g, m, v instead of that plus a clamp kernel per grid.  Returns the
        keys whose bounds were applied here (the caller clamps the others)."""
        clamped = set()
        for k, p in self.variables.items():
            g = grads.get(k)
            if g is None:
                continue
            t, m, v = self.state.get(k, (0, torch.zeros_like(p), torch.zeros_like(p)))
            t += 1
            lr = self.lr.get(k, self.lr_default)
            lr_t = lr * (1 - self.beta_2 ** t) ** 0.5 / (1 - self.beta_1 ** t)
            if _fused_adam_ok(p, g, m, v):
                # one fused pass on the device (drt_adam_step) instead of seven elementwise kernels
                from ._native import native
                with torch.cuda.device(p.device):
                    if bounds and k in bounds:
                        lo, hi = bounds[k]
                        native().adam_step_clamped(torch.cuda.current_stream().cuda_stream, p.data_ptr(), g.data_ptr(), m.data_ptr(),
                                                   v.data_ptr(), p.numel(), self.beta_1, self.beta_2, self.epsilon, lr_t,
                                                   float("-inf") if lo is None else float(lo), float("inf") if hi is None else float(hi))
                        clamped.add(k)
                    else:
                        native().adam_step(torch.cuda.current_stream().cuda_stream, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                           p.numel(), self.beta_1, self.beta_2, self.epsilon, lr_t)
                p.view(-1)[:0].zero_()                                     # bumps the tensor version (the medium is re-bound), no work
            else:
                m.mul_(self.beta_1).add_(g, alpha=1 - self.beta_1)
                v.mul_(self.beta_2).addcmul_(g, g, value=1 - self.beta_2)
                p.addcdiv_(m, v.sqrt().add_(self.epsilon), value=-lr_t)    # in place: bumps the tensor version
            self.state[k] = (t, m, v)
        return clamped


class SGD(Adam):
    """mi.ad.SGD without momentum."""

    @torch.no_grad()
    def step(self, grads, bounds=None):
        for k, p in self.variables.items():
            if grads.get(k) is not None:
                p.add_(grads[k], alpha=-self.lr.get(k, self.lr_default))
        return set()


def param_bounds(scene_config: SceneConfig, keys) -> Dict[str, tuple]:
    """The valid range of each parameter (optimize.py:169-179), (lo, hi) with None = open."""
    out = {}
    for k in keys:
        if k.endswith('sigma_t.data'):
            out[k] = (0.0, scene_config.max_density)
        elif k.endswith('emission.data'):
            out[k] = (0.0, None)
        elif k.endswith('albedo.data'):
            out[k] = (0.0, 1.0)
    return out


@torch.no_grad()
def enforce_valid_params(scene_config: SceneConfig, opt, skip=()) -> None:
    """optimize.py:169-179.  `skip`: keys whose range the optimizer step has applied already (Adam.step(bounds=...))."""
    for k, v in opt.items():
        if k in skip:
            continue
        if k.endswith('sigma_t.data'):
            v.clamp_(0, scene_config.max_density)
        elif k.endswith('emission.data'):
            v.clamp_(min=0)
        elif k.endswith('albedo.data'):
            v.clamp_(0, 1)
        else:
            raise ValueError(k)


def adjusted_majorant_res_factor(res_factor: int, density_res) -> int:
    """optimize.py:182-193: the largest factor <= the configured one that leaves >= 4 supergrid cells
    along the shortest side, else 0 (supergrid disabled)."""
    if res_factor > 1:
        min_side = min(density_res[:3])
        while res_factor > 1 and (min_side // res_factor) < 4:
            res_factor -= 1
    if res_factor <= 1:
        res_factor = 0
    return res_factor


@torch.no_grad()
def upsample_grid(values: torch.Tensor, new_res) -> torch.Tensor:
    """optimize.py:203-225: first-order interpolation identical to
    `scipy.ndimage.zoom(order=1, mode='nearest', prefilter=False, grid_mode=True)` = trilinear
    resampling with half-voxel-centred coordinates and edge replication, on the device."""
    z, y, x, c = values.shape
    if tuple(new_res) == (z, y, x, c):
        return values.detach().clone()
    assert new_res[-1] == c
    v = values.permute(3, 0, 1, 2).unsqueeze(0)                       # (1, C, Z, Y, X)
    out = F.interpolate(v, size=tuple(new_res[:3]), mode='trilinear', align_corners=False)
    return out[0].permute(1, 2, 3, 0).contiguous()


def save_params(output_dir: str, scene_config: SceneConfig, params: Dict[str, torch.Tensor], name: str, medium: GridMedium) -> None:
    """python/util.py:55-71: one `.vol` per parameter, `<name>-medium1_sigma_t.vol`."""
    os.makedirs(output_dir, exist_ok=True)                            # util.py:57 (create_checkpoint always does)
    for key in scene_config.param_keys:
        if not key.endswith('.data'):
            raise NotImplementedError(f'Checkpointing of parameter {key}')
        var_name = '_'.join(key[:-len('.data')].strip().split('.'))
        write_vol(os.path.join(output_dir, f'{name}-{var_name}.vol'), params[key], medium.bbox_min, medium.bbox_max)


IMAGE_EXT = '.pfm'      # the reference writes .exr (optimize.py:50,131); OpenEXR does not exist here (image_io.py)


def render_reference_image(scene_config: SceneConfig, to_render: Dict[int, Optional[str]], seed: int = 1234,
                           max_rays_per_pass: int = 720 * 720 * 2048) -> Dict[int, torch.Tensor]:
    """python/optimize.py:24-50: render the sensors `to_render` ({sensor id: file name or None}) of the reference
    scene at `ref_spp` with `ref_integrator`, in passes of at most `max_rays_per_pass` rays (pass i uses seed + i,
    the result is the mean of the passes), and write each to its file.  Returns {sensor id: (H, W, 3) tensor}."""
    scene = scene_config.scene
    integrator = get_int_config(scene_config.ref_integrator).create(max_depth=scene_config.max_depth)
    ref_spp = scene_config.ref_spp
    out = {}
    for s, fname in to_render.items():
        sensor = scene.sensors[s]
        total_rays = sensor.width * sensor.height * ref_spp
        pass_count = -(-total_rays // max_rays_per_pass)
        spp_per_pass = -(-ref_spp // pass_count)
        assert spp_per_pass * pass_count >= ref_spp
        result = None
        for pass_i in range(pass_count):
            image = render_primal(scene, integrator, s, spp_per_pass, seed + pass_i) / pass_count
            result = image if result is None else result + image
        result = result.view(sensor.height, sensor.width, 3)
        if fname:
            write_image(fname, result)
        out[s] = result
    return out


def get_reference_image_paths(scene_config: SceneConfig, overwrite: bool = False) -> Dict[int, str]:
    """python/optimize.py:53-68: `<references>/ref_<sensor id>` for every sensor of the configuration; the
    missing ones (all with `overwrite`) are rendered first."""
    if not scene_config.references:
        raise ValueError('SceneConfig.references (the directory of the reference renderings) is not set')
    os.makedirs(scene_config.references, exist_ok=True)
    paths = {s: os.path.join(scene_config.references, f'ref_{s:06d}{IMAGE_EXT}') for s in scene_config.sensors}
    missing = dict(paths) if overwrite else {s: f for s, f in paths.items() if not os.path.isfile(f)}
    if missing:
        render_reference_image(scene_config, missing)
    return paths


def load_reference_images(paths: Dict[int, str], batchify: bool = False, device=None):
    """python/optimize.py:71-85: one (n_sensors, H, W, 3) tensor in the order of `paths` (batched rendering gathers
    from it) or {sensor id: (H, W, 3) tensor}."""
    imgs = {s: torch.from_numpy(read_image(f)).to(device) for s, f in paths.items()}
    if batchify:
        return torch.stack(list(imgs.values()))
    return imgs


def render_previews(output_dir: str, opt_config: OptimizationConfig, scene_config: SceneConfig, scene: Scene,
                    integrator, it_i) -> List[str]:
    """python/optimize.py:108-131: `opt<suffix>_<sensor>` for every preview sensor, seed 1234, `preview_spp`.
    `scene.sensors` is the configuration's sensor list; `scene_config.preview_sensors` holds ids of the full scene."""
    if it_i == 'initial':
        if not opt_config.render_initial:
            return []
        suffix = '_init'
    elif it_i == 'final':
        if not opt_config.render_final:
            return []
        suffix = '_final'
    elif isinstance(it_i, int):
        suffix = f'_{it_i:08d}'
    else:
        assert isinstance(it_i, str)
        suffix = it_i
    preview_spp = opt_config.preview_spp or opt_config.spp
    full = Scene(medium=scene.medium, emitter=scene.emitter, sensors=scene_config.scene.sensors)
    written = []
    for s in scene_config.preview_sensors:
        sensor = full.sensors[s]
        fname = os.path.join(output_dir, f'opt{suffix}_{s:04d}{IMAGE_EXT}')
        image = render_primal(full, integrator, s, preview_spp, 1234)
        write_image(fname, image.view(sensor.height, sensor.width, 3))
        written.append(fname)
    return written


def _scene_with(scene: Scene, params: Dict[str, torch.Tensor], factor: int) -> Scene:
    m = scene.medium
    medium = GridMedium(sigma_t=params.get(SIGMA_T_KEY, m.sigma_t), albedo=params.get(ALBEDO_KEY, m.albedo),
                        bbox_min=m.bbox_min, bbox_max=m.bbox_max, scale=m.scale, majorant_resolution_factor=factor,
                        emission=params.get(EMISSION_KEY, m.emission))
    return Scene(medium=medium, emitter=scene.emitter, sensors=scene.sensors)


def run_optimization(output_dir: Optional[str], opt_config: OptimizationConfig, scene_config: SceneConfig,
                     int_config, ref_images: Optional[torch.Tensor] = None, progress: Optional[Callable] = None,
                     shard=None):
    """python/optimize.py:275-365.  `ref_images`: (n_sensors, H, W, 3) reference tensor; rendered from
    `scene_config.scene` at `ref_spp` with `ref_integrator` when None (optimize.py:24-87).
    Returns (scene, params, opt, losses).

    `scene_config.param_keys` may be any subset / superset of the integrator's keys (the reference's configs
    list sigma_t, albedo AND emission whatever the integrator, scene_config.py:148): keys the integrator does
    not read simply receive no gradient, keys it reads but that are not optimised are taken from the scene.

    `shard` (a `ShardSpec` with world > 1, one process per GPU): the batch / the image pixels of every
    iteration are dealt across the ranks, the local loss is scaled to its share of the global loss and the
    gradient grids are summed with one all-reduce per backward; every rank then takes the identical
    optimizer step."""
    from .distributed import ShardSpec, allreduce_scalar, local_loss_scale, verify_pending
    from . import losses as _losses
    shard = shard or ShardSpec()
    if shard.partitioned:
        # every rank back-propagates its share n_local / n_global of the loss: that is the global gradient only for losses
        # that are sums over entries, and needs at least one entry per rank (an empty share's loss is 0 / 0)
        separable = (_losses.l1, _losses.l2, _losses.huber, _losses.average, _losses.mean_relative_absolute_error,
                     _losses.mean_relative_squared_error)
        import functools
        loss_fn = opt_config.loss
        while isinstance(loss_fn, functools.partial):            # e.g. partial(huber, delta=...): still a sum over entries
            loss_fn = loss_fn.func
        if loss_fn not in separable:
            raise ValueError(f"sharded run_optimization needs a loss that is a sum over image entries (l1, l2, huber, "
                             f"mean_relative_*), not {getattr(opt_config.loss, '__name__', opt_config.loss)!r}: its gradient "
                             f"is not the sum of the ranks' partial gradients")
        if opt_config.batch_size is not None and opt_config.batch_size < shard.world:
            raise ValueError(f"batch_size {opt_config.batch_size} < world size {shard.world}: some ranks would get no entry")
        if opt_config.batch_size is None:
            # sensor mode: the image's pixels are dealt out in chunks; an image with fewer than chunk_pixels * world pixels
            # (or not a multiple of it) leaves ranks empty - their 0 / 0 loss would spread through the all-reduce
            for s_idx in scene_config.sensors:
                sen = scene_config.scene.sensors[s_idx]
                if shard.n_local_pixels(sen.width * sen.height) <= 0:      # (raises itself when the image cannot be dealt)
                    raise ValueError(f"sensor {s_idx}: {sen.width}x{sen.height} pixels leave ranks of a world of {shard.world} empty")
    int_config = get_int_config(int_config)
    if int_config.name == 'nerf-drt-fused':
        raise ValueError("'nerf-drt-fused' renders two images per ray ([n, 6]: nerf | volpathsimple) for the fused benchmark pass; "
                         "run_optimization compares one image with the references - use 'nerf' or 'volpathsimple-drt'")
    scene0 = scene_config.scene
    dev = scene0.medium.sigma_t.device
    integrator = int_config.create(max_depth=scene_config.max_depth)
    keys = list(scene_config.param_keys)
    sensors = [scene0.sensors[i] for i in scene_config.sensors]
    n_sensors = len(sensors)
    film = (sensors[0].width, sensors[0].height)
    spp_grad = opt_config.spp
    spp_primal = spp_grad * opt_config.primal_spp_factor
    if output_dir:
        os.makedirs(os.path.join(output_dir, 'params'), exist_ok=True)       # util.py:55-71 always creates it

    if ref_images is None:                                             # reference renderings (optimize.py:284-285)
        if scene_config.references:                                    # cached on disk; rank 0 renders what is missing
            if shard.partitioned:
                import torch.distributed as dist
                if shard.rank == 0:
                    get_reference_image_paths(scene_config)
                dist.barrier()
            ref_images = load_reference_images(get_reference_image_paths(scene_config), batchify=True, device=dev)
        else:
            rendered = render_reference_image(scene_config, {s: None for s in scene_config.sensors})
            ref_images = torch.stack([rendered[s] for s in scene_config.sensors])

    # --- initialisation (optimize.py:134-166)
    full = {SIGMA_T_KEY: scene0.medium.sigma_t, ALBEDO_KEY: scene0.medium.albedo, EMISSION_KEY: scene0.medium.emission}
    base_res = tuple(scene0.medium.sigma_t.shape[:3])
    n_up = len(opt_config.upsample) if opt_config.upsample else 0
    params: Dict[str, torch.Tensor] = {}
    for k in keys:
        if k not in full:
            raise ValueError(f'Unknown parameter key "{k}" (known: {sorted(full)})')
        channels = 1 if k == SIGMA_T_KEY else 3
        shape = tuple(full[k].shape) if full[k] is not None else base_res + (channels,)
        init_res = tuple(max(1, s // (2 ** n_up)) for s in shape[:3]) + (shape[-1],)
        if n_up and 1 in init_res[:3]:
            raise ValueError(f'Initial resolution not supported: {init_res}. Maybe reduce upsample_steps?')
        v = scene_config.start_from_value[k]
        if v is None:
            assert not opt_config.upsample
            if full[k] is None:
                raise ValueError(f'Parameter "{k}" has neither an initial value nor a grid in the scene')
            params[k] = full[k].detach().clone()
        else:
            params[k] = torch.full(init_res, float(v), dtype=torch.float32, device=dev)
    for k in integrator.param_keys:
        if k not in params and full[k] is None:
            raise ValueError(f'The integrator reads "{k}" but it is neither optimised nor present in the scene')

    def current_grids():
        """The grids the integrator reads: optimised ones from `params`, the others from the scene AS THEY ARE - a fixed grid next to
        optimised ones at another resolution (multi-resolution schedules upsample only what is optimised, optimize.py:228-252 of the
        reference) stays on its own lattice, as Mitsuba keeps it; colour grids that differ from sigma_t's lattice run the own-lattice
        kernels (drt_set_colour_resolution).  Albedo and emission must agree with each other where an integrator reads both."""
        out = {}
        for k in (SIGMA_T_KEY, ALBEDO_KEY, EMISSION_KEY):
            if k in params:
                out[k] = params[k]
            elif full[k] is not None:
                out[k] = full[k]
        return out

    grids = current_grids()
    factor = adjusted_majorant_res_factor(scene_config.majorant_resolution_factor, grids[SIGMA_T_KEY].shape)
    opt = opt_config.optimizer(params)
    scene = _scene_with(Scene(scene0.medium, scene0.emitter, sensors), grids, factor)
    table = sensors_to_device(sensors, dev)
    writer = bool(output_dir) and shard.rank == 0                      # one rank writes checkpoints and previews
    if writer and opt_config.checkpoint_initial:
        save_params(os.path.join(output_dir, 'params'), scene_config, params, 'initial', scene.medium)
    if writer:
        render_previews(output_dir, opt_config, scene_config, scene, integrator, 'initial')   # optimize.py:320
        for s in scene_config.preview_sensors:                         # the matching references, for comparison (:321-324)
            if s in scene_config.sensors:
                write_image(os.path.join(output_dir, f'ref_{s:04d}{IMAGE_EXT}'), ref_images[scene_config.sensors.index(s)])

    host_rng = torch.Generator().manual_seed(93483)                    # sensor choice (optimize.py:291,344)
    history = []
    for it_i in range(opt_config.n_iter):
        seed = sample_tea_32(2 * it_i + 0, opt_config.base_seed)[0]
        seed_grad = sample_tea_32(2 * it_i + 1, opt_config.base_seed)[0]
        opt.set_learning_rate(opt_config.learning_rates(scene_config, it_i))
        if opt_config.should_upsample(it_i):                           # optimize.py:228-252
            for k in keys:
                old = opt[k]
                new_res = tuple(2 * r for r in old.shape[:3]) + (old.shape[-1],)
                opt[k] = upsample_grid(old, new_res)
                params[k] = opt[k]
            grids = current_grids()
            factor = adjusted_majorant_res_factor(scene_config.majorant_resolution_factor, grids[SIGMA_T_KEY].shape)
            scene = _scene_with(scene, grids, factor)
        # leaves: what the integrator differentiates; only the optimised ones require (and receive) gradients
        leaves = {k: (params[k].detach().requires_grad_(True) if k in params else grids[k].detach())
                  for k in integrator.param_keys}
        if opt_config.batch_size is not None:                          # batched rendering (:332-341)
            image, _, _, sensor_idx, pixel_idx = render_batch(
                opt_config.batch_size, scene, sensors=sensors, params=leaves, integrator=integrator,
                spp=spp_primal, spp_grad=spp_grad, seed=seed, seed_grad=seed_grad, sensor_table=table, shard=shard)
            ref_values = gather_ref_values(ref_images, sensor_idx, pixel_idx)
            n_global = opt_config.batch_size
        else:                                                          # sensor-based rendering (:342-348)
            s_i = int(torch.rand((), generator=host_rng).item() * n_sensors)
            image = render(scene, params=leaves, integrator=integrator, sensor=s_i, spp=spp_primal,
                           spp_grad=spp_grad, seed=seed, seed_grad=seed_grad, shard=shard if shard.partitioned else None)
            ref_values = ref_images[s_i].reshape(-1, 3)
            n_global = ref_values.shape[0]
            if shard.partitioned:
                ref_values = ref_values[shard.pixel_indices(n_global, ref_values.device)]
        # the losses normalise by the local entry count: scale to this rank's share of the global loss
        loss_value = opt_config.loss(image, ref_values) * local_loss_scale(image.shape[0], n_global)
        loss_value.backward()                                          # dr.backward (:350)
        done = opt.step({k: leaves[k].grad for k in keys if k in leaves and leaves[k].requires_grad},    # :352
                        bounds=param_bounds(scene_config, keys))
        enforce_valid_params(scene_config, opt, skip=done or ())       # :353 (what the fused step did not clamp itself)
        # (the loss stays on the device: the reference does not read it back at all - python/optimize.py:325-358 logs nothing per iteration -, and
        #  a float() here would make the host wait for iteration i before it can enqueue iteration i + 1.  `history` is converted once, behind
        #  the loop; a `progress` callback receives the 0-d device tensor and pays for the wait only if it looks at the value)
        total = allreduce_scalar(loss_value.detach()) if shard.partitioned else loss_value.detach()
        history.append(total)
        if shard.partitioned and opt_config.checkpoint_stride and it_i > 0 and it_i % opt_config.checkpoint_stride == 0:
            verify_pending()                                           # (no checkpoint of parameters that took an unsummed gradient)
        if writer and it_i > 0 and opt_config.checkpoint_stride and it_i % opt_config.checkpoint_stride == 0:
            save_params(os.path.join(output_dir, 'params'), scene_config, params, f'{it_i:08d}', scene.medium)
        if writer and it_i > 0 and opt_config.preview_stride and it_i % opt_config.preview_stride == 0:   # :357-358
            render_previews(output_dir, opt_config, scene_config, scene, integrator, it_i)
        if progress:
            progress(it_i, history[-1])
    if shard.partitioned:
        # the packed all-reduce of the last backward pass left its check to "the next call": this is it (raised, never silent)
        verify_pending()
    if writer and opt_config.checkpoint_final:
        save_params(os.path.join(output_dir, 'params'), scene_config, params, 'final', scene.medium)
    if writer:
        render_previews(output_dir, opt_config, scene_config, scene, integrator, 'final')     # :362
    history = [float(v) for v in torch.stack(history).tolist()] if history else []            # ONE device -> host copy for the whole run
    return scene, params, opt, history
