"""Seeded synthetic stand-ins for the reference's scenes.

The real assets (`volumes/janga-smoke-264-136-136.vol`, the dust-devil grids,
envmaps; python/scene_config.py:108,152) are a separate download that is not part
of the reference repository, so the benchmark volumes are committed as
*generators*, not data.  Field names and defaults follow `SceneConfig`
(python/scene_config.py:9-72): max_depth 64, albedo 0.6 for smoke, sand-like albedo
for the dust devil, constant white emitter, perspective sensors on a ring.
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn.functional as F

from .scene import ConstantEmitter, GridMedium, PerspectiveSensor, Scene


def _fbm(res: int, seed: int, octaves: int = 5, base: int = 4, device="cpu") -> torch.Tensor:
    """Value-noise fBm on a res^3 grid in [0,1]: per octave a seeded random lattice
    (generated on the CPU for reproducibility) upsampled trilinearly."""
    out = torch.zeros((1, 1, res, res, res), dtype=torch.float32, device=device)
    amp, total = 1.0, 0.0
    for k in range(octaves):
        n = base * (2 ** k) + 1
        g = torch.Generator().manual_seed(seed * 1000 + k)
        lattice = torch.rand((1, 1, n, n, n), generator=g, dtype=torch.float32).to(device)
        out += amp * F.interpolate(lattice, size=(res, res, res), mode="trilinear", align_corners=True)
        total += amp
        amp *= 0.5
    return (out / total)[0, 0]


def _shape_density(d: torch.Tensor, majorant: float, floor: float = 0.06, q: float = 0.97) -> torch.Tensor:
    """Normalise a raw density field: the q-quantile of the occupied voxels maps to the
    majorant (values above are clipped, so the maximum is not a lone outlier), values
    below `floor` of it become exactly empty space."""
    flat = d.reshape(-1)
    occ = flat[flat > 1e-4 * flat.max()]
    sub = occ[:: max(1, occ.numel() // 1_000_000)]
    ref = torch.quantile(sub, q).clamp_min(1e-12)
    d = torch.clamp(d / ref, max=1.0)
    d = torch.where(d < floor, torch.zeros_like(d), d)
    return d * majorant


def _coords(res: int, device):
    c = (torch.arange(res, dtype=torch.float32, device=device) + 0.5) / res
    z, y, x = torch.meshgrid(c, c, c, indexing="ij")   # grids are (Z, Y, X)
    return x, y, z


def ring_sensors(n: int, radius: float, height: float, target=(0.0, 0.0, 0.0), fov: float = 30.0,
                 width: int = 512, film_height: int = 512) -> List[PerspectiveSensor]:
    """`n` perspective sensors on a horizontal ring looking at `target`
    (the reference's multi-view setups list 62/63 sensors, scene_config.py:114,164)."""
    out = []
    for i in range(n):
        a = 2.0 * math.pi * i / n
        out.append(PerspectiveSensor(origin=(radius * math.cos(a), height, radius * math.sin(a)),
                                     target=target, up=(0.0, 1.0, 0.0), fov=fov,
                                     width=width, height=film_height))
    return out


def constant_cube_scene(res: int = 64, sigma_t: float = 1.0, albedo: float = 0.8,
                        film: int = 128, device="cpu") -> Scene:
    """BASELINE config 1: res^3 constant-sigma_t cube, box [-0.5,1.5]^3, camera and light
    of the reference fixture (tests/test_integrators.py:46-77)."""
    st = torch.full((res, res, res, 1), float(sigma_t), dtype=torch.float32, device=device)
    al = torch.full((res, res, res, 3), float(albedo), dtype=torch.float32, device=device)
    medium = GridMedium(sigma_t=st, albedo=al, bbox_min=(-0.5, -0.5, -0.5), bbox_max=(1.5, 1.5, 1.5))
    sensor = PerspectiveSensor(origin=(4.0, 4.0, 4.0), target=(0.0, -0.15, 0.0), fov=30.0,
                               width=film, height=film)
    return Scene(medium=medium, emitter=ConstantEmitter((1.0, 0.8, 0.2)), sensors=[sensor])


def _film(film):
    """(width, height) of a film given as one number (square) or a pair - the paper's scenes render 720 x 620 (scene_config.py:100-101)."""
    return (int(film), int(film)) if isinstance(film, (int, float)) else (int(film[0]), int(film[1]))


def smoke_scene(res: int = 128, film=512, seed: int = 1234, device="cpu",
                optical_side: float = 16.0) -> Scene:
    """BASELINE config 2 stand-in for janga-smoke: sigma_t = s * max(0, fbm - 0.45) * plume,
    s chosen so that majorant * bbox side = `optical_side`; albedo 0.6
    (scene_config.py:118); constant white emitter; one sensor 2.5 box sides away."""
    x, y, z = _coords(res, device)
    n = _fbm(res, seed, device=device)
    r2 = (x - 0.5) ** 2 + (z - 0.5) ** 2
    width = 0.05 + 0.18 * y                       # plume widens with height
    plume = torch.exp(-r2 / (2.0 * width ** 2)) * torch.clamp(1.2 - y, 0.0, 1.0) * torch.clamp(y * 8.0, 0.0, 1.0)
    d = torch.clamp(n - 0.45, min=0.0) * plume
    side = 2.0
    d = _shape_density(d, optical_side / side)
    st = d.unsqueeze(-1).contiguous()
    al = torch.full((res, res, res, 3), 0.6, dtype=torch.float32, device=device)
    medium = GridMedium(sigma_t=st, albedo=al, bbox_min=(-1.0, -1.0, -1.0), bbox_max=(1.0, 1.0, 1.0))
    fw, fh = _film(film)
    sensor = PerspectiveSensor(origin=(0.0, 0.6, 5.0), target=(0.0, 0.0, 0.0), fov=30.0, width=fw, height=fh)
    return Scene(medium=medium, emitter=ConstantEmitter((1.0, 1.0, 1.0)), sensors=[sensor])


def smoke_scene_janga_shape(film=(720, 620), seed: int = 1234, device="cpu", optical_side: float = 16.0) -> Scene:
    """A smoke plume on a grid of janga-smoke's REAL shape, 264 x 136 x 136 voxels (`volumes/janga-smoke-264-136-136.vol`,
    scene_config.py:108; the asset itself is a separate download): the generator of `smoke_scene` at 264^3, the plume's axis laid along
    the grid's long (x) axis and the other two axes cropped to 136 voxels around it; cubic voxels, i.e. a box of 3.88 x 2 x 2."""
    cube = smoke_scene(res=264, film=film, seed=seed, device=device, optical_side=optical_side)
    st = cube.medium.sigma_t[..., 0]                               # (Z, Y, X): the plume rises along Y
    st = st.permute(0, 2, 1)                                       # -> its axis along the last (x) index
    st = st[64:200, 64:200, :].contiguous().unsqueeze(-1)          # 136 x 136 x 264
    al = torch.full(tuple(st.shape[:3]) + (3,), 0.6, dtype=torch.float32, device=device)
    hx = 264.0 / 136.0
    medium = GridMedium(sigma_t=st, albedo=al, bbox_min=(-hx, -1.0, -1.0), bbox_max=(hx, 1.0, 1.0))
    fw, fh = _film(film)
    sensor = PerspectiveSensor(origin=(1.5, 1.2, 7.5), target=(0.0, 0.0, 0.0), fov=30.0, width=fw, height=fh)
    return Scene(medium=medium, emitter=ConstantEmitter((1.0, 1.0, 1.0)), sensors=[sensor])


def dust_devil_scene(res: int = 256, film=512, seed: int = 4321, device="cpu",
                     optical_side: float = 20.0, n_sensors: int = 1) -> Scene:
    """BASELINE config 3 / headline stand-in for dust-devil: a swirling vortex column,
    sigma_t (res^3 x 1) + sand-like albedo (res^3 x 3); `n_sensors` on a ring
    (63 in the reference, scene_config.py:164)."""
    x, y, z = _coords(res, device)
    n = _fbm(res, seed, device=device)
    cx = 0.5 + 0.06 * torch.sin(6.0 * y)
    cz = 0.5 + 0.06 * torch.cos(6.0 * y)
    dx, dz = x - cx, z - cz
    r = torch.sqrt(dx * dx + dz * dz)
    theta = torch.atan2(dz, dx)
    R = 0.06 + 0.30 * y ** 1.5                    # funnel radius grows with height
    wall = torch.exp(-((r - 0.75 * R) / (0.30 * R + 1e-3)) ** 2)
    swirl = 0.55 + 0.45 * torch.sin(3.0 * theta + 14.0 * y)
    base_cloud = torch.exp(-((y - 0.04) / 0.05) ** 2) * torch.exp(-(r / 0.35) ** 2)
    d = (wall * swirl * torch.clamp(1.1 - y, 0.0, 1.0) + 0.8 * base_cloud) * torch.clamp(n * 1.8 - 0.5, min=0.0)
    side = 2.0
    d = _shape_density(d, optical_side / side)
    st = d.unsqueeze(-1).contiguous()
    sand = torch.tensor([0.8, 0.65, 0.45], dtype=torch.float32, device=device)
    al = (sand * (0.9 + 0.1 * _fbm(res, seed + 1, octaves=3, device=device)).unsqueeze(-1)).contiguous()
    medium = GridMedium(sigma_t=st, albedo=al, bbox_min=(-1.0, -1.0, -1.0), bbox_max=(1.0, 1.0, 1.0))
    fw, fh = _film(film)
    sensors = ring_sensors(n_sensors, radius=5.0, height=0.6, fov=30.0, width=fw, film_height=fh)
    return Scene(medium=medium, emitter=ConstantEmitter((1.0, 1.0, 1.0)), sensors=sensors)
