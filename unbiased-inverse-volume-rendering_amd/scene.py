"""Scene description for the DRT hot path: one heterogeneous medium in an
axis-aligned box, one infinite emitter, perspective sensors.

This is the slice of a Mitsuba scene that `VolpathSimpleIntegrator` is allowed to
see (reference: python/integrators/volpathsimple.py:11-17 "no surfaces, one medium
in a convex bounding volume with a null BSDF, one infinite emitter") expressed as
plain data.  Parameter tensors keep Mitsuba's `VolumeGrid` layout `(Z, Y, X, C)`
and are addressed by the reference's keys (`medium1.sigma_t.data`,
`medium1.albedo.data`; python/scene_config.py:98).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

SIGMA_T_KEY = "medium1.sigma_t.data"
ALBEDO_KEY = "medium1.albedo.data"
EMISSION_KEY = "medium1.emission.data"


def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


@dataclass
class PerspectiveSensor:
    """`perspective` sensor + `hdrfilm` with a box filter
    (reference fixture: tests/test_integrators.py:46-67).

    The frame follows Mitsuba's `look_at`: `left = normalize(cross(up, dir))`,
    `up' = cross(dir, left)`; film sample (0, 0) is the top-left corner and the
    field of view is measured along the x axis.
    """
    origin: Sequence[float]
    target: Sequence[float]
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fov: float = 30.0
    width: int = 128
    height: int = 128

    def frame(self) -> Dict[str, np.ndarray]:
        o = np.asarray(self.origin, dtype=np.float64)
        d = _normalize(np.asarray(self.target, dtype=np.float64) - o)
        left = _normalize(np.cross(np.asarray(self.up, dtype=np.float64), d))
        up = np.cross(d, left)
        tan_x = math.tan(math.radians(self.fov) * 0.5)
        tan_y = tan_x * self.height / self.width
        f32 = lambda a: np.asarray(a, dtype=np.float32)
        return dict(origin=f32(o), left=f32(left), up=f32(up), dir=f32(d),
                    tan_x=np.float32(tan_x), tan_y=np.float32(tan_y))


@dataclass
class ConstantEmitter:
    """`constant` environment emitter (tests/test_integrators.py:73-77)."""
    radiance: Sequence[float] = (1.0, 1.0, 1.0)


@dataclass
class EnvmapEmitter:
    """`envmap` environment emitter (python/scene_config.py:102,152,210,262,313): a lat-long RGB
    bitmap of shape (H, W, 3) float32 (row 0 = the +Y pole; numpy array or torch tensor - a device
    tensor for the HIP path), multiplied by `scale` and rotated by the 3x3 `to_world`.

    Local direction (sin phi sin theta, cos theta, -cos phi sin theta) <-> uv = (phi / 2pi,
    theta / pi); bilinear lookup (wrap in u, clamp in v); importance sampling proportional to the
    texel's 3x3-neighbourhood peak luminance x sin(theta) (piecewise constant), combined with
    phase-function sampling by the power heuristic (volpathsimple.py:273-278,391)."""
    pixels: object
    scale: float = 1.0
    to_world: Sequence[Sequence[float]] = ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))

    @property
    def resolution(self):
        h, w = self.pixels.shape[:2]
        return int(w), int(h)

    def to_world_flat(self):
        R = np.asarray(self.to_world, dtype=np.float32).reshape(3, 3)
        return [float(v) for v in R.reshape(-1)]

    @staticmethod
    def rotation_y(degrees: float):
        """`rotate(y, angle)` as the reference's scene files orient their environment maps."""
        a = np.deg2rad(degrees)
        c, s = float(np.cos(a)), float(np.sin(a))
        return ((c, 0.0, s), (0.0, 1.0, 0.0), (-s, 0.0, c))


@dataclass
class GridMedium:
    """`heterogeneous` medium with `gridvolume` sigma_t / albedo and an isotropic
    phase function, bounded by an axis-aligned box (tests/test_integrators.py:79-111).

    sigma_t : (Z, Y, X, 1) float32, albedo : (Z, Y, X, 3) float32 - numpy arrays or
    torch tensors (device tensors for the HIP path).
    """
    sigma_t: object
    albedo: object
    bbox_min: Sequence[float] = (0.0, 0.0, 0.0)
    bbox_max: Sequence[float] = (1.0, 1.0, 1.0)
    scale: float = 1.0
    # 0 = global majorant (Mitsuba default); the reference's optimisation scenes
    # use 8 (python/scene_config.py:36).
    majorant_resolution_factor: int = 0
    # (Z, Y, X, 3) emission grid, only read by the `nerf` integrator (medium.get_emission, nerf.py:164)
    emission: object = None

    @property
    def resolution(self):
        """(X, Y, Z)"""
        z, y, x = self.sigma_t.shape[:3]
        return (int(x), int(y), int(z))


@dataclass
class Scene:
    medium: GridMedium
    emitter: ConstantEmitter
    sensors: List[PerspectiveSensor] = field(default_factory=list)

    def params(self) -> Dict[str, object]:
        """The differentiable parameters, keyed like `mi.traverse(scene)`."""
        p = {SIGMA_T_KEY: self.medium.sigma_t}
        if self.medium.albedo is not None:
            p[ALBEDO_KEY] = self.medium.albedo
        if self.medium.emission is not None:
            p[EMISSION_KEY] = self.medium.emission
        return p


def cube_test_scene(resx: int = 128, resy: int = 128, density_scale: float = 1.0) -> Scene:
    """The fully specified 3x3x3 fixture of the reference's tests
    (tests/test_integrators.py:19-116): sigma_t and albedo grids, medium box
    [-0.5, 1.5]^3 (`translate(-0.5) * scale(2)` of the unit cube), camera at
    (4,4,4) looking at (0,-0.15,0), fov 30, constant emitter (1.0, 0.8, 0.2).
    The cube mesh with a null BSDF is replaced by the analytic box (comment at :106).
    """
    n = 3
    sigma_t = np.full((n, n, n, 1), 0.5, dtype=np.float32)
    sigma_t[0, 0, 0, 0] = 0.1
    sigma_t[0, n - 1, 0, 0] = 2.0
    sigma_t[0, 0, n - 1, 0] = 0.2
    base = np.ones((n, n, n, 3), dtype=np.float32) * np.array([0.3, 0.5, 0.9], dtype=np.float32)
    ramp = (np.arange(n, dtype=np.float32) + 1.0) / np.float32(n)
    base[..., 0] *= np.square(ramp)[:, None, None]
    base[..., 1] *= (1.0 - ramp)[:, None, None]
    base[..., 1] *= np.square(ramp)[None, :, None]
    albedo = np.clip(base, 0.0, 1.0).astype(np.float32)
    # the fixture's emission grid is the unclipped base (tests/test_integrators.py:27-37); here <= 1 anyway
    medium = GridMedium(sigma_t=sigma_t, albedo=albedo,
                        bbox_min=(-0.5, -0.5, -0.5), bbox_max=(1.5, 1.5, 1.5),
                        scale=density_scale, emission=base.astype(np.float32).copy())
    sensor = PerspectiveSensor(origin=(4.0, 4.0, 4.0), target=(0.0, -0.15, 0.0),
                               up=(0.0, 1.0, 0.0), fov=30.0, width=resx, height=resy)
    return Scene(medium=medium, emitter=ConstantEmitter((1.0, 0.8, 0.2)), sensors=[sensor])


def scene_to(scene: Scene, device) -> Scene:
    """Copy of `scene` whose parameter grids are contiguous float32 torch tensors on
    `device` (what `mi.load_dict` does for a cuda_ad_rgb variant)."""
    import torch
    m = scene.medium

    def conv(a):
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            return a.detach().to(device=device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)

    medium = GridMedium(sigma_t=conv(m.sigma_t), albedo=conv(m.albedo), bbox_min=tuple(m.bbox_min),
                        bbox_max=tuple(m.bbox_max), scale=m.scale,
                        majorant_resolution_factor=m.majorant_resolution_factor, emission=conv(m.emission))
    emitter = scene.emitter
    if isinstance(emitter, EnvmapEmitter):
        emitter = EnvmapEmitter(pixels=conv(emitter.pixels), scale=emitter.scale, to_world=emitter.to_world)
    return Scene(medium=medium, emitter=emitter, sensors=list(scene.sensors))
