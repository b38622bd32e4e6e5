"""Mitsuba `.vol` grid files (N3): the format `mi.VolumeGrid.write` produces for the reference's
checkpoints and warm starts (python/util.py:55-71, python/scene_config.py:123-141).

Layout [M3-ext] (Mitsuba 3 `VolumeGrid`; restated, not verifiable against a reference file here):
    bytes 'V','O','L', uint8 version = 3, int32 type = 1 (float32),
    int32 xres, yres, zres, int32 channels, 6 x float32 bbox (min xyz, max xyz),
    then xres*yres*zres*channels float32, x fastest, channels interleaved - i.e. a C-order array
    of shape (Z, Y, X, C), the layout of the parameter tensors.  Little endian.
"""
from __future__ import annotations

import struct

import numpy as np

_HEADER = struct.Struct("<3sBiiiii6f")


def write_vol(path: str, data, bbox_min=(0.0, 0.0, 0.0), bbox_max=(1.0, 1.0, 1.0)) -> None:
    """data: (Z, Y, X, C) or (Z, Y, X) array / torch tensor."""
    if hasattr(data, "detach"):
        data = data.detach().cpu().numpy()
    a = np.ascontiguousarray(data, dtype="<f4")
    if a.ndim == 3:
        a = a[..., None]
    if a.ndim != 4:
        raise ValueError(f"expected a (Z,Y,X,C) grid, got shape {a.shape}")
    z, y, x, c = a.shape
    with open(path, "wb") as f:
        f.write(_HEADER.pack(b"VOL", 3, 1, x, y, z, c, *map(float, bbox_min), *map(float, bbox_max)))
        f.write(a.tobytes())


def read_vol(path: str):
    """-> (data (Z,Y,X,C) float32, bbox_min, bbox_max)"""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise ValueError(f"{path}: truncated header")
        magic, version, dtype, x, y, z, c, *bbox = _HEADER.unpack(head)
        if magic != b"VOL" or version != 3:
            raise ValueError(f"{path}: not a Mitsuba VOL v3 file")
        if dtype != 1:
            raise ValueError(f"{path}: unsupported encoding {dtype} (only float32 = 1)")
        n = x * y * z * c
        raw = f.read(4 * n)
        if len(raw) != 4 * n:
            raise ValueError(f"{path}: expected {4 * n} data bytes, found {len(raw)}")
    data = np.frombuffer(raw, dtype="<f4").reshape(z, y, x, c).astype(np.float32)
    return data, tuple(bbox[:3]), tuple(bbox[3:])


def medium_from_vol(medium_filename: str, albedo_filename=None, emission_filename=None, albedo_value: float = 0.6,
                    scale: float = 1.0, majorant_resolution_factor: int = 0, device=None):
    """A `GridMedium` from `.vol` files - the `medium_filename` / `albedo_filename` / `emission_filename` variables of
    the reference's scene descriptions (python/scene_config.py:84-141), e.g. the checkpoints of a previous run for a
    warm start (`janga-smoke-from-nerf`: `<output>/<run>/nerf/params/final-medium1_sigma_t.vol`, :123-141).

    The bounding box is the one stored in the sigma_t file.  Without an albedo file the albedo is the constant
    `albedo_value` on sigma_t's lattice.  An albedo / emission file keeps ITS OWN resolution (the reference pairs a
    264x136x136 density with 256x128x128 albedo / emission grids, :108-110; Mitsuba interpolates every grid on its own
    lattice): the integrators then run the own-lattice kernels (drt_set_colour_resolution, csrc/drt_own.hip).  Albedo and
    emission files must share one lattice."""
    import torch
    from .scene import GridMedium
    sig, bmin, bmax = read_vol(medium_filename)
    if sig.shape[3] != 1:
        raise ValueError(f"{medium_filename}: expected a 1-channel density grid, found {sig.shape[3]} channels")
    res3 = sig.shape[:3]

    def load(path, what):
        if path is None:
            return None
        g, _, _ = read_vol(path)
        if g.shape[3] not in (1, 3):
            raise ValueError(f"{path}: expected a 1- or 3-channel {what} grid, found {g.shape[3]} channels")
        t = torch.from_numpy(g)
        if t.shape[3] == 1:
            t = t.expand(-1, -1, -1, 3)
        return t.contiguous()

    albedo = load(albedo_filename, "albedo")
    if albedo is None:
        albedo = torch.full(tuple(res3) + (3,), float(albedo_value), dtype=torch.float32)
    emission = load(emission_filename, "emission")
    if emission is not None and albedo_filename is not None and tuple(emission.shape[:3]) != tuple(albedo.shape[:3]):
        raise ValueError(f"{albedo_filename} and {emission_filename}: albedo and emission grids must share one lattice, "
                         f"found {tuple(albedo.shape[:3])} and {tuple(emission.shape[:3])}")
    if emission is not None and albedo_filename is None:                  # (a constant albedo follows the emission's lattice)
        albedo = torch.full(tuple(emission.shape[:3]) + (3,), float(albedo_value), dtype=torch.float32)
    to = (lambda t: t.to(device)) if device is not None else (lambda t: t)
    return GridMedium(sigma_t=to(torch.from_numpy(sig.copy())), albedo=to(albedo), bbox_min=bmin, bbox_max=bmax, scale=scale,
                      majorant_resolution_factor=majorant_resolution_factor,
                      emission=to(emission) if emission is not None else None)
