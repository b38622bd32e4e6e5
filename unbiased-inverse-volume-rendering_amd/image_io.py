"""Image files for reference renderings and previews (N3).

The reference writes OpenEXR through `mi.Bitmap(...).write` (python/optimize.py:50, :131); neither OpenEXR nor
Mitsuba exists here, so images are stored as **PFM** (Portable Float Map: the header `PF\\n<width> <height>\\n-1.0\\n`
followed by width*height*3 little-endian float32, rows bottom to top - readable by every HDR viewer) or as `.npy`.
Same precision as a float32 EXR, no compression.
"""
from __future__ import annotations

import os

import numpy as np


def _to_numpy(img) -> np.ndarray:
    if hasattr(img, "detach"):
        img = img.detach().cpu().numpy()
    a = np.asarray(img, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] not in (1, 3):
        raise ValueError(f"expected an (H, W, 3) or (H, W, 1) image, got shape {a.shape}")
    return a


def write_image(path: str, img) -> None:
    """`img`: (H, W, 3) or (H, W, 1) float array / tensor, row 0 = top row (the film's layout)."""
    a = _to_numpy(img)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        np.save(path, a)
    elif ext == ".pfm":
        h, w, c = a.shape
        with open(path, "wb") as f:
            f.write(("PF" if c == 3 else "Pf").encode("ascii") + b"\n" + f"{w} {h}\n".encode("ascii") + b"-1.0\n")
            f.write(np.ascontiguousarray(a[::-1], dtype="<f4").tobytes())       # PFM stores the bottom row first
    elif ext == ".exr":
        raise NotImplementedError("OpenEXR is not available in this build: use .pfm or .npy")
    else:
        raise ValueError(f"unsupported image extension '{ext}' (.pfm, .npy)")


def read_image(path: str) -> np.ndarray:
    """-> (H, W, C) float32, row 0 = top row."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return _to_numpy(np.load(path))
    if ext != ".pfm":
        raise ValueError(f"unsupported image extension '{ext}' (.pfm, .npy)")
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic not in (b"PF", b"Pf"):
            raise ValueError(f"{path}: not a PFM file")
        c = 3 if magic == b"PF" else 1
        dims = f.readline().split()
        if len(dims) != 2:
            raise ValueError(f"{path}: bad PFM dimensions line")
        w, h = int(dims[0]), int(dims[1])
        scale = float(f.readline().strip())
        raw = f.read(4 * w * h * c)
        if len(raw) != 4 * w * h * c:
            raise ValueError(f"{path}: expected {4 * w * h * c} data bytes, found {len(raw)}")
    a = np.frombuffer(raw, dtype="<f4" if scale < 0 else ">f4").reshape(h, w, c)
    return np.ascontiguousarray(a[::-1]).astype(np.float32)
