"""N3: image files for reference renderings / previews (image_io.py; the reference writes EXR through mi.Bitmap,
python/optimize.py:50,131 - PFM here).  Bytes are checked against a file assembled by hand from the PFM definition."""
import struct

import numpy as np
import pytest


def test_pfm_bytes_match_hand_assembled_file(uivr, tmp_path):
    img = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3) / 7.0      # H = 2, W = 3
    path = str(tmp_path / "a.pfm")
    uivr.write_image(path, img)
    want = b"PF\n3 2\n-1.0\n"
    for row in (1, 0):                                                       # bottom row first
        for x in range(3):
            want += struct.pack("<3f", *img[row, x])
    assert open(path, "rb").read() == want
    back = uivr.read_image(path)
    assert back.dtype == np.float32 and back.shape == (2, 3, 3)
    np.testing.assert_array_equal(back, img)


def test_image_roundtrips_and_errors(uivr, tmp_path):
    import torch
    rng = np.random.default_rng(3)
    img = rng.standard_normal((5, 4, 3)).astype(np.float32)
    img[0, 0, 0] = np.inf
    for ext in (".pfm", ".npy"):
        p = str(tmp_path / ("b" + ext))
        uivr.write_image(p, torch.from_numpy(img))
        np.testing.assert_array_equal(uivr.read_image(p), img)
    grey = img[..., :1]
    p = str(tmp_path / "g.pfm")
    uivr.write_image(p, grey)
    assert open(p, "rb").read(3) == b"Pf\n"
    np.testing.assert_array_equal(uivr.read_image(p), grey)
    # a big-endian file (positive scale) reads the same
    be = str(tmp_path / "be.pfm")
    with open(be, "wb") as f:
        f.write(b"PF\n4 5\n1.0\n" + np.ascontiguousarray(img[::-1]).astype(">f4").tobytes())
    np.testing.assert_array_equal(uivr.read_image(be), img)
    with pytest.raises(NotImplementedError):
        uivr.write_image(str(tmp_path / "c.exr"), img)
    with pytest.raises(ValueError):
        uivr.write_image(str(tmp_path / "c.png"), img)
    with pytest.raises(ValueError):
        uivr.write_image(p, img[0])
    trunc = str(tmp_path / "t.pfm")
    with open(trunc, "wb") as f:
        f.write(b"PF\n4 5\n-1.0\n" + b"\0" * 10)
    with pytest.raises(ValueError):
        uivr.read_image(trunc)
    with open(trunc, "wb") as f:
        f.write(b"P6\n4 5\n255\n")
    with pytest.raises(ValueError):
        uivr.read_image(trunc)
