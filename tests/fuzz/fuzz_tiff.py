"""Randomised round trip of the native TIFF reader (host decode path: no GPU needed): planes of random size and dtype written by the
minimal baseline writer (uncompressed / Deflate, random strip heights) and read back through `ImageMultiTiff`.
usage: python tests/fuzz/fuzz_tiff.py [first_seed=0] [n=300]"""
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, ".")
from aliby_amd import synth  # noqa: E402
from aliby_amd.io import image as im  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 300)
tmp = Path(tempfile.mkdtemp(prefix="aliby_fuzz_tiff_"))
for seed in range(first, first + n):
    rng = np.random.default_rng(31000 + seed)
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 400))
    dtype = rng.choice([np.uint16, np.uint16, np.uint8])
    kind = int(rng.integers(0, 3))
    if kind == 0:
        plane = rng.integers(0, np.iinfo(dtype).max + 1, (h, w)).astype(dtype)
    elif kind == 1:
        plane = np.full((h, w), int(rng.integers(0, 255)), dtype)  # (compresses to almost nothing)
    else:
        plane = (np.add.outer(np.arange(h), np.arange(w)) % 251).astype(dtype)
    path = tmp / f"p{seed}.tif"
    synth.write_tiff(path, plane, compression=("deflate" if rng.random() < 0.5 else None), rows_per_strip=int(rng.integers(1, h + 3)))
    img = im.ImageMultiTiff(path, capture_order="YX")
    got = np.asarray(img.data)
    assert got.shape[-2:] == (h, w) and got.dtype == plane.dtype, (seed, got.shape, got.dtype)
    assert np.array_equal(got.reshape(h, w), plane), (seed, "pixels")
    path.unlink()
print(f"{n} seeds ok")
