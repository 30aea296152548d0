"""
Writes the TIFF decoder's fixtures: small files produced by two independent encoders (tifffile 2021.7.2 + imagecodecs
2021.8.26 in the build container's conda interpreter, and Pillow's libtiff 4.7 binding in the system interpreter) next
to the arrays that went in (`expected.npz`).  The decoder under test (aliby_amd/csrc/ingest.hip) shares no code with either.

    /opt/conda/bin/python3.9 tests/golden/make_tiff_fixtures.py tifffile
    python tests/golden/make_tiff_fixtures.py pillow
"""
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).parent / "tiff"


def images():
    rng = np.random.default_rng(20260821)
    yy, xx = np.mgrid[0:40, 0:52]
    smooth = (400 + 30 * yy + 11 * xx).astype(np.uint16)
    noisy = (smooth + rng.integers(0, 900, smooth.shape)).astype(np.uint16)
    big = (rng.integers(0, 4096, (96, 128)) + 64 * np.mgrid[0:96, 0:128][0]).astype(np.uint16)
    return {
        "smooth": smooth,
        "noisy": noisy,
        "big": big,
        "bytes": (noisy >> 3).astype(np.uint8),
        "signed": (noisy.astype(np.int32) - 1500).astype(np.int16),
        "floats": (noisy / 7.0).astype(np.float32),
        "stack": np.stack([noisy + 100 * k for k in range(6)]).astype(np.uint16),
        "rgb": np.stack([(noisy >> 4), (noisy >> 5), (noisy >> 6)], -1).astype(np.uint8),
    }


def with_tifffile():
    import tifffile

    im = images()
    w = tifffile.imwrite
    w(OUT / "plain_strips7.tif", im["noisy"], rowsperstrip=7)
    w(OUT / "deflate_pred.tif", im["noisy"], compression="zlib", predictor=True, rowsperstrip=16)
    w(OUT / "deflate_pred_big.tif", im["big"], compression="zlib", predictor=True)
    w(OUT / "zstd_tiles16.tif", im["noisy"], compression="zstd", tile=(16, 16))
    w(OUT / "plain_tiles32x16.tif", im["noisy"], tile=(32, 16))
    w(OUT / "packbits_u8.tif", im["bytes"], compression="packbits")
    w(OUT / "bigendian_deflate_pred.tif", im["noisy"], byteorder=">", compression="zlib", predictor=True)
    w(OUT / "bigendian_plain.tif", im["noisy"], byteorder=">")
    w(OUT / "bigtiff_f32.tif", im["floats"], bigtiff=True)
    w(OUT / "signed_deflate.tif", im["signed"], compression="zlib")
    w(OUT / "pages6_deflate.tif", im["stack"], compression="zlib", photometric="minisblack")
    w(OUT / "imagej_t2z3.tif", im["stack"].reshape(2, 3, 1, 40, 52), imagej=True, metadata={"axes": "TZCYX"})
    w(OUT / "rgb_chunky.tif", im["rgb"], photometric="rgb")
    np.savez_compressed(OUT / "expected.npz", **im)


def with_pillow():
    from PIL import Image

    im = images()
    Image.fromarray(im["noisy"]).save(OUT / "pil_lzw.tif", compression="tiff_lzw")
    Image.fromarray(im["big"]).save(OUT / "pil_lzw_big.tif", compression="tiff_lzw")  # > 4096 codes: table resets
    Image.fromarray(im["big"]).save(OUT / "pil_lzw_pred_big.tif", compression="tiff_lzw", tiffinfo={317: 2})
    Image.fromarray(im["smooth"]).save(OUT / "pil_lzw_smooth.tif", compression="tiff_lzw")
    Image.fromarray(im["noisy"]).save(OUT / "pil_deflate.tif", compression="tiff_adobe_deflate")
    Image.fromarray(im["bytes"]).save(OUT / "pil_packbits_u8.tif", compression="packbits")


if __name__ == "__main__":
    OUT.mkdir(exist_ok=True)
    {"tifffile": with_tifffile, "pillow": with_pillow}[sys.argv[1]]()
