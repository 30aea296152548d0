"""
Writes the Blosc decoder's fixtures (run in the build container's conda interpreter, which has imagecodecs 2021.8.26 linked
against c-blosc 1.21.0; the decoder under test, aliby_amd/csrc/ingest.hip, shares no code with it):

    /opt/conda/bin/python3.9 tests/golden/make_blosc_fixtures.py

  tests/golden/blosc/frames.npz     one Blosc frame per case (`f_<case>`, uint8) next to the bytes that went in (`x_<case>`)
  tests/golden/blosc/c5.zarr/       a zarr v2 group shaped like BASELINE config 5 (array "0": [T=1, C=2, Z=8, Y=128, X=160] uint16,
                                    one chunk per Z plane, Blosc lz4 + byte shuffle = zarr's default compressor) written by hand
                                    (the .zarray / .zgroup JSON is the layout zarr 2.x produces), and c5_expected.npz (pixels, nuclei) beside it
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
OUT = Path(__file__).parent / "blosc"


def main():
    import imagecodecs

    OUT.mkdir(exist_ok=True)
    rng = np.random.default_rng(20260821)
    yy, xx = np.mgrid[0:48, 0:64]
    smooth = (400 + 30 * yy + 11 * xx)
    data = {
        "u16_smooth": smooth.astype(np.uint16),
        "u16_noisy": (smooth + rng.integers(0, 900, smooth.shape)).astype(np.uint16),
        "u8": ((smooth >> 2) & 255).astype(np.uint8),
        "f32": (smooth / 7.0 + rng.normal(0, 1, smooth.shape)).astype(np.float32),
        "u16_random": rng.integers(0, 65536, (32, 48), dtype=np.uint16),      # incompressible: memcpyed frames / raw streams
        "u16_odd": (smooth.ravel()[:2345]).astype(np.uint16),                   # a leftover block that is not a multiple of 8 elements
        "labels": np.repeat(np.repeat(rng.integers(0, 40, (12, 16)), 8, 0), 8, 1).astype(np.uint16),  # long runs: far matches
        "empty": np.zeros(0, np.uint16),
    }
    frames = {}
    for name, arr in data.items():
        raw = arr.tobytes()
        for comp in ("blosclz", "lz4", "lz4hc", "zlib", "zstd"):
            for shuffle in (0, 1, 2):
                for blocksize in (None, 4096):
                    if blocksize is not None and (name not in ("u16_noisy", "u16_odd", "f32") or shuffle == 0 and comp != "lz4"):
                        continue
                    case = f"{name}__{comp}__s{shuffle}" + ("" if blocksize is None else f"__b{blocksize}")
                    kw = {} if blocksize is None else {"blocksize": blocksize}
                    enc = imagecodecs.blosc_encode(raw, level=5, compressor=comp, typesize=arr.dtype.itemsize, shuffle=shuffle, **kw)
                    assert imagecodecs.blosc_decode(enc) == raw
                    frames["f_" + case] = np.frombuffer(enc, np.uint8)
        frames["x_" + name] = np.frombuffer(raw, np.uint8)
    np.savez_compressed(OUT / "frames.npz", **frames)
    print(len([k for k in frames if k.startswith("f_")]), "frames,", sum(v.size for k, v in frames.items() if k.startswith("f_")), "bytes")

    # ---- a config-5-shaped zarr v2 group
    from aliby_amd import synth

    f = synth.make_fov(5, 0, shape=(128, 160), n_channels=2, n_z=8, n_target=12)
    px = f["pixels"][None]  # [T=1, C, Z, Y, X]
    root = OUT / "c5.zarr"
    (root / "0").mkdir(parents=True, exist_ok=True)
    (root / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
    (root / "0" / ".zarray").write_text(json.dumps({
        "chunks": [1, 1, 1, 128, 160], "compressor": {"blocksize": 0, "clevel": 5, "cname": "lz4", "id": "blosc", "shuffle": 1},
        "dtype": "<u2", "fill_value": 0, "filters": None, "order": "C", "shape": list(px.shape), "zarr_format": 2}, indent=4))
    for c in range(px.shape[1]):
        for z in range(px.shape[2]):
            enc = imagecodecs.blosc_encode(px[0, c, z].tobytes(), level=5, compressor="lz4", typesize=2, shuffle=1)
            (root / "0" / f"0.{c}.{z}.0.0").write_bytes(enc)
    np.savez_compressed(OUT / "c5_expected.npz", pixels=px, nuclei=f["nuclei"])
    print("zarr group:", sum(p.stat().st_size for p in (root / "0").iterdir()), "bytes")


if __name__ == "__main__":
    main()
