"""Randomised differential run of every feature family against the oracle on irregular label images: thresholded smoothed noise
(concave blobs, holes, one-pixel specks, objects on the frame border, very different sizes in one frame).
usage: python tests/fuzz/fuzz_features.py [first_seed=0] [n=12] [extras]     (GPU box; well under a second per seed)"""
import sys
import time

import numpy as np
from scipy import ndimage as ndi

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_edge_cases import _run_both  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
EXTRAS = len(sys.argv) > 3 and sys.argv[3] == "extras"  # other pixel types / reducers / per-feature keywords
MONO = ["intensity", "feret", "zernike", "radial_zernikes", "texture", "radial_distribution", "median", "max2p5pc", "mean", "std"]
for seed in range(first, first + n):
    rng = np.random.default_rng(1000 + seed)
    Y, X = int(rng.integers(96, 260)), int(rng.integers(96, 300))
    sigma = float(rng.uniform(1.5, 5.0))
    field = ndi.gaussian_filter(rng.standard_normal((Y, X)), sigma)
    lab, k = ndi.label(field > np.quantile(field, rng.uniform(0.55, 0.85)))
    # a few specks and a thin line, then sequential labels in raster order of first appearance (as a segmenter hands them on)
    for _ in range(3):
        y, x = int(rng.integers(0, Y)), int(rng.integers(0, X))
        if lab[y, x] == 0 and not lab[max(0, y - 1) : y + 2, max(0, x - 1) : x + 2].any():
            k += 1
            lab[y, x] = k
    if k > 60:  # (the oracle loops over objects x families: keep a seed within seconds)
        keep = rng.choice(np.arange(1, k + 1), 60, replace=False)
        lab = np.where(np.isin(lab, keep), lab, 0)
    _, inv = np.unique(lab, return_inverse=True)
    lab = inv.reshape(Y, X).astype(np.uint16)
    C, Z = 2, int(rng.integers(1, 3))
    px = rng.integers(0, 4000, size=(1, C, Z, Y, X)).astype(np.uint16)
    px += (ndi.gaussian_filter(rng.standard_normal((Y, X)), 3.0) * 6000 + 8000).clip(0, 40000).astype(np.uint16)[None, None, None]
    t0 = time.perf_counter()
    tree = {"None": {"None": ["sizeshape", "area", "volume"]}, 0: {"max": MONO}, 1: {"max": ["intensity", "texture"]}}
    if EXTRAS:
        # other pixel types, reducers and per-feature keywords, drawn per seed
        kind = ("u16", "f32", "u8")[seed % 3]
        if kind == "f32":
            px = (px.astype(np.float32) / np.float32(45000.0)).clip(0, 1)
        elif kind == "u8":
            px = (px >> 8).astype(np.uint8)
        kw = {"intensity": {"edge_measurements": bool(rng.integers(0, 2))},
              "texture": {"scale": int(rng.integers(1, 6)), "gray_levels": int(rng.choice([8, 64, 256]))},
              "radial_distribution": ({"bin_count": int(rng.integers(2, 7))} if rng.random() < 0.5
                                      else {"scaled": False, "bin_count": int(rng.integers(2, 6)), "maximum_radius": int(rng.integers(6, 40))}),
              "granularity": {"granular_spectrum_length": int(rng.integers(2, 7)), "subsample_size": float(rng.choice([0.25, 0.5, 1.0])),
                              "image_sample_size": float(rng.choice([0.25, 0.5])), "element_size": int(rng.integers(3, 11))},
              "manders_fold": {"thr": float(rng.integers(1, 60))}, "rwc": {"thr": float(rng.integers(1, 60))},
              "costes": {"scale_max": 255.0 if kind == "u8" else 65535.0} if rng.random() < 0.5 else {}}
        red = "add" if (Z > 1 and rng.random() < 0.5) else "max"
        tree = {"None": {"None": ["sizeshape"]}, 0: {red: ["intensity", "texture", "radial_distribution", "granularity", "radial_zernikes"]},
                1: {"max": ["texture", "median", "total", "max5px_median", "moment_of_inertia"]}}
        _run_both(tree, [lab], px, kw=kw)
        _run_both({(0, 1): {"None": {red: ["pearson", "costes", "manders_fold", "rwc"]}}}, [lab], px, multi=True, kw=kw)
        print(f"seed {seed}: {Y}x{X}, {int(lab.max())} objects, {kind}, {red}, {kw['texture']}, {kw['radial_distribution']}: ok "
              f"({time.perf_counter() - t0:.1f} s)", flush=True)
        continue
    _run_both(tree, [lab], px)
    _run_both({(0, 1): {"None": {"max": ["pearson", "costes", "manders_fold", "rwc"]}}}, [lab], px, multi=True)
    print(f"seed {seed}: {Y}x{X}, {int(lab.max())} objects, areas {np.bincount(lab.ravel())[1:].min()}..{np.bincount(lab.ravel())[1:].max()}, "
          f"sigma {sigma:.1f}: ok ({time.perf_counter() - t0:.1f} s)", flush=True)
