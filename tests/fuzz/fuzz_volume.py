"""Randomised differential run of the volume extension (per-plane labels stitched along Z, 3-D intensity block) against its CPU
restatement: random blob stacks whose cross-sections appear, vanish, split and merge from plane to plane, several thresholds.
usage: python tests/fuzz/fuzz_volume.py [first_seed=0] [n=40]     (GPU box)"""
import sys

import numpy as np
import torch
from scipy import ndimage as ndi

sys.path.insert(0, ".")
from aliby_amd.extraction.engine import FeatureEngine  # noqa: E402
from oracle import volume_restated as vr  # noqa: E402

eng = FeatureEngine()
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
for seed in range(first, first + n):
    rng = np.random.default_rng(19000 + seed)
    F, Z, Y, X = int(rng.integers(1, 3)), int(rng.integers(2, 8)), int(rng.integers(32, 100)), int(rng.integers(32, 120))
    planes = np.zeros((F, Z, Y, X), np.uint16)
    for f in range(F):
        field = ndi.gaussian_filter(rng.standard_normal((Z, Y, X)), (float(rng.uniform(0.5, 2.0)), 3.0, 3.0))
        for z in range(Z):
            if rng.random() < 0.1:
                continue  # an empty plane
            lab, _ = ndi.label(field[z] > np.quantile(field, rng.uniform(0.6, 0.85)))
            planes[f, z] = lab
    thr = float(rng.choice([0.01, 0.01, 0.25, 0.6]))
    vol, counts = eng.stitch_planes(torch.from_numpy(planes).cuda(), threshold=thr)
    got = vol.cpu().numpy()
    for f in range(F):
        want, k = vr.stitch3d(planes[f], thr)
        assert int(counts[f]) == k and np.array_equal(got[f], want), (seed, f, thr, int(counts[f]), k)
    C = int(rng.integers(1, 3))
    px = rng.integers(0, 60000, size=(F, C, Z, Y, X)).astype(np.uint16)
    for c in range(C):
        feats = eng.intensity3d(vol, torch.from_numpy(px).cuda(), c, [int(v) for v in counts]).cpu().numpy()
        want = np.concatenate([vr.intensity3d(got[f], px[f, c]) for f in range(F)]) if sum(int(v) for v in counts) else np.zeros((0, 12))
        assert feats.shape == want.shape and np.allclose(feats, want, rtol=1e-10, atol=1e-9, equal_nan=True), (seed, "intensity3d", c)
    print(f"seed {seed}: {F} stacks of {Z} x {Y}x{X}, threshold {thr}, objects {[int(v) for v in counts]}: ok", flush=True)
