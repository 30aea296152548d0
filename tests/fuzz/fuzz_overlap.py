"""Randomised differential run of the overlapping-mask extraction (BABY-style [S,Y,X] stacks per tile with arbitrary original
labels, empty planes, label values colliding across planes) against the oracle's restatement of the reference's flow.
usage: python tests/fuzz/fuzz_overlap.py [first_seed=0] [n=40]     (GPU box)"""
import sys
from functools import partial

import numpy as np

sys.path.insert(0, ".")
from aliby_amd.extraction.extract import extract_tree, process_tree_masks_overlap  # noqa: E402
from oracle import aliby_extract as ox  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
for seed in range(first, first + n):
    rng = np.random.default_rng(17000 + seed)
    Y, X = int(rng.integers(48, 140)), int(rng.integers(48, 160))
    yy, xx = np.mgrid[0:Y, 0:X]
    tiles = int(rng.integers(1, 4))
    masks = []
    for _ in range(tiles):
        S = int(rng.integers(1, 5))
        stack = np.zeros((S, Y, X), np.int32)
        for s in range(S):
            if rng.random() < 0.2:
                continue  # an empty plane
            for _ in range(int(rng.integers(1, 5))):
                cy, cx, r = int(rng.integers(0, Y)), int(rng.integers(0, X)), int(rng.integers(2, 18))
                stack[s][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = int(rng.integers(1, 60))
        masks.append(stack)
    if not any(m.any() for m in masks):
        masks[0][0, 5:9, 5:9] = 3
    C, Z = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    pixels = rng.integers(100, 5000, size=(tiles, C, Z, Y, X)).astype(np.uint16)
    tree = {"None": {"None": ["sizeshape", "area"]}, 0: {"max": ["intensity", "mean"]}}
    if Z > 1:
        tree[C - 1] = {"add": ["intensity"]} if C > 1 else tree[0]
    inst, res = process_tree_masks_overlap(tree, masks, pixels, partial(extract_tree, overlap=True))
    inst_o, res_o, inv_o = ox.process_tree_masks_overlap(tree, masks, pixels)
    assert inst == inst_o, (seed, "instructions")
    assert {k: dict(v) for k, v in res.inverse_mappings.items()} == inv_o, (seed, "inverse mappings")
    assert len(res) == len(res_o)
    for i, (a, b) in enumerate(zip(res, res_o)):
        if isinstance(b, dict):
            for k in b:
                if k.endswith("Orientation"):
                    continue
                assert np.allclose(a[k], b[k], rtol=1e-4, atol=1e-8, equal_nan=True), (seed, inst[i], k, a[k], b[k])
        else:
            assert np.isclose(a, b, rtol=1e-4, equal_nan=True), (seed, inst[i], a, b)
    print(f"seed {seed}: {tiles} tiles of {[m.shape[0] for m in masks]} planes, {Y}x{X}, {len(inst)} results: ok", flush=True)
