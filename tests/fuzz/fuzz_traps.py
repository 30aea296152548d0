"""Randomised differential run of trap detection (`segment_traps`: the reference's scikit-image chain, one float64 kernel per
call) against the CPU restatement: trap grids of random spacing / jitter / frame size, several tile sizes.
usage: python tests/fuzz/fuzz_traps.py [first_seed=0] [n=20]     (GPU box; the restatement takes about a second per frame)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aliby_amd import synth  # noqa: E402
from aliby_amd.tile.traps import segment_traps  # noqa: E402
from oracle.traps_restated import segment_traps as oracle_traps  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
for seed in range(first, first + n):
    rng = np.random.default_rng(23000 + seed)
    shape = (int(rng.integers(300, 640)), int(rng.integers(300, 640)))
    spacing, jitter = int(rng.integers(100, 160)), int(rng.integers(0, 10))
    img, centres = synth.trap_image(seed=500 + seed, shape=shape, spacing=spacing, first=int(rng.integers(70, 110)), jitter=jitter)
    tile = int(rng.choice([117, 117, 96, 128]))
    t0 = time.perf_counter()
    def run(fn):  # (a frame without a usable template makes the reference raise: the error is part of the behaviour compared)
        try:
            return [tuple(int(v) for v in c) for c in fn(img, tile)]
        except Exception as e:  # noqa: BLE001
            return [("raised", type(e).__name__, str(e))]

    got, want = run(segment_traps), run(oracle_traps)
    assert got == want, (seed, shape, spacing, tile, got[:4], want[:4])
    print(f"seed {seed}: {shape}, spacing {spacing} +- {jitter}, tile {tile}: {len(got)} traps of {len(centres)} drawn: ok "
          f"({time.perf_counter() - t0:.1f} s)", flush=True)
