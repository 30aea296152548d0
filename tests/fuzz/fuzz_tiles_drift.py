"""Randomised differential runs against the oracle: network tiling + blending at random frame sizes / tile sizes / overlaps
(float32 bit-exact), and the drift estimate (phase cross-correlation) on randomly shifted, noisy frames.
usage: python tests/fuzz/fuzz_tiles_drift.py [first_seed=0] [n=40]     (GPU box)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402
from aliby_amd.segment.cellpose_hip import CellposeModel  # noqa: E402
from aliby_amd.tile.drift import phase_cross_correlation  # noqa: E402
from oracle import cellpose_restated as cr  # noqa: E402
from oracle.drift_restated import phase_cross_correlation as oracle_pcc  # noqa: E402

eng = FeatureEngine()
model = CellposeModel(flows_override=lambda x: None)
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
for seed in range(first, first + n):
    rng = np.random.default_rng(7000 + seed)
    F, Y, X = int(rng.integers(1, 4)), int(rng.integers(20, 420)), int(rng.integers(20, 520))
    bsize, overlap = int(rng.choice([64, 128, 224, 256])), float(rng.choice([0.1, 0.1, 0.25, 0.5]))
    g = model._geometry(Y, X, bsize, overlap)
    yp1, yp2, xp1, xp2 = cr.pad_to_16(Y, X)
    Ly, Lx = Y + yp1 + yp2, X + xp1 + xp2
    assert (g["ypad1"], g["xpad1"], g["Ly"], g["Lx"]) == (yp1, xp1, Ly, Lx), (seed, "padding")
    norm = rng.standard_normal((F, Y, X)).astype(np.float32)
    nt = F * g["ny"] * g["nx"]
    tiles = torch.empty((nt, 2, g["by"], g["bx"]), dtype=torch.float32, device="cuda")
    _lib.check(eng.lib.aliby_make_tiles(eng.ctx.handle, _ptr(torch.from_numpy(norm).cuda()), F, Y, X, g["ypad1"], g["xpad1"], g["Ly"],
                                        g["Lx"], g["by"], g["bx"], g["ny"], g["nx"], _ptr(g["ys"]), _ptr(g["xs"]), 2, _ptr(tiles),
                                        _stream_ptr()))
    got = tiles.cpu().numpy().reshape(F, g["ny"] * g["nx"], 2, g["by"], g["bx"])
    ys_o = xs_o = None
    for k in range(F):
        padded = np.zeros((2, Ly, Lx), np.float32)
        padded[0, yp1 : yp1 + Y, xp1 : xp1 + X] = norm[k]
        want, ys_o, xs_o = cr.make_tiles(padded, bsize, overlap)
        assert np.array_equal(got[k], want), (seed, "tiles", k)
    yt = rng.standard_normal((nt, 3, g["by"], g["bx"])).astype(np.float32)
    dP = torch.empty((F, 2, Y, X), dtype=torch.float32, device="cuda")
    prob = torch.empty((F, Y, X), dtype=torch.float32, device="cuda")
    _lib.check(eng.lib.aliby_average_tiles(eng.ctx.handle, _ptr(torch.from_numpy(yt).cuda()), F, Y, X, g["ypad1"], g["xpad1"], g["Ly"],
                                           g["Lx"], g["by"], g["bx"], g["ny"], g["nx"], _ptr(g["ys"]), _ptr(g["xs"]),
                                           _ptr(g["taper"]), _ptr(dP), _ptr(prob), _stream_ptr()))
    dP, prob = dP.cpu().numpy(), prob.cpu().numpy()
    per = g["ny"] * g["nx"]
    for k in range(F):
        full = cr.average_tiles(yt[k * per : (k + 1) * per], ys_o, xs_o, Ly, Lx)
        crop = full[:, yp1 : yp1 + Y, xp1 : xp1 + X]
        assert np.array_equal(dP[k], crop[:2]) and np.array_equal(prob[k], crop[2]), (seed, "blend", k, bsize, overlap, Y, X)
    # ---- drift
    H, W = int(rng.integers(32, 200)), int(rng.integers(32, 200))
    ref = (rng.integers(0, 3000, (H, W)) + 2000 * (rng.random((H, W)) > 0.97)).astype(np.uint16)
    sh = (int(rng.integers(-H // 3, H // 3 + 1)), int(rng.integers(-W // 3, W // 3 + 1)))
    mov = (np.roll(ref, sh, axis=(0, 1)).astype(np.int64) + rng.integers(0, 200, (H, W))).clip(0, 65535).astype(np.uint16)
    a, b = phase_cross_correlation(ref, mov), oracle_pcc(ref, mov)
    assert a.tolist() == b.tolist(), (seed, "drift", sh, a.tolist(), b.tolist())
    print(f"seed {seed}: {F} x {Y}x{X} tiles {bsize}/{overlap} ({g['ny']}x{g['nx']}), drift {sh}: ok", flush=True)
