"""Randomised differential runs of the smaller kernels against the oracle: crop + median pad (random windows leaving the frame on
any side), the IoU stitch tracker over random label sequences (objects appearing, vanishing, splitting, label gaps), percentile
normalisation (narrow ranges, ties, constant images) and the object table (random labels with gaps).
usage: python tests/fuzz/fuzz_stager_tracker.py [first_seed=0] [n=50]     (GPU box)"""
import sys

import numpy as np
import torch
from scipy import ndimage as ndi

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr, to_device_u16  # noqa: E402
from aliby_amd.segment.cellpose_hip import CellposeModel  # noqa: E402
from aliby_amd.track.stitch import StitchTracker  # noqa: E402
from oracle import cellpose_restated as cr  # noqa: E402
from oracle import tiler_ref  # noqa: E402
from oracle.track_restated import stitch_rois as oracle_rois  # noqa: E402

eng = FeatureEngine()
model = CellposeModel(flows_override=lambda x: None)
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 50)


def random_labels(rng, shape, q):
    field = ndi.gaussian_filter(rng.standard_normal(shape), float(rng.uniform(1.5, 5.0)))
    lab, k = ndi.label(field > np.quantile(field, q))
    if k and rng.random() < 0.5:  # label gaps: drop a few ids without renumbering
        lab[np.isin(lab, rng.choice(np.arange(1, k + 1), max(1, k // 4), replace=False))] = 0
    return lab.astype(np.uint16)


for seed in range(first, first + n):
    rng = np.random.default_rng(9000 + seed)
    # ---- crop + median pad
    C, Z, Y, X = int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(40, 120)), int(rng.integers(40, 140))
    stack = rng.integers(0, 65535, size=(C, Z, Y, X), dtype=np.uint16)
    h, w = int(rng.integers(8, 64)), int(rng.integers(8, 64))
    F = int(rng.integers(1, 8))
    rects = np.stack([rng.integers(-h, Y, F), rng.integers(-w, X, F), np.full(F, h), np.full(F, w)], axis=1).astype(np.int32)
    out = torch.zeros((F, C, Z, h, w), dtype=torch.uint16, device="cuda")
    flags = np.zeros(F, np.int32)
    _lib.check(eng.lib.aliby_crop_pad_u16(eng.ctx.handle, _ptr(torch.from_numpy(stack).cuda()), C, Z, Y, X, _ptr(rects), F, h, w,
                                          _ptr(out), _ptr(flags), _stream_ptr()))
    got = out.cpu().numpy()
    for f, r in enumerate(rects):
        rg = (slice(int(r[0]), int(r[0]) + h), slice(int(r[1]), int(r[1]) + w))
        for c in range(C):
            try:
                want = tiler_ref.if_out_of_bounds_pad(stack[c], rg)
            except ValueError:
                # a window entirely outside the frame whose padding escapes the reference's NaN rule (its comparison pairs the
                # pads with (h, w) column-wise: a window with 4 w <= h lying wholly left of the frame): np.pad refuses an empty axis.
                # Unreachable through the tiler (square tiles); the kernel's answer for it is not compared.
                continue
            if np.isnan(want).any():
                assert flags[f] == 1, (seed, "pad flag", f, r)
            else:
                assert flags[f] == 0 and np.array_equal(got[f, c], want.astype(np.uint16)), (seed, "crop", f, c, r)
    # ---- tracker
    T, tiles = int(rng.integers(2, 5)), int(rng.integers(1, 4))
    shape = (int(rng.integers(24, 80)), int(rng.integers(24, 80)))
    seqs = [[random_labels(rng, shape, rng.uniform(0.5, 0.9)) if rng.random() > 0.1 else np.zeros(shape, np.uint16) for _ in range(T)]
            for _ in range(tiles)]
    thr = float(rng.choice([0.25, 0.25, 0.05, 0.6]))
    trk = StitchTracker(stitch_threshold=thr, engine=eng)
    info_g = info_c = None
    for t in range(1, T):
        masks = [[s[t - 1], s[t]] for s in seqs]
        info_g = trk(masks, info_g)
        info_c = oracle_rois(masks, info_c, stitch_threshold=thr)
        assert dict(info_g) == info_c, (seed, "track", t, thr)
    # ---- normalize99
    img = rng.integers(0, int(rng.choice([2, 50, 4000, 65535])), size=(int(rng.integers(1, 4)), Y, X)).astype(np.uint16)
    if rng.random() < 0.2:
        img[0] = int(rng.integers(0, 65535))
    norm = model.normalize(torch.from_numpy(img).cuda()).cpu().numpy()
    for k in range(img.shape[0]):
        assert np.array_equal(norm[k], cr.normalize99(img[k])), (seed, "normalize99", k)
    # ---- object table
    lab = random_labels(rng, (Y, X), rng.uniform(0.4, 0.9))
    tab = eng.object_table(to_device_u16(lab[None]))
    assert tab.n_obj == int(lab.max()), (seed, "object table size")  # (one row per label 1..max: an absent label is an empty row)
    sl = ndi.find_objects(lab.astype(np.int32))
    for row in tab.host:
        s = sl[int(row["label"]) - 1]
        assert int(row["area"]) == int((lab == row["label"]).sum()), (seed, "area")
        if s is not None:
            assert (int(row["y0"]), int(row["x0"]), int(row["y1"]), int(row["x1"])) == (s[0].start, s[1].start, s[0].stop, s[1].stop), (seed, "bbox")
    print(f"seed {seed}: ok", flush=True)
