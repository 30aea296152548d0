"""Randomised differential run of the Cellpose dynamics (flows -> label image) against the oracle: flow fields of irregular
label images (thresholded smoothed noise: concave blobs, holes, specks, touching the border) with noise on the flows, perturbed
cell probabilities, random thresholds; several frames per call.  Bit-exact labels expected.
usage: python tests/fuzz/fuzz_dynamics.py [first_seed=0] [n=20]     (GPU box)"""
import sys
import time

import numpy as np
import torch
from scipy import ndimage as ndi

sys.path.insert(0, ".")
from aliby_amd import synth  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine  # noqa: E402
from aliby_amd.segment.dynamics import masks_from_flows  # noqa: E402
from oracle import cellpose_restated as cr  # noqa: E402

eng = FeatureEngine()
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
for seed in range(first, first + n):
    rng = np.random.default_rng(5000 + seed)
    Y, X = int(rng.integers(64, 200)), int(rng.integers(64, 240))
    F = int(rng.integers(1, 4))
    dPs, probs = [], []
    for f in range(F):
        field = ndi.gaussian_filter(rng.standard_normal((Y, X)), float(rng.uniform(2.0, 6.0)))
        lab, k = ndi.label(field > np.quantile(field, rng.uniform(0.5, 0.8)))
        if k == 0 or rng.random() < 0.1:
            lab = np.zeros((Y, X), np.int32)  # an empty frame now and then
        dP, prob = synth.analytic_flows(lab.astype(np.uint16))
        dP = dP + rng.normal(0, float(rng.uniform(0.0, 1.5)), dP.shape).astype(np.float32)
        prob = prob + rng.normal(0, float(rng.uniform(0.0, 2.0)), prob.shape).astype(np.float32)
        dPs.append(dP.astype(np.float32))
        probs.append(prob.astype(np.float32))
    kw = dict(cellprob_threshold=float(rng.choice([0.0, 0.0, -1.0, 1.5])), flow_threshold=float(rng.choice([0.4, 0.4, 0.0, 1.0])),
              min_size=int(rng.choice([15, 15, 1, 40])), niter=int(rng.choice([200, 200, 50])))
    t0 = time.perf_counter()
    labels, counts = masks_from_flows(eng, torch.from_numpy(np.stack(dPs)).cuda(), torch.from_numpy(np.stack(probs)).cuda(), **kw)
    got = labels.cpu().numpy()
    for f in range(F):
        want = cr.compute_masks(dPs[f], probs[f], **kw)
        assert got[f].max() == want.max() == counts[f], (seed, f, int(got[f].max()), int(want.max()), counts[f])
        assert np.array_equal(got[f], want), (seed, f, kw, int((got[f] != want).sum()))
    print(f"seed {seed}: {F} x {Y}x{X}, masks {[int(c) for c in counts]}, {kw}: ok ({time.perf_counter() - t0:.1f} s)", flush=True)
