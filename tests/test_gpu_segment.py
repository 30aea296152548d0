"""
GPU parity of the Cellpose post-network dynamics (HIP) against the CPU restatement, bit-exact labels.
Flows are analytic (derived from synthetic ground truth): Cellpose weights are not obtainable offline
(SURVEY.md §8d), so the network itself is exercised separately with random weights.
"""

import numpy as np
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu


def _flows(shape, nt, fov, objset):
    f = synth.make_fov(1, fov, shape=shape, n_target=nt)
    gt = f[objset]
    dP, prob = synth.analytic_flows(gt)
    return gt, dP, prob


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
def test_dynamics_bit_exact_vs_oracle(engine, objset):
    import torch
    from aliby_amd.segment.dynamics import masks_from_flows
    from oracle import cellpose_restated as cr

    tiles = [_flows((256, 288), 26, fov, objset) for fov in (0, 1, 2)]
    # tile 2: add noise to the flows so that the flow-error QC actually removes something,
    # a speck below min_size, and a ring-shaped mask with a hole
    gt, dP, prob = tiles[2]
    rng = np.random.default_rng(3)
    bad = gt == 3
    dP = dP.copy()
    dP[:, bad] = rng.normal(0, 4, size=(2, int(bad.sum()))).astype(np.float32)
    prob = prob.copy()
    prob[5:8, 5:8] = 6.0
    tiles[2] = (gt, dP, prob)
    dPs = np.stack([t[1] for t in tiles])
    probs = np.stack([t[2] for t in tiles])
    labels, n, pf = masks_from_flows(engine, torch.from_numpy(dPs).cuda(), torch.from_numpy(probs).cuda(),
                                     return_endpoints=True)
    torch.cuda.synchronize()
    got = labels.cpu().numpy()
    pf = pf.cpu().numpy()
    for k, (gt, dP, prob) in enumerate(tiles):
        cp = prob > 0
        inds = np.nonzero(cp)
        p_ref = cr.follow_flows(((dP * cp) / np.float32(5.0)).astype(np.float32), inds, 200)
        assert np.array_equal(pf[k, 0][inds], p_ref[0]) and np.array_equal(pf[k, 1][inds], p_ref[1]), "end points differ"
        want = cr.compute_masks(dP, prob)
        assert got[k].max() == want.max() == n[k]
        assert np.array_equal(got[k], want), f"tile {k}: {int((got[k] != want).sum())} pixels differ"
    # the clean tiles recover the ground-truth partition
    gt = tiles[0][0]
    assert n[0] == gt.max()


def test_dynamics_edge_cases(engine):
    import torch
    from aliby_amd.segment.dynamics import masks_from_flows
    from oracle import cellpose_restated as cr

    Y, X = 96, 128
    # no foreground at all; everything foreground with zero flow (one big mask -> removed as > 40 %)
    dP = np.zeros((2, 2, Y, X), np.float32)
    prob = np.stack([np.full((Y, X), -6, np.float32), np.full((Y, X), 6, np.float32)])
    labels, n = masks_from_flows(engine, torch.from_numpy(dP).cuda(), torch.from_numpy(prob).cuda())
    got = labels.cpu().numpy()
    for k in range(2):
        want = cr.compute_masks(dP[k], prob[k])
        assert np.array_equal(got[k], want)
    assert n.tolist() == [0, int(got[1].max())]
