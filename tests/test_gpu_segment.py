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
    # the counts are the frames' largest labels: handed to the object table they save its own pass (aliby_amd/runner.py does)
    t_own, t_known = engine.object_table(labels), engine.object_table(labels, max_labels=n)
    assert np.array_equal(t_own.host, t_known.host) and np.array_equal(t_own.offsets, t_known.offsets)
    assert torch.equal(t_own.dev[: 32 * t_own.n_obj], t_known.dev[: 32 * t_known.n_obj])
    with pytest.raises(ValueError, match="one non-negative count per frame"):
        engine.object_table(labels, max_labels=n[:2])
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


@pytest.mark.parametrize("bsize,overlap", [(224, 0.1), (256, 0.1), (128, 0.25)])
def test_normalize_tiles_blend_bit_exact(engine, bsize, overlap):
    """normalize99 / make_tiles / average_tiles kernels vs the NumPy restatement (float32 bit-exact), at cellpose 3's tile
    geometry (224, 0.1), cellpose 4's eval default (256) and an odd one: `eval(bsize=..., tile_overlap=...)` per call."""
    import torch
    from aliby_amd.segment.cellpose_hip import CellposeModel, pad_amounts, taper_mask, tile_starts
    from oracle import cellpose_restated as cr

    f = synth.make_fov(2, 0, shape=(300, 520), n_channels=1, n_target=30)
    imgs = np.stack([f["pixels"][0, 0], np.full((300, 520), 7, np.uint16), f["pixels"][0, 0][::-1].copy()])
    model = CellposeModel(flows_override=lambda x: None)
    dev = torch.from_numpy(imgs).cuda()
    norm = model.normalize(dev).cpu().numpy()
    for k in range(3):
        want = cr.normalize99(imgs[k])
        assert np.array_equal(norm[k], want), k
    assert (norm[1] == 0).all()  # constant image -> zeros
    # geometry helpers agree with the restatement
    yp1, yp2, xp1, xp2 = pad_amounts(300, 520)
    assert (yp1, yp2, xp1, xp2) == cr.pad_to_16(300, 520)
    Ly, Lx = 300 + yp1 + yp2, 520 + xp1 + xp2
    ys, by = tile_starts(Ly, bsize, overlap)
    xs, bx = tile_starts(Lx, bsize, overlap)
    ys_o, by_o = cr.tile_starts(Ly, bsize, overlap)
    xs_o, bx_o = cr.tile_starts(Lx, bsize, overlap)
    assert ys.tolist() == ys_o.tolist() and xs.tolist() == xs_o.tolist() and (by, bx) == (by_o, bx_o)
    assert np.array_equal(taper_mask(bsize, bsize), cr.taper_mask(bsize))
    assert np.array_equal(taper_mask(216, 256), cr.taper_mask(216, 256)) and np.array_equal(taper_mask(176, 208), cr.taper_mask(176, 208))
    # tiles
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    g = model._geometry(300, 520, bsize, overlap)
    assert (g["by"], g["bx"]) == (bsize, bsize)
    F = 3
    nt = F * g["ny"] * g["nx"]
    tiles = torch.empty((nt, 2, g["by"], g["bx"]), dtype=torch.float32, device="cuda")
    dn = torch.from_numpy(norm).cuda()
    _lib.check(engine.lib.aliby_make_tiles(engine.ctx.handle, _ptr(dn), F, 300, 520, g["ypad1"], g["xpad1"], g["Ly"], g["Lx"],
                                           g["by"], g["bx"], g["ny"], g["nx"], _ptr(g["ys"]), _ptr(g["xs"]), 2, _ptr(tiles),
                                           _stream_ptr()))
    got = tiles.cpu().numpy().reshape(F, g["ny"] * g["nx"], 2, g["by"], g["bx"])
    for k in range(F):
        padded = np.zeros((2, Ly, Lx), np.float32)
        padded[0, yp1 : yp1 + 300, xp1 : xp1 + 520] = norm[k]
        want, _, _ = cr.make_tiles(padded, bsize, overlap)
        assert np.array_equal(got[k], want)
    # blending of arbitrary network outputs
    rng = np.random.default_rng(0)
    yt = rng.standard_normal((nt, 3, g["by"], g["bx"])).astype(np.float32)
    dP = torch.empty((F, 2, 300, 520), dtype=torch.float32, device="cuda")
    prob = torch.empty((F, 300, 520), dtype=torch.float32, device="cuda")
    dyt = torch.from_numpy(yt).cuda()
    _lib.check(engine.lib.aliby_average_tiles(engine.ctx.handle, _ptr(dyt), F, 300, 520, g["ypad1"], g["xpad1"], g["Ly"],
                                              g["Lx"], g["by"], g["bx"], g["ny"], g["nx"], _ptr(g["ys"]), _ptr(g["xs"]),
                                              _ptr(g["taper"]), _ptr(dP), _ptr(prob), _stream_ptr()))
    dP, prob = dP.cpu().numpy(), prob.cpu().numpy()
    per = g["ny"] * g["nx"]
    for k in range(F):
        full = cr.average_tiles(yt[k * per : (k + 1) * per], ys_o, xs_o, Ly, Lx)
        crop = full[:, yp1 : yp1 + 300, xp1 : xp1 + 520]
        assert np.array_equal(dP[k], crop[:2]) and np.array_equal(prob[k], crop[2])


def test_network_forward_runs_and_is_deterministic(engine):
    """Random-weight U-Net: shapes, determinism and finite outputs (weights are not obtainable offline)."""
    import torch
    from aliby_amd.segment.cellpose_hip import CellposeModel

    f = synth.make_fov(1, 0, shape=(256, 256), n_channels=1, n_target=20)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CellposeModel(seed=3)
        # the default is cellpose 4's (use_bfloat16=True): the hand-written MFMA network, not PyTorch's convolutions
        assert model.fused is not None and model.net_dtype == torch.bfloat16
        assert CellposeModel(seed=3, use_bfloat16=False).fused is None and CellposeModel(seed=3, net_dtype="float32").fused is None
    x = torch.from_numpy(f["pixels"][0]).cuda()  # [1,Y,X]
    dP1, p1 = model.run_network(x)
    dP2, p2 = model.run_network(x)
    assert dP1.shape == (1, 2, 256, 256) and p1.shape == (1, 256, 256)
    assert torch.isfinite(dP1).all() and torch.isfinite(p1).all()
    # every kernel of the forward is deterministic (fixed tile->workgroup map, no atomics): the tolerance is slack, not need
    assert torch.allclose(dP1, dP2, atol=1e-4) and torch.allclose(p1, p2, atol=1e-4)
    assert model.net.flops_per_pixel() > 1e5


def test_segment_step_matches_reference_closure_semantics(engine):
    """dispatch_segmenter(kind='cellpose') closure: channel select, Z max-projection, uint16 [Y,X] result
    (segment/dispatch.py:179-234), labels bit-exact vs the oracle."""
    import torch
    from aliby_amd.segment.dispatch import dispatch_segmenter
    from oracle import cellpose_restated as cr

    f = synth.make_fov(4, 1, shape=(200, 240), n_channels=2, n_z=3, n_target=14)
    gt = f["nuclei"]
    dP, prob = synth.analytic_flows(gt)
    seen = {}

    def override(x):
        seen["plane"] = x.cpu().numpy()
        return torch.from_numpy(dP[None]).cuda(), torch.from_numpy(prob[None]).cuda()

    segment = dispatch_segmenter(kind="cellpose", channel_to_segment=1, setup_params=dict(flows_override=override))
    pixels = f["pixels"][None]  # [F=1,C,Z,Y,X]
    labels = segment(pixels)
    assert labels.dtype == np.uint16 and labels.shape == (200, 240)
    assert np.array_equal(seen["plane"][0], cr.select_and_project(pixels, 1)[0])
    want = cr.finish_labels(cr.compute_masks(dP, prob))
    assert np.array_equal(labels, want)
    with pytest.raises(Exception, match="Invalid segmentation method"):
        dispatch_segmenter(kind="nope", channel_to_segment=0)


def test_fused_unet_matches_module_forward(engine):
    """FusedUNet (the hand-written bf16 network: MFMA conv units on every level by default; with mfma_levels=() the A/B fallback of
    library convolutions + the fused pointwise kernel) vs the plain fp32 module forward."""
    import torch
    from aliby_amd.segment.fused_unet import FusedUNet
    from aliby_amd.segment.unet import build_network

    net = build_network(seed=5, device="cuda")
    # make BatchNorm statistics and affine parameters non-trivial so that every folded term is exercised
    g = torch.Generator(device="cpu").manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    x = torch.randn(6, 2, 224, 224, generator=g).cuda().contiguous()
    with torch.no_grad():
        y_ref, s_ref = net(x)
    for levels in ((0, 1, 2, 3), (0, 1), ()):  # MFMA conv unit on all levels (default) | levels 0-1 | MIOpen convs + fused pointwise
        fused = FusedUNet(net, engine, mfma_levels=levels)
        y, s = fused(x)
        assert y.shape == y_ref.shape == (6, 3, 224, 224) and y.dtype == torch.float32
        err = (y - y_ref).norm() / y_ref.norm()
        assert err < 0.03, (levels, float(err))
        assert ((s - s_ref).norm() / s_ref.norm()) < 0.03
        print("fused unet", levels, "rel l2 err vs fp32 module:", float(err))
    # deep levels: the K-loop unit (fp32 accumulation over the whole 9*CIN reduction, csrc/nn_conv_deep.hip) against round 1's
    # K/N-slice launches, whose partial sums were rounded to bf16 in HBM between slices: the error against the fp32 module
    # must not be larger, and both numbers are recorded for DESIGN.md
    kloop, ksplit = FusedUNet(net, engine), FusedUNet(net, engine)
    ksplit.deep_kernel = False
    assert kloop.deep_kernel
    errs = {}
    for name, f in (("k_loop", kloop), ("k_split", ksplit)):
        y, _ = f(x)
        errs[name] = float((y - y_ref).norm() / y_ref.norm())
        errs[name + "_max_abs"] = float((y - y_ref).abs().max())
    print("deep levels, rel l2 err vs fp32 module:", errs)
    assert errs["k_loop"] <= errs["k_split"] * 1.02, errs
    try:
        import json, pathlib

        pathlib.Path("gpurun_out").mkdir(exist_ok=True)
        pathlib.Path("gpurun_out/unet_error_kloop_vs_ksplit.json").write_text(json.dumps(errs))
    except OSError:
        pass
    # the output head in the last unit's epilogue gives the bits of the separate head kernel
    with_head = FusedUNet(net, engine)
    separate = FusedUNet(net, engine)
    separate.fused_head = False
    assert with_head.fused_head
    ya, _ = with_head(x)
    yb, _ = separate(x)
    assert torch.equal(ya, yb)
    # ... and so does running conv2 + conv3 of the level-0 blocks as one wave-specialised launch or as two launches
    unpaired = FusedUNet(net, engine)
    unpaired.fused_pair = False
    assert with_head.fused_pair
    yc, sc_ = unpaired(x)
    assert torch.equal(ya, yc)
    unpaired.fused_head = False
    yd, _ = unpaired(x)
    assert torch.equal(ya, yd)


@pytest.mark.parametrize("n,h,w", [(4, 256, 256), (3, 128, 128), (2, 176, 208), (5, 64, 96), (2, 32, 32), (2, 16, 16), (7, 24, 40)])
def test_fused_unet_at_other_tile_sizes(engine, n, h, w):
    """The network at tile sizes other than cellpose 3's 224 x 224 — `eval(bsize=256)` (cellpose 4's default), small images
    (one tile of the image's own padded size, a multiple of 16), odd batch sizes: the packed / tall / paired launch forms are
    chosen per shape, so every choice is checked against the fp32 module forward."""
    import torch
    from aliby_amd.segment.fused_unet import FusedUNet
    from aliby_amd.segment.unet import build_network

    net = build_network(seed=6, device="cuda")
    g = torch.Generator(device="cpu").manual_seed(2)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    x = torch.randn(n, 2, h, w, generator=g).cuda().contiguous()
    # (the fp32 module is the yardstick, not a benchmark: without this, MIOpen searches its kernels anew for every shape — 20 s each —
    # once an earlier test's CellposeModel has switched the benchmark mode on)
    bench_mode, torch.backends.cudnn.benchmark = torch.backends.cudnn.benchmark, False
    try:
        with torch.no_grad():
            y_ref, s_ref = net(x)
    finally:
        torch.backends.cudnn.benchmark = bench_mode
    y, s = FusedUNet(net, engine)(x)
    assert y.shape == y_ref.shape == (n, 3, h, w)
    err = float((y - y_ref).norm() / y_ref.norm())
    assert err < 0.03, err
    assert float((s - s_ref).norm() / s_ref.norm()) < 0.03


def test_dynamics_large_mask_uses_global_scratch(engine):
    """A mask whose padded bounding box (13k cells x 17 bytes = 224 KB) cannot live in LDS: the heat-diffusion QC and the hole
    filling take their grid-strided global-scratch variants; labels stay bit-exact against the CPU restatement."""
    import torch
    from aliby_amd.segment.dynamics import masks_from_flows
    from oracle import cellpose_restated as cr

    Y, X = 288, 320
    yy, xx = np.mgrid[0:Y, 0:X]
    gt = np.zeros((Y, X), np.uint16)
    gt[((yy - 140) / 52.0) ** 2 + ((xx - 160) / 60.0) ** 2 <= 1.0] = 1     # ~9.8k pixels (under the 40 % size cut), box 105 x 121
    hole = ((yy - 118) / 4.0) ** 2 + ((xx - 190) / 5.0) ** 2 <= 1.0        # an off-centre hole for the fill stage
    gt[270:282, 290:312] = 2
    dP, prob = synth.analytic_flows(gt)
    prob = prob.copy()
    prob[hole] = -6.0  # background inside the big mask: not followed, filled back at the end
    labels, n, _ = masks_from_flows(engine, torch.from_numpy(dP[None]).cuda(), torch.from_numpy(prob[None]).cuda(), niter=200,
                                    return_endpoints=True)
    torch.cuda.synchronize()
    want = cr.finish_labels(cr.compute_masks(dP, prob)) if hasattr(cr, "finish_labels") else cr.compute_masks(dP, prob)
    got = labels.cpu().numpy()[0]
    assert int(got.max()) == int(want.max()) == int(n[0]) == 2
    assert np.array_equal(got, want), f"{int((got != want).sum())} pixels differ"


def test_network_graph_replay_matches_eager(engine):
    """The network forward is captured into a hipGraph per batch shape and replayed (product path, no event timing);
    capture + two replays give exactly the eager result."""
    import warnings

    import torch
    from aliby_amd.segment.cellpose_hip import CellposeModel

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CellposeModel(net_dtype="bfloat16", seed=3, batch_size=5)
    f = synth.make_fov(1, 4, shape=(300, 420), n_target=12)
    img = torch.from_numpy(f["pixels"][0, 0][None]).cuda()  # [1,Y,X] uint16 -> 6 tiles: one batch of 5 + one of 1
    model.use_graph = False
    dP0, pr0 = model.run_network(img)
    model.use_graph = True
    outs = [model.run_network(img) for _ in range(3)]  # capture, replay, replay
    assert model.use_graph and len(model._graphs) == 2, "graph capture was refused"
    for dP, pr in outs:
        assert torch.equal(dP, dP0) and torch.equal(pr, pr0)


def test_masks_are_stable_under_the_networks_numeric_error(engine):
    """north_star asks for bit-exact object IDs; with random weights (no checkpoint is obtainable offline) the network's own
    output is noise, so its effect on masks cannot be measured directly.  A proxy that can: perturb analytic flow fields by the
    relative error the bf16 network shows against the fp32 module (0.0096 relative L2, test_fused_unet_matches_module_forward;
    3x that as a margin) and measure how the masks move.  Recorded for DESIGN.md §4."""
    import json
    import pathlib

    import torch
    from aliby_amd.segment.dynamics import masks_from_flows

    tiles = [_flows((512, 512), 64, fov, "nuclei") for fov in (10, 11, 12, 13)]
    dP = np.stack([t[1] for t in tiles])
    prob = np.stack([t[2] for t in tiles])
    base, n0 = masks_from_flows(engine, torch.from_numpy(dP).cuda(), torch.from_numpy(prob).cuda())
    base = base.cpu().numpy()
    rng = np.random.default_rng(11)
    report = {}
    for rel in (0.0096, 0.03):
        def noisy(a):
            rms = float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
            return (a + rng.normal(0.0, rel * rms, size=a.shape)).astype(np.float32)

        got, n1 = masks_from_flows(engine, torch.from_numpy(noisy(dP)).cuda(), torch.from_numpy(noisy(prob)).cuda())
        got = got.cpu().numpy()
        same_count = int(sum(int(a) == int(b) for a, b in zip(n0, n1)))
        ious, identical, total = [], 0, 0
        for f in range(len(tiles)):
            for lab in range(1, int(n0[f]) + 1):
                m = base[f] == lab
                other = np.bincount(got[f][m]).argmax()
                inter = np.logical_and(m, got[f] == other).sum() if other else 0
                union = np.logical_or(m, got[f] == other).sum() if other else m.sum()
                ious.append(inter / union)
                identical += int(other != 0 and inter == union)
                total += 1
        report[str(rel)] = dict(tiles_with_equal_object_count=same_count, objects=total, identical_masks=identical,
                                mean_iou=float(np.mean(ious)), min_iou=float(np.min(ious)),
                                pixels_changed=int((((base > 0) != (got > 0))).sum()), pixels=int(base.size))
        assert same_count == len(tiles), report
        assert np.mean(ious) > 0.97 and np.min(ious) > 0.7, report
    print("mask stability under flow perturbation:", report)
    try:
        pathlib.Path("gpurun_out").mkdir(exist_ok=True)
        pathlib.Path("gpurun_out/mask_stability.json").write_text(json.dumps(report, indent=1))
    except OSError:
        pass


def test_network_with_fused_first_pair_matches_the_separate_launches(engine):
    """Round 3: the first layer + second unit + projection of the network run as one launch (FusedUNet.fused_first); the whole
    forward gives exactly the bits of the path with the separate first-layer launch."""
    import warnings

    import torch
    from aliby_amd.segment.cellpose_hip import CellposeModel

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CellposeModel(net_dtype="bfloat16", seed=5, batch_size=4)
    f = synth.make_fov(1, 9, shape=(300, 420), n_target=12)
    img = torch.from_numpy(f["pixels"][0, 0][None]).cuda()
    assert model.fused is not None and model.fused.fused_first
    got = model.run_network(img)
    model.fused.fused_first = False
    want = model.run_network(img)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_network_last_block_as_pair_with_head_matches_the_two_launches(engine):
    """Round 3: the output head is two MFMA k-steps in all of its forms, and the last block (conv2 + conv3 + head) runs as the
    wave-specialised pair (ALIBY_NET_PAIR_HEAD=1; a tie in time, so not the default); the forward gives exactly the bits of the conv2 launch followed by the unit-with-head launch."""
    import warnings

    import torch
    from aliby_amd.segment.cellpose_hip import CellposeModel

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CellposeModel(net_dtype="bfloat16", seed=6, batch_size=4)
    f = synth.make_fov(1, 11, shape=(300, 420), n_target=12)
    img = torch.from_numpy(f["pixels"][0, 0][None]).cuda()
    assert model.fused is not None and model.fused.fused_pair
    model.fused.pair_head = True
    got = model.run_network(img)
    model.fused.pair_head = False
    want = model.run_network(img)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


@pytest.mark.gpu
def test_dynamics_do_not_depend_on_what_the_workspace_held(engine):
    """The dynamics initialise their per-label words and seed map where they use them instead of clearing 30 bytes per padded
    pixel (csrc/dynamics.hip); the workspace is reused from call to call with another batch size = another layout.  Whatever
    it held before — zeros, ones, a previous batch's words — the labels are the same."""
    import torch
    from aliby_amd.segment import dynamics

    tiles = [_flows((224, 256), 14, 40 + k, "nuclei") for k in range(3)]
    dP = torch.from_numpy(np.stack([t[1] for t in tiles])).cuda()
    prob = torch.from_numpy(np.stack([t[2] for t in tiles])).cuda()
    ref, n_ref = dynamics.masks_from_flows(engine, dP, prob)
    ref = ref.cpu().numpy()
    (ws,) = [w for w in dynamics._workspaces.values()]
    g = torch.Generator(device="cuda").manual_seed(1)
    for fill in (0x00, 0xFF, 0x7F, 0x01, "random"):
        if fill == "random":
            ws.copy_(torch.randint(0, 256, (ws.numel(),), dtype=torch.uint8, device="cuda", generator=g))
        else:
            ws.fill_(fill)
        for sl in (slice(0, 3), slice(2, 3), slice(0, 1)):  # other batch sizes lay the same buffer out differently
            got, n = dynamics.masks_from_flows(engine, dP[sl].contiguous(), prob[sl].contiguous())
            assert np.array_equal(n, n_ref[sl]), (fill, sl)
            assert np.array_equal(got.cpu().numpy(), ref[sl]), (fill, sl)


def test_dynamics_direct_gather_form_has_the_same_labels(engine, monkeypatch):
    """ALIBY_DYN_DIRECT=1: the flow following gathers from (dP, cellprob) themselves instead of from a normalised copy of the flow
    field (less HBM traffic, a little slower, not the default): end points and labels bit for bit those of the default form."""
    import torch
    from aliby_amd.segment.dynamics import masks_from_flows

    tiles = [_flows((256, 288), 26, fov, "cells") for fov in (0, 1, 2)]
    dP = torch.from_numpy(np.stack([t[1] for t in tiles])).cuda()
    prob = torch.from_numpy(np.stack([t[2] for t in tiles])).cuda()
    la, na, pa = masks_from_flows(engine, dP, prob, return_endpoints=True)
    la, pa = la.clone(), pa.clone()
    monkeypatch.setenv("ALIBY_DYN_DIRECT", "1")
    lb, nb, pb = masks_from_flows(engine, dP, prob, return_endpoints=True)
    assert torch.equal(la, lb) and np.array_equal(na, nb)
    fg = (prob > 0)[:, None].expand_as(pa)
    assert torch.equal(pa[fg], pb[fg])


@pytest.mark.gpu
def test_dynamics_do_not_depend_on_the_order_of_the_foreground_list(engine, monkeypatch):
    """The foreground pixels are compacted in 4096-pixel chunks whose order in the list is the order their workgroups reserved
    space — usually ascending, not always.  Round 3 shipped a first-position reduction that assumed ascending positions inside a
    run of one label: where a wave of the list straddled two chunks that had landed the other way round, inside a mask spanning
    both, the mask's first position came out too large and two masks swapped their numbers (about once in ten runs of a
    7-position job).  ALIBY_DEBUG_FG_REVERSE=1 makes the chunks land in DESCENDING order every time; the frame below has a mask
    across the first chunk boundary (rows 15 | 16 of a 256-pixel-wide frame) and a small one beside its upper part."""
    import torch
    from aliby_amd.segment.dynamics import masks_from_flows
    from oracle import cellpose_restated as cr

    yy, xx = np.mgrid[0:64, 0:256]
    frames = []
    for k in range(6):  # (the critical stretch must not contain a wave boundary of the list: a few sizes, one of them will do)
        gt = np.zeros((64, 256), np.int32)
        gt[(yy - 16) ** 2 + (xx - 40) ** 2 <= (8 + 0.5 * k) ** 2] = 1
        gt[(yy - 13) ** 2 + (xx - 150) ** 2 <= 3.2 ** 2] = 2
        frames.append((gt,) + synth.analytic_flows(gt))
    dP = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    prob = torch.from_numpy(np.stack([f[2] for f in frames])).cuda()
    monkeypatch.delenv("ALIBY_DEBUG_FG_REVERSE", raising=False)
    ref, n_ref = masks_from_flows(engine, dP, prob)
    monkeypatch.setenv("ALIBY_DEBUG_FG_REVERSE", "1")
    got, n = masks_from_flows(engine, dP, prob)
    assert list(n_ref) == [2] * 6 and np.array_equal(n, n_ref)
    for k, (gt, d, p) in enumerate(frames):
        want = cr.compute_masks(d, p)
        assert np.array_equal(ref[k].cpu().numpy(), want), k
        assert np.array_equal(got[k].cpu().numpy(), want), (k, "reversed list")


@pytest.mark.gpu
def test_segmenters_of_equal_parameters_share_one_model(engine):
    """init_step builds a segmenter per position, as the reference does; the model behind it (weights packed, workspaces) is
    built once per parameter set and process (segment/dispatch.py _model_for); other parameters get another model;
    release_pinned() drops them."""
    import warnings

    from aliby_amd import runner
    from aliby_amd.segment import dispatch

    runner.release_pinned()
    override = lambda x: None  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = dispatch.dispatch_segmenter("cellpose", 0, setup_params=dict(flows_override=override))
        b = dispatch.dispatch_segmenter("cellpose", 1, setup_params=dict(flows_override=override))
        c = dispatch.dispatch_segmenter("cellpose", 0, setup_params=dict(flows_override=override, seed=4))
        d = dispatch.dispatch_segmenter("cellpose", 0, setup_params=dict(flows_override=lambda x: None))
    assert a is not b and a.model is b.model and a.channel_to_segment == 0 and b.channel_to_segment == 1
    assert c.model is not a.model and d.model is not a.model
    assert len(dispatch._MODELS) == 3
    runner.release_pinned()
    assert dispatch._MODELS == []


@pytest.mark.gpu
def test_eval_keywords_of_cellpose_are_honoured_or_refused(engine):
    """`segment(pixels, **kw)` hands its keywords to `model.eval` (dispatch.py:208-215).  Tile geometry per call is honoured and
    changes the flows' blending only; keywords that would change the result in ways that are not built (diameter resizing,
    augment, invert, ...) raise instead of being dropped; neutral values and cellpose-3 habits (channels=[0, 0]) pass."""
    import torch
    from aliby_amd.segment.cellpose_hip import CellposeModel

    f = synth.make_fov(2, 3, shape=(300, 520), n_channels=1, n_target=20)
    x = torch.from_numpy(f["pixels"][0, 0][None].copy()).cuda()
    model = CellposeModel()
    base = model.eval(x)
    same = model.eval(x, channels=[0, 0], diameter=30.0, resample=True, augment=False, rescale=None, batch_size=3)
    assert torch.equal(base[0], same[0]) and torch.equal(base[1][1], same[1][1])  # tiles per forward do not change a bit
    wide = model.eval(x, bsize=256)
    assert wide[1][1].shape == base[1][1].shape and not torch.equal(wide[1][1], base[1][1])  # other tiles, other blending
    for kw in (dict(diameter=17), dict(augment=True), dict(invert=True), dict(rescale=0.5), dict(channels=[2, 1]), dict(resample=False)):
        with pytest.raises(NotImplementedError):
            model.eval(x, **kw)
    with pytest.raises(TypeError):
        model.eval(x, no_such_keyword=1)
    with pytest.raises(ValueError):
        CellposeModel(model_type="cyto3")  # (a NAME is something cellpose downloads: the checkpoint's path goes in pretrained_model)
    with pytest.raises(NotImplementedError):
        CellposeModel(nchan=3)
    with pytest.raises(TypeError):
        CellposeModel(no_such_option=1)
    # cellpose's normalize option dict: neutral entries pass, `normalize: False` switches the percentile step off, the rest raises
    opts = model.eval(x, normalize=dict(norm3D=False, percentile=[1, 99], lowhigh=None, sharpen_radius=0, invert=False))
    assert torch.equal(opts[1][1], base[1][1])
    raw = model.eval(x, normalize=dict(normalize=False))
    assert torch.equal(raw[1][1], model.eval(x, normalize=False)[1][1]) and not torch.equal(raw[1][1], base[1][1])
    for bad in (dict(percentile=[2, 98]), dict(tile_norm_blocksize=100), dict(lowhigh=[0, 1000]), dict(invert=True)):
        with pytest.raises(NotImplementedError):
            model.eval(x, normalize=bad)
