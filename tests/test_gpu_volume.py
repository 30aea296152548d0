"""
GPU parity of the round-3 "Z-stack as a volume" extension (BASELINE config 5, "true 3-D"; beyond what the reference wires —
parity unpinned against cellpose, checked here against the CPU restatement in oracle/volume_restated.py and against the
synthetic ground truth).
"""
import numpy as np
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu


def _stack(seed=3, shape=(128, 160), n_z=8, n_target=10):
    f = synth.make_fov(5, seed, shape=shape, n_channels=2, n_z=n_z, n_target=n_target)
    gt = synth.ellipsoid_planes(f["nuclei"], n_z, seed=seed)
    return f, gt


def _per_plane_flows(gt):
    from oracle import tiler_ref

    dP, prob = [], []
    for z in range(gt.shape[0]):
        d, p = synth.analytic_flows(tiler_ref.relabel_sequential(gt[z]))
        dP.append(d)
        prob.append(p)
    return np.stack(dP), np.stack(prob)


def test_planes_are_stitched_into_the_volume_like_the_restatement(engine):
    import torch

    from aliby_amd.segment.dispatch import dispatch_segmenter
    from oracle import cellpose_restated as cr
    from oracle import volume_restated as vr

    f, gt = _stack()
    dP, prob = _per_plane_flows(gt)

    def override(x):
        assert x.shape[0] == gt.shape[0]
        return torch.from_numpy(dP).cuda(), torch.from_numpy(prob).cuda()

    segment = dispatch_segmenter(kind="cellpose", channel_to_segment=0, setup_params=dict(flows_override=override))
    labels2d = segment(f["pixels"][None], do_3D=True)
    volume, counts = segment.last_volume
    got = volume.cpu().numpy()[0]
    planes = np.stack([cr.finish_labels(cr.compute_masks(dP[z], prob[z])) for z in range(gt.shape[0])])
    want, n = vr.stitch3d(planes, 0.01)
    assert int(counts[0]) == n and np.array_equal(got, want)
    # against the ground truth: the same partition of the voxels (labels differ by a renaming), one object per ellipsoid
    present = np.unique(gt[gt > 0])
    assert n == len(present)
    pairs = np.unique(np.stack([gt[gt > 0], got[gt > 0]]), axis=1)
    assert pairs.shape[1] == n and np.array_equal((got > 0), (gt > 0))
    # the step result is the reference's collapse of the 3-D labels: max over Z, then relabel_sequential (dispatch.py:216-223)
    from oracle import tiler_ref

    assert labels2d.dtype == np.uint16 and np.array_equal(labels2d, tiler_ref.relabel_sequential(want.max(axis=0)))


def test_intensity3d_matches_numpy_on_the_volume(engine):
    import torch

    from aliby_amd.extraction.features import intensity3d_names
    from oracle import volume_restated as vr

    f, gt = _stack(seed=5, n_z=6)
    from oracle import tiler_ref

    vol = tiler_ref.relabel_sequential(gt).astype(np.uint16)  # labels 1..n over the volume
    n = int(vol.max())
    stack2 = np.stack([vol, np.roll(vol, 7, axis=2)])                       # two stacks in one call
    px = np.stack([f["pixels"], f["pixels"][::-1].copy()])                 # [F=2, C, Z, Y, X]
    counts = [int(stack2[0].max()), int(stack2[1].max())]
    assert len(intensity3d_names()) == 12
    for c in (0, 1):
        got = engine.intensity3d(torch.from_numpy(stack2).cuda(), torch.from_numpy(px).cuda(), c, counts).cpu().numpy()
        want = np.concatenate([vr.intensity3d(stack2[k], px[k, c]) for k in range(2)])
        assert got.shape == want.shape == (2 * n, 12)
        assert np.allclose(got, want, rtol=1e-10, atol=1e-9, equal_nan=True)
        assert np.array_equal(got[:, [0, 1, 4, 5]], want[:, [0, 1, 4, 5]])  # counts, sums, min, max: integers, exact
    again = engine.intensity3d(torch.from_numpy(stack2).cuda(), torch.from_numpy(px).cuda(), 1, counts).cpu().numpy()
    assert np.array_equal(again, got, equal_nan=True)  # integer sums: run-to-run identical


def test_stitch_threshold_range_and_lut_overflow(engine):
    import torch

    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    lab = torch.zeros((1, 8, 8), dtype=torch.uint16, device="cuda")
    lab[0, 2:5, 2:5] = 1
    off = np.asarray([0, 1], np.int32)
    lut = torch.tensor([70000], dtype=torch.int32, device="cuda")
    out = torch.empty_like(lab)
    with pytest.raises(OverflowError):
        _lib.check(engine.lib.aliby_labels_apply_lut(engine.ctx.handle, _ptr(lab), 1, 8, 8, _ptr(off), _ptr(lut), _ptr(out), _stream_ptr()))
    lut[0] = 9
    _lib.check(engine.lib.aliby_labels_apply_lut(engine.ctx.handle, _ptr(lab), 1, 8, 8, _ptr(off), _ptr(lut), _ptr(out), _stream_ptr()))
    host = out.cpu().numpy()
    assert int(host.max()) == 9 and int((host > 0).sum()) == 9
