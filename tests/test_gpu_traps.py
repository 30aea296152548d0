"""
Trap detection on the GPU: every kernel of csrc/traps.hip against the CPU restatement
(oracle/traps_restated.py, itself pinned to the reference's functions by tests/golden/reference_traps.json), then the two
reference functions end to end against those fixtures, then the tiler using them.
Tolerances: float64 images agree to 1e-9 relative (sums are taken in another order); binary images, labels, region
areas and the returned coordinates are exact.
"""

import json
from pathlib import Path

import numpy as np
import pytest
import torch

from aliby_amd import synth

pytestmark = pytest.mark.gpu

CASES = json.loads((Path(__file__).parent / "golden" / "reference_traps.json").read_text())["cases"]


def _close(got, want, rtol=1e-9, atol=1e-9):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=rtol, atol=atol), float(np.abs(got - want).max())


def test_resampling_kernels_match_the_restatement(engine):
    from aliby_amd.tile import traps as gt
    from oracle import traps_restated as tr

    image, _ = synth.trap_image(seed=5, shape=(190, 230))
    f = image.astype(np.float64)
    dev = gt._f64(image)
    _close(gt.rescale(dev, 0.4, integer_input=True), tr.rescale(image, 0.4, integer_input=True))
    _close(gt.rescale(dev, 0.35), tr.rescale(f, 0.35))
    small = tr.rescale(f, 0.35)
    _close(gt.rescale(gt._f64(small), 1 / 0.35), tr.rescale(small, 1 / 0.35))
    _close(gt.resize(dev, (190, 230)), f)  # same shape: identity
    templ = f[20:61, 30:71]
    for angle in (0, 90, 180, 270, 33.0):
        _close(gt.rotate(gt._f64(templ), angle, 2999.5), tr.rotate(templ, angle, 2999.5))
    for scale in (0.5, 1.1666666666666665, 2.0):
        _close(gt.rescale(gt._f64(templ), scale), tr.rescale(templ, scale))


@pytest.mark.parametrize("radius", [0, 1, 2, 5])
def test_rank_entropy(engine, radius):
    from aliby_amd.tile import traps as gt
    from oracle import traps_restated as tr

    rng = np.random.default_rng(radius)
    u8 = rng.integers(0, 7, (61, 83)).astype(np.uint8)
    u8[20:40, 30:60] = rng.integers(0, 256, (20, 30))
    dev = torch.from_numpy(u8).cuda()
    out = torch.empty(u8.shape, dtype=torch.float64, device="cuda")
    gt._call("aliby_trap_entropy", dev.data_ptr(), 61, 83, radius, out.data_ptr())
    _close(out, tr.rank_entropy(u8, radius), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_closing_label_and_regions(engine, k):
    from aliby_amd.tile import traps as gt
    from oracle import traps_restated as tr

    rng = np.random.default_rng(40 + k)
    bw = rng.random((97, 120)) < 0.32
    bw[30:50, 40:80] |= rng.random((20, 40)) < 0.8
    bw[0:6, 10:30] = True  # touches the border: clear_border removes it
    closed = gt._closing(torch.from_numpy(bw.astype(np.uint8)).cuda(), k)
    want = tr.closing_square(bw, k)
    assert np.array_equal(closed.cpu().numpy().astype(bool), want)
    got = gt._regions(closed)
    ref = tr.regions(tr.clear_border(want))
    assert [r["area"] for r in got] == [r["area"] for r in ref]
    for a, b in zip(got, ref):
        assert np.allclose(a["centroid"], b["centroid"], rtol=1e-12)
        assert np.isclose(a["major_axis_length"], b["major_axis_length"], rtol=1e-9, atol=1e-9)


def test_label_handles_spirals_and_empty_images(engine):
    from aliby_amd.tile import traps as gt
    from scipy import ndimage as ndi

    bw = np.zeros((64, 64), bool)
    for r in range(2, 30, 4):  # nested open rings joined into one long snake
        bw[r, r:64 - r] = bw[63 - r, r:64 - r] = True
        bw[r:64 - r, r] = bw[r + 4:64 - r, 63 - r] = True
    for img in (bw, np.zeros((5, 9), bool), np.ones((7, 3), bool)):
        H, W = img.shape
        lab = torch.empty((H, W), dtype=torch.int32, device="cuda")
        gt._call("aliby_trap_label", torch.from_numpy(img.astype(np.uint8)).cuda().data_ptr(), H, W, lab.data_ptr())
        want, n = ndi.label(img, structure=np.ones((3, 3), bool))
        got = lab.cpu().numpy()
        assert len(np.unique(got[got > 0])) == n
        # same partition, and labels ordered by first raster pixel like scipy / skimage
        firsts = sorted(np.unique(got[got > 0]))
        for rank, l in enumerate(firsts, 1):
            assert np.array_equal(got == l, want == rank)


def test_match_template_and_peaks(engine):
    from aliby_amd.tile import traps as gt
    from oracle import traps_restated as tr

    image, _ = synth.trap_image(seed=8, shape=(300, 340))
    small = tr.rescale(image.astype(np.float64), 0.35)
    for th, tw in ((41, 41), (20, 20), (33, 48)):
        templ = small[10:10 + th, 12:12 + tw].copy()
        got = gt.match_template(gt._f64(small), gt._f64(templ))
        want = tr.match_template(small, templ)
        _close(got, want, rtol=1e-7, atol=1e-9)
        assert np.isclose(gt._percentile(got**2, 99.9), np.percentile(want**2, 99.9), rtol=1e-7)
    matched = tr.match_template(small, small[10:51, 12:53].copy()) ** 2
    big = tr.rescale(matched, 1 / 0.35)
    for dist, border in ((81, 39), (20, 0), (5, 7)):
        assert np.array_equal(gt.peak_local_max(gt._f64(big), dist, border), tr.peak_local_max(big, dist, border))


@pytest.mark.parametrize("index", [0, 1, 2])
def test_segment_traps_matches_the_reference_fixtures(engine, index):
    from aliby_amd.tile import traps as gt
    from oracle import traps_restated as tr

    case = CASES[index]
    image, _ = synth.trap_image(seed=case["seed"])
    tile = case["tile_size"]
    for tag, downscale in (("first", 0.4), ("retry", 1)):
        want = case[tag]
        found = gt.trap_regions(image, tile, downscale=downscale)
        ref = tr.trap_regions(image, tile, downscale=downscale)
        assert found["disk_radius"] == want["disk_radius"]
        _close(found["entropy"], ref["entropy"], rtol=1e-10, atol=1e-10)
        assert np.isclose(found["otsu"], want["otsu"], rtol=1e-10)
        assert np.array_equal(found["bw"].cpu().numpy().astype(bool), ref["bw"])
        assert int(found["bw"].sum()) == want["foreground_pixels"]
        assert [r["area"] for r in found["regions"]] == [r["area"] for r in ref["regions"]]
        cents = np.array([r["centroid"] for r in found["valid"]]).round().astype(int)
        assert cents.tolist() == want["valid_centroids"]
    traps = gt.segment_traps(image, tile)
    assert traps.dtype.kind == "i" and traps.tolist() == case["segment_traps"]
    # identify_trap_locations alone, with the reference's own intermediate template size
    y, x = case["first"]["valid_centroids"][0]
    lo, hi = tile // 2, -(tile // -2)
    one = gt.identify_trap_locations(image, image[y - lo: y + hi, x - lo: x + hi].astype(float))
    assert sorted(map(tuple, one.tolist())) == sorted(map(tuple, tr.identify_trap_locations(image, image[y - lo: y + hi, x - lo: x + hi].astype(float)).tolist()))


def test_segment_traps_errors(engine):
    from aliby_amd.tile import traps as gt

    flat = np.full((256, 256), 3000, np.uint16)
    flat[::2] += 1
    with pytest.raises(Exception, match="No valid tiles found"):
        gt.segment_traps(flat, 117)


def test_tiler_detects_traps_on_the_first_frame(engine):
    """Tiler with tile_size set (config 4): centres from segment_traps, near-edge ones dropped (tiler.py:672-696),
    tiles cropped around them with the drift of later frames; a frame the detector cannot handle falls back to the
    centre tile with the reference's warning."""
    from aliby_amd.tile.tiler import ImageArray, Tiler, TilerParameters

    case = CASES[0]
    frame, _ = synth.trap_image(seed=case["seed"])
    tczyx = np.stack([frame, np.roll(frame, (2, -1), (0, 1))])[:, None, None]
    tiler = Tiler.from_image(ImageArray(tczyx), TilerParameters(tile_size=117, ref_channel=0))
    out = tiler.run_tp(0)
    half = 117 // 2
    want = [c for c in case["segment_traps"] if half < c[0] < 512 - half and half < c[1] < 512 - half]
    assert len(want) >= 4
    assert out["pixels"].shape == (len(want), 1, 1, 117, 117)
    for k, (y, x) in enumerate(want):
        assert np.array_equal(out["pixels"][k, 0, 0], frame[y - half: y + 59, x - half: x + 59]), k
    flat = np.full((1, 1, 1, 300, 300), 3000, np.uint16)
    with pytest.warns(UserWarning, match="Trap detection failed"):
        centre = Tiler.from_image(ImageArray(flat), TilerParameters(tile_size=117, ref_channel=0)).run_tp(0)
    assert centre["pixels"].shape[0] == 1
