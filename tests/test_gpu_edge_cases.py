"""
GPU parity on the edge cases: objects too large for LDS (global-scratch kernel variants), float32 pixel
planes, several tiles with ragged label counts, Z > 1 reduced once on the device, maximum label values.
"""

import numpy as np
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu


def _close(a, b, key, rtol=1e-4):
    a, b = np.asarray(a, float), np.asarray(b, float)
    if key.endswith("Orientation"):
        # an axis has no sign: +90 and -90 degrees are one orientation, reached from either side of atan2's branch cut when the
        # mixed central moment is zero up to rounding (a vertical line: exactly zero in the kernel's integer moments, +-1e-17
        # in float moments); +-45 is the tie rule of an isotropic tensor
        a, b = np.atleast_1d(a), np.atleast_1d(b)
        flip = (np.isclose(np.abs(a), 45.0) & np.isclose(np.abs(b), 45.0)) | (np.isclose(np.abs(a), 90.0) & np.isclose(np.abs(b), 90.0))
        a, b = a[~flip], b[~flip]
    if "ZernikePhase" in key:  # an angle: -pi and +pi are one phase (a real negative moment, either side of atan2's branch cut)
        a, b = np.atleast_1d(a), np.atleast_1d(b)
        wrap = np.isclose(np.abs(a), np.pi) & np.isclose(np.abs(b), np.pi)
        a, b = a[~wrap], b[~wrap]
    # InfoMeas2 = sqrt(1 - exp(-2 (HXY2 - HXY))): where the two entropies are equal (independent grey levels) the argument is a
    # few ulps of 1 and the square root turns 4e-16 into 3e-8 — or into exactly 0 when the difference rounds to zero
    atol = 1e-6 if "InfoMeas2" in key else 1e-8
    assert np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True), (key, a.ravel()[:4], b.ravel()[:4])


def _run_both(tree, masks, pixels, multi=False, kw=None):
    from aliby_amd.extraction.extract import extract_tree, extract_tree_multi, process_tree_masks
    from oracle import aliby_extract as ox

    fn, fo = (extract_tree_multi, ox.extract_tree_multi) if multi else (extract_tree, ox.extract_tree)
    inst, res = process_tree_masks(tree, masks, pixels, fn, cp_measure_kwargs=kw)
    inst_o, res_o = ox.process_tree_masks(tree, masks, pixels, fo, cp_measure_kwargs=kw)
    assert inst == inst_o and len(res) == len(res_o)
    for i, (a, b) in enumerate(zip(res, res_o)):
        if isinstance(b, dict):
            assert set(a) == set(b)
            for k in b:
                if "ZernikePhase" in k:  # the phase of a vanishing moment (a one-pixel object, a symmetric one) is rounding noise
                    mag = k.replace("ZernikePhase", "ZernikeMagnitude")
                    if mag in b and np.all(np.abs(np.asarray(b[mag], float)) < 1e-9) and np.all(np.abs(np.asarray(a[mag], float)) < 1e-9):
                        continue
                _close(a[k], b[k], f"{inst[i][1]}/{k}")
        else:
            _close(a, float(b), str(inst[i][1]))
    return inst, res


def test_huge_objects_use_global_scratch_and_match(engine):
    """One object of ~46k pixels (does not fit any LDS budget) next to small ones, every family."""
    Y, X = 320, 352
    lab = np.zeros((Y, X), np.uint16)
    yy, xx = np.mgrid[0:Y, 0:X]
    lab[((yy - 150) / 120.0) ** 2 + ((xx - 170) / 130.0) ** 2 <= 1.0] = 1          # big ellipse
    lab[((yy - 150) / 20.0) ** 2 + ((xx - 170) / 25.0) ** 2 <= 1.0] = 0            # with a hole
    lab[290:300, 300:330] = 2
    lab[5:9, 5:9] = 3
    rng = np.random.default_rng(4)
    px = rng.integers(200, 40000, size=(1, 2, 1, Y, X)).astype(np.uint16)
    for c in range(2):
        px[0, c, 0][lab == 1] += (2000 * np.sin(xx[lab == 1] / (9.0 + c)) + 2000).astype(np.uint16)
    feats = ["intensity", "feret", "zernike", "radial_zernikes", "texture", "radial_distribution"]
    tree = {"None": {"None": ["sizeshape", "area", "volume"]}, 0: {"max": feats + ["median", "max2p5pc"]}}
    _run_both(tree, [lab], px)
    _run_both({(0, 1): {"None": {"max": ["pearson", "costes", "manders_fold", "rwc"]}}}, [lab], px, multi=True)


def test_float32_planes_every_family(engine):
    f = synth.make_fov(1, 8, shape=(192, 208), n_channels=2, n_target=14)
    pixels = (f["pixels"].astype(np.float32) / np.float32(30000.0)).clip(0, 1)[None]  # [1,C,1,Y,X] float32 in [0,1]
    tree = {0: {"max": ["intensity", "radial_zernikes", "texture", "radial_distribution", "mean", "std", "median"]}}
    _run_both(tree, [f["cells"]], pixels)
    _run_both({(0, 1): {"None": {"max": ["pearson", "costes", "manders_fold", "rwc"]}}}, [f["cells"]], pixels, multi=True)


def test_uint8_planes_every_family(engine):
    """8-bit sources: stored as uint16 on the device, but texture takes their grey level as img_as_ubyte does for uint8 (the
    value itself, not value >> 8 — which would be zero everywhere); Z = 2 so that the max-projection carries the 8-bit mark."""
    f = synth.make_fov(1, 9, shape=(176, 192), n_channels=2, n_z=2, n_target=12)
    pixels = (f["pixels"] >> 6).clip(0, 255).astype(np.uint8)[None]  # [1,C,2,Y,X] uint8
    assert pixels.max() > 100
    tree = {0: {"max": ["intensity", "radial_zernikes", "texture", "radial_distribution", "mean", "median"], "add": ["mean"]},
            1: {"max": ["texture"]}}
    inst, res = _run_both(tree, [f["cells"]], pixels)
    tex = [np.asarray(r["Entropy_3_00_256"], float) for r in res if isinstance(r, dict) and "Entropy_3_00_256" in r]
    assert tex and any((t[np.isfinite(t)] > 0).any() for t in tex)  # (all-zero grey levels would give no entropy at all)
    _run_both({(0, 1): {"None": {"max": ["pearson", "costes", "manders_fold", "rwc"]}}}, [f["cells"]], pixels, multi=True)


def test_thin_objects_where_every_pixel_is_the_cone_top(engine):
    """cell.py's min_maj_approximation (cell.py:207-229) on an object one or two pixels thick — every pixel at the same distance
    from the edge — ends in `distance_transform_edt` of a frame without background, for which scipy returns the distance to a
    virtual point above the frame's first column.  volume / eccentricity of such objects carry that number (it depends on where
    the object lies in the frame); the kernel returns the same.  Found by tests/fuzz/fuzz_features.py (the kernel returned 0)."""
    lab = np.zeros((96, 128), np.uint16)
    lab[5, 7] = 1                  # one pixel
    lab[20:22, 30:32] = 2          # 2 x 2
    lab[40, 10:19] = 3             # a 1 x 9 line
    lab[60:68, 100:102] = 4        # an 8 x 2 bar
    lab[70:90, 20:50] = 5          # an ordinary object beside them
    lab[94:96, 126:128] = 6        # 2 x 2 in the last corner of the frame
    px = np.random.default_rng(1).integers(100, 5000, size=(1, 1, 1, 96, 128)).astype(np.uint16)
    tree = {"None": {"None": ["volume", "eccentricity", "conical_volume", "spherical_volume", "area"]}, 0: {"max": ["mean", "max2p5pc"]}}
    inst, res = _run_both(tree, [lab], px)
    vols = [float(r) for i, r in zip(inst, res) if i[1][-1] == "volume"]
    assert len(vols) == 6 and all(v > 0 for v in vols) and vols[5] > vols[1]  # (the same 2 x 2 shape, further from the origin)


def test_ragged_tiles_and_z_reduction(engine):
    """Three tiles with 0 / few / many objects and Z = 3: one max-projection on the device."""
    tiles = [synth.make_fov(4, k, shape=(160, 176), n_channels=2, n_z=3, n_target=n) for k, n in ((0, 10), (1, 4))]
    masks = [tiles[0]["nuclei"], np.zeros((160, 176), np.uint16), tiles[1]["cells"]]
    pixels = np.stack([tiles[0]["pixels"], tiles[0]["pixels"][::-1].copy(), tiles[1]["pixels"]])  # [3,2,3,Y,X]
    tree = {"None": {"None": ["sizeshape"]}, 1: {"max": ["intensity", "texture"]}, 0: {"max": ["intensity"]}}
    inst, res = _run_both(tree, masks, pixels)
    tiles_seen = sorted({t[0][0] for t in inst})
    assert tiles_seen == [0, 2]
    _run_both({(0, 1): {"None": {"max": ["pearson", "rwc"]}}}, masks, pixels, multi=True)
    # per-feature kwargs reach the kernels
    _run_both({0: {"max": ["intensity", "texture", "radial_distribution"]}}, masks, pixels,
              kw={"intensity": {"edge_measurements": False}, "texture": {"scale": 2}, "radial_distribution": {"bin_count": 5}})


def test_label_values_up_to_uint16_limit(engine):
    """Sparse label ids near 65534: rows exist for 1..max (absent ones NaN), as process_tree_masks enumerates."""
    from aliby_amd.extraction.extract import extract_tree, process_tree_masks

    lab = np.zeros((64, 64), np.uint16)
    lab[4:10, 4:10] = 1
    lab[20:30, 20:40] = 300
    px = np.random.default_rng(0).integers(0, 9000, size=(1, 1, 1, 64, 64), dtype=np.uint16)
    inst, res = process_tree_masks({0: {"max": ["intensity"]}}, [lab], px, extract_tree)
    assert len(res) == 300
    assert res[0]["Intensity_MaxIntensity"][0] == px[0, 0, 0][lab == 1].max()
    assert np.isnan(res[150]["Intensity_MeanIntensity"][0])
    assert np.isclose(res[299]["Intensity_MeanIntensity"][0], px[0, 0, 0][lab == 300].mean())
    from aliby_amd.segment.dispatch import _to_uint16_labels

    with pytest.raises(OverflowError):
        _to_uint16_labels(np.array([[65535]], dtype=np.int64))
