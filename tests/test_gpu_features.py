"""
GPU parity: HIP feature kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.
Tolerance: 1e-4 relative for floats (BASELINE.json north_star); integer-valued features
(areas, bboxes, counts, positions) must match exactly.
"""

import numpy as np
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _compare(names, got, ref_dict, exact=(), skip=()):
    bad = []
    for j, name in enumerate(names):
        if name in skip:
            continue
        r = np.asarray(ref_dict[name], dtype=float)
        g = got[:, j]
        if name in exact:
            ok = np.array_equal(np.nan_to_num(g, nan=-1), np.nan_to_num(r, nan=-1))
        else:
            ok = np.allclose(g, r, rtol=RTOL, atol=1e-9, equal_nan=True)
        if not ok:
            err = np.nan_to_num(np.abs(g - r) / (np.abs(r) + 1e-12), nan=np.inf)
            k = int(np.argmax(err))
            bad.append((name, k, g[k], r[k]))
    assert not bad, f"mismatch (name, object, hip, oracle): {bad[:8]} ... {len(bad)} columns"


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
@pytest.mark.parametrize("edge", [True, False])
def test_intensity_matches_oracle(engine, objset, edge):
    import torch
    from oracle import cp_measure_restated as cpm
    from aliby_amd.extraction import features as feat
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    f = synth.make_fov(1, 0)
    labels = f[objset]
    planes = f["pixels"][:, 0]  # [C,Y,X]
    dl = to_device_u16(labels[None])
    dp, dt = to_device_planes(planes[None])
    tab = engine.object_table(dl)
    assert tab.n_obj == labels.max()
    names = feat.intensity_names(edge)
    for ch in range(planes.shape[0]):
        out = engine.new_output(tab.n_obj, len(names))
        engine.intensity(dl, dp, dt, ch, tab, out, 0, edge_measurements=edge)
        torch.cuda.synchronize()
        ref = cpm.get_intensity(labels, planes[ch], edge_measurements=edge)
        _compare(names, out.cpu().numpy(), ref,
                 exact=("Intensity_MinIntensity", "Intensity_MaxIntensity", "Intensity_IntegratedIntensity",
                        "Location_MaxIntensity_X", "Location_MaxIntensity_Y"))


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
def test_sizeshape_matches_oracle(engine, objset):
    import torch
    from oracle import cp_measure_restated as cpm
    from aliby_amd.extraction import features as feat
    from aliby_amd.extraction.engine import to_device_u16

    f = synth.make_fov(1, 0)
    labels = f[objset]
    dl = to_device_u16(labels[None])
    tab = engine.object_table(dl)
    # object table is bit-exact vs scipy
    from scipy import ndimage as ndi

    sl = ndi.find_objects(labels.astype(np.int32))
    for i, s in enumerate(sl):
        row = tab.host[i]
        assert (row["y0"], row["y1"], row["x0"], row["x1"]) == (s[0].start, s[0].stop, s[1].start, s[1].stop)
    assert np.array_equal(tab.host["area"], np.bincount(labels.ravel())[1:])
    names = feat.sizeshape_names()
    out = engine.new_output(tab.n_obj, len(names))
    engine.sizeshape(dl, tab, out, 0)
    torch.cuda.synchronize()
    ref = cpm.get_sizeshape(labels)
    _compare(names, out.cpu().numpy(), ref,
             exact=("Area", "BoundingBoxArea", "BoundingBoxMaximum_X", "BoundingBoxMaximum_Y", "BoundingBoxMinimum_X",
                    "BoundingBoxMinimum_Y", "EulerNumber", "ConvexArea"))


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
@pytest.mark.parametrize("mode", ["u16", "f32_unit"])
def test_coloc_matches_oracle(engine, objset, mode):
    """pearson / manders_fold / rwc / costes for every channel pair of a 3-channel FOV."""
    import torch
    from itertools import combinations
    from oracle import cp_measure_restated as cpm
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    f = synth.make_fov(1, 1, shape=(320, 320), n_channels=3, n_target=30)
    labels = f[objset]
    planes = f["pixels"][:, 0]
    if mode == "f32_unit":  # CellProfiler-style [0,1] floats: exercises the Costes bisection properly
        planes = (planes.astype(np.float32) / np.float32(65535.0)).astype(np.float32)
    dl = to_device_u16(labels[None])
    dp, dt = to_device_planes(planes[None])
    tab = engine.object_table(dl)
    # the same metrics through the all-pairs launch (one workgroup per object, a wave per pair)
    pair_list = list(combinations(range(3), 2))
    allout = engine.new_output(tab.n_obj, 8 * len(pair_list))
    assert engine.coloc_pairs(dl, dp, dt, [(pr, dict(pearson=8 * i, manders_fold=8 * i + 2, rwc=8 * i + 4, costes=8 * i + 6))
                                           for i, pr in enumerate(pair_list)], tab, allout)
    torch.cuda.synchronize()
    allgot = allout.cpu().numpy()
    for pi, (c0, c1) in enumerate(pair_list):
        out = engine.new_output(tab.n_obj, 8)
        engine.coloc(dl, dp, dt, c0, c1, tab, out, dict(pearson=0, manders_fold=2, rwc=4, costes=6))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.allclose(allgot[:, 8 * pi: 8 * pi + 8], got, rtol=1e-9, atol=1e-12, equal_nan=True)
        # the reference evaluates one full-frame binary mask per object (extract.py:222-226); RWC's
        # ranks and Costes' threshold are per call, so the oracle is driven the same way
        ref = {}
        for lab in range(1, int(labels.max()) + 1):
            one = (labels == lab).astype(np.uint16)
            for fn in cpm.get_correlation_measurements().values():
                for k, v in fn(planes[c0], planes[c1], one).items():
                    ref.setdefault(k, []).append(v[0])
        names = ["Correlation_Pearson", "Correlation_Slope", "Correlation_Manders_1", "Correlation_Manders_2",
                 "Correlation_RWC_1", "Correlation_RWC_2", "Correlation_Costes_1", "Correlation_Costes_2"]
        _compare(names, got, ref)
        _compare(names, allgot[:, 8 * pi: 8 * pi + 8], ref)
    # a subset of metrics, and the refusal when the pixel lists would not fit the kernel's LDS budget
    part = engine.new_output(tab.n_obj, 4)
    assert engine.coloc_pairs(dl, dp, dt, [((0, 2), dict(pearson=0)), ((1, 2), dict(costes=2))], tab, part)
    torch.cuda.synchronize()
    assert np.allclose(part.cpu().numpy()[:, :2], allgot[:, 8:10], rtol=1e-9, equal_nan=True)
    assert np.allclose(part.cpu().numpy()[:, 2:], allgot[:, 22:24], rtol=1e-9, equal_nan=True)
    big = type(tab).__new__(type(tab))
    big.__dict__.update(tab.__dict__)
    big.max_area = 1 << 16
    assert engine.coloc_pairs(dl, dp, dt, [((0, 1), dict(pearson=0)), ((0, 2), dict(pearson=2))], big, part) is False


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
def test_zernike_and_mec_match_oracle(engine, objset):
    import torch
    from oracle import zernike_restated as zr
    from aliby_amd.extraction import features as feat
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    f = synth.make_fov(1, 2, shape=(300, 340), n_target=30)
    labels = f[objset]
    planes = f["pixels"][:, 0]
    dl = to_device_u16(labels[None])
    dp, dt = to_device_planes(planes[None])
    tab = engine.object_table(dl)
    mec = engine.mec(dl, tab).cpu().numpy()
    centres, radii = zr.minimum_enclosing_circle(labels)
    assert np.allclose(mec[:, 2], radii, rtol=1e-9), (mec[:3], radii[:3])
    assert np.allclose(mec[:, :2], centres, rtol=1e-9, atol=1e-9)
    # shape Zernikes
    names = feat.zernike_names()
    out = engine.new_output(tab.n_obj, 30)
    engine.zernike(dl, None, 0, 0, tab, out, 0, weighted=False)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = zr.get_zernike(labels)
    for j, n in enumerate(names):
        assert np.allclose(got[:, j], ref[n], rtol=RTOL, atol=1e-7), (n, got[:3, j], ref[n][:3])
    # intensity-weighted Zernikes: magnitudes then phases
    names = feat.radial_zernike_names()
    for ch in range(planes.shape[0]):
        out = engine.new_output(tab.n_obj, 60)
        engine.zernike(dl, dp, dt, ch, tab, out, 0, weighted=True)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        ref = zr.get_radial_zernikes(labels, planes[ch])
        scale = np.abs(ref[names[0]])  # Z_00 magnitude = mean intensity inside the disc
        for j in range(30):
            assert np.allclose(got[:, j], ref[names[j]], rtol=RTOL, atol=1e-7 * scale.max()), names[j]
            mag = ref[names[j]]
            dphi = np.angle(np.exp(1j * (got[:, 30 + j] - ref[names[30 + j]])))
            ok = (np.abs(dphi) < 1e-4) | (mag < 1e-6 * scale)
            assert ok.all(), (names[30 + j], got[~ok, 30 + j], ref[names[30 + j]][~ok])


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
@pytest.mark.parametrize("mode", ["u16", "f32_unit", "u8"])
def test_texture_matches_oracle(engine, objset, mode):
    import torch
    from oracle import texture_restated as tx
    from aliby_amd.extraction import features as feat
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    f = synth.make_fov(1, 4, shape=(288, 320), n_target=28)
    labels = f[objset]
    planes = f["pixels"][:, 0]
    if mode == "f32_unit":
        planes = (planes.astype(np.float32) / np.float32(20000.0)).clip(0, 1).astype(np.float32)
    if mode == "u8":  # 8-bit sources: img_as_ubyte leaves them alone (no >> 8), in storage they are widened to uint16
        planes = (planes >> 6).clip(0, 255).astype(np.uint8)
        assert len(np.unique(planes)) > 100
    dl = to_device_u16(labels[None])
    dp, dt = to_device_planes(planes[None])
    assert dt == {"u16": 0, "f32_unit": 1, "u8": 4}[mode]
    tab = engine.object_table(dl)
    names = feat.texture_names(3, 256)
    for ch in range(planes.shape[0]):
        out = engine.new_output(tab.n_obj, 52)
        engine.texture(dl, dp, dt, ch, tab, out, 0)
        torch.cuda.synchronize()
        ref = tx.get_texture(labels, planes[ch])
        _compare(names, out.cpu().numpy(), ref)
    # a different scale and grey-level count, and an object too small for any pair (NaN row)
    tiny = np.zeros((64, 64), np.uint16)
    tiny[10:12, 10:12] = 1
    tiny[30:50, 20:45] = 2
    px = np.random.default_rng(2).integers(0, 65535, size=(1, 1, 64, 64), dtype=np.uint16)
    dl, (dp, dt) = to_device_u16(tiny[None]), to_device_planes(px)
    tab = engine.object_table(dl)
    out = engine.new_output(tab.n_obj, 52)
    engine.texture(dl, dp, dt, 0, tab, out, 0, scale=5, gray_levels=64)
    torch.cuda.synchronize()
    ref = tx.get_texture(tiny, px[0, 0], scale=5, gray_levels=64)
    got = out.cpu().numpy()
    assert np.isnan(got[0]).all() and np.isnan(ref["Contrast_5_00_64"][0])
    _compare(feat.texture_names(5, 64), got, ref)


@pytest.mark.parametrize("objset", ["nuclei", "cells"])
def test_radial_distribution_matches_oracle(engine, objset):
    import torch
    from oracle import radial_restated as rr
    from aliby_amd.extraction import features as feat
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    f = synth.make_fov(1, 5, shape=(280, 300), n_target=26)
    labels = f[objset].copy()
    # one object touching the image border and one concave object (geodesic != straight line)
    labels[0:14, 40:70] = labels.max() + 1
    lab_c = labels.max() + 1
    labels[200:240, 200:206] = lab_c
    labels[200:206, 200:240] = lab_c
    labels[234:240, 200:240] = lab_c
    planes = f["pixels"][:, 0]
    dl = to_device_u16(labels[None])
    dp, dt = to_device_planes(planes[None])
    tab = engine.object_table(dl)
    for bin_count in (4, 6):
        names = feat.radial_distribution_names(bin_count)
        assert names == rr.names(bin_count)
        for ch in range(planes.shape[0]):
            out = engine.new_output(tab.n_obj, 3 * bin_count)
            engine.radial_distribution(dl, dp, dt, ch, tab, out, 0, bin_count=bin_count)
            torch.cuda.synchronize()
            ref = rr.get_radial_distribution(labels, planes[ch], bin_count=bin_count)
            _compare(names, out.cpu().numpy(), ref)
    # scaled=False (CellProfiler's unscaled bins): rings of maximum_radius / bin_count pixels + the overflow ring
    for bin_count, maximum_radius in ((4, 10), (5, 40), (4, 3)):
        names = feat.radial_distribution_names(bin_count, scaled=False)
        assert names == rr.names(bin_count, scaled=False) and len(names) == 3 * (bin_count + 1) and names[bin_count].endswith("_Overflow")
        out = engine.new_output(tab.n_obj, len(names))
        assert engine.radial_distribution(dl, dp, dt, 0, tab, out, 0, bin_count=bin_count, scaled=False, maximum_radius=maximum_radius) == len(names)
        torch.cuda.synchronize()
        ref = rr.get_radial_distribution(labels, planes[0], bin_count=bin_count, scaled=False, maximum_radius=maximum_radius)
        _compare(names, out.cpu().numpy(), ref)
        got = out.cpu().numpy()
        present = np.bincount(labels.ravel())[1:] > 0
        assert np.allclose(got[present][:, : bin_count + 1].sum(1), 1.0)  # the rings + the overflow partition the intensity
    # through the registry, with kwargs (cp_measure_kwargs -> loaders.py:71-73)
    from aliby_amd.extraction.extract import extract_tree, process_tree_masks
    from oracle import aliby_extract as ox

    kw = {"radial_distribution": {"scaled": False, "maximum_radius": 8, "bin_count": 3}}
    tree = {1: {"max": ["radial_distribution"]}}
    inst, res = process_tree_masks(tree, labels, f["pixels"][None], extract_tree, cp_measure_kwargs=kw)
    inst_o, res_o = ox.process_tree_masks(tree, labels, f["pixels"][None], ox.extract_tree, cp_measure_kwargs=kw, max_objects=6)
    assert list(res[0]) == list(res_o[0]) and "RadialDistribution_MeanFrac_Overflow" in res[0]
    for a, b in zip(res, res_o):
        for k in b:
            assert np.allclose(a[k], b[k], rtol=1e-4, atol=1e-9, equal_nan=True), k


def test_stream_fanout_is_bit_identical_to_single_stream(engine, monkeypatch):
    """The feature families run on four side HIP streams (families._FanOut); a race between them would show up as a
    difference against the single-stream evaluation of the same batch.  Also checks the asynchronous rows download."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.batch import extract_batch

    fovs = [synth.make_fov(2, i, shape=(256, 320), n_channels=3, n_target=24) for i in range(3)]
    px = torch.from_numpy(np.stack([f["pixels"] for f in fovs])).cuda()  # [F,C,1,Y,X]
    labels = torch.from_numpy(np.stack([f["nuclei"] for f in fovs])).cuda()
    mono = {"None": {"None": ["sizeshape"]}}
    for c in range(3):
        mono[c] = {"max": ["radial_zernikes", "intensity", "feret", "texture", "radial_distribution", "zernike"]}
    multi = {(a, b): {"None": {"max": ["pearson", "costes", "manders_fold", "rwc"]}} for a in range(3) for b in range(a + 1, 3)}
    results = {}
    for n_streams in ("1", "4"):
        monkeypatch.setenv("ALIBY_FEATURE_STREAMS", n_streams)
        m1, names1, table = extract_batch(engine, labels, (px, _lib.U16), mono)
        m2, names2, _ = extract_batch(engine, labels, (px, _lib.U16), multi, multi=True, table=table)
        handle = engine.to_host_async((m1, m2), slot=int(n_streams) % 2)
        a1, a2 = handle.wait()
        results[n_streams] = (a1.copy(), a2.copy())
        assert np.array_equal(a1, engine.to_host(m1), equal_nan=True)
    for x, y in zip(results["1"], results["4"]):
        assert x.shape == y.shape and x.shape[0] == table.n_obj
        assert np.array_equal(x, y, equal_nan=True)


def test_granularity_matches_oracle(engine):
    """cp_measure "granularity" (CellProfiler MeasureGranularity).  PARITY UNPINNED as a whole: cp_measure is not available
    offline, the oracle (oracle/granularity_restated.py) restates CellProfiler's published algorithm and only its primitives
    are pinned (scikit-image 0.18.3 fixture, tests/test_oracle_golden.py).  float64 on both sides, rtol 1e-4 as north_star."""
    import torch
    from oracle import granularity_restated as gr
    from aliby_amd.extraction import features as feat
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    cases = [
        ((256, 256), dict()),                                             # CellProfiler's defaults
        ((250, 301), dict()),                                             # shapes that are not multiples of the subsampling
        ((200, 232), dict(subsample_size=0.5, image_sample_size=0.5, element_size=4, granular_spectrum_length=6)),
        ((96, 120), dict(subsample_size=1.0, image_sample_size=1.0, element_size=3, granular_spectrum_length=5)),
        ((180, 180), dict(subsample_size=0.5, image_sample_size=1.0, element_size=5, granular_spectrum_length=4)),
    ]
    for k, (shape, kw) in enumerate(cases):
        fovs = [synth.make_fov(1, 40 + 2 * k + i, shape=shape, n_target=14) for i in range(2)]
        labels = np.stack([f["cells"] for f in fovs])
        labels[1, 0:9, 10:30] = labels[1].max() + 1  # an object on the border
        planes = np.stack([f["pixels"][:, 0] for f in fovs])  # [F, C, Y, X] uint16
        dl = to_device_u16(labels)
        dp, dt = to_device_planes(planes)
        tab = engine.object_table(dl)
        L = kw.get("granular_spectrum_length", 16)
        names = feat.granularity_names(L)
        assert names == gr.names(L)
        for ch in range(planes.shape[1]):
            out = engine.new_output(tab.n_obj, L + 2)
            out.fill_(-7.0)
            assert engine.granularity(dl, dp, dt, ch, tab, out, 1, **kw) == L
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert (got[:, 0] == -7.0).all() and (got[:, -1] == -7.0).all()  # only its own columns are written
            row = 0
            for f in range(len(fovs)):
                ref = gr.get_granularity(labels[f], planes[f, ch], **kw)
                n = int(labels[f].max())
                _compare(names, got[row : row + n, 1 : 1 + L], ref)
                row += n
            assert row == tab.n_obj
    # float32 pixels, and the objects-as-image-mask variant of the kernel (mask = every object of the tile, bilinear sampling)
    f = synth.make_fov(1, 77, shape=(240, 240), n_target=16)
    labels = f["cells"][None]
    planes = (f["pixels"][:, 0].astype(np.float32) / 65535.0)[None]
    dl = to_device_u16(labels)
    dp, dt = to_device_planes(planes)
    tab = engine.object_table(dl)
    for kw in (dict(), dict(image_mask="objects", mask_order=1), dict(image_mask="objects", mask_order=1, subsample_size=0.5, image_sample_size=0.5, element_size=3, granular_spectrum_length=5)):
        L = kw.get("granular_spectrum_length", 16)
        out = engine.new_output(tab.n_obj, L)
        engine.granularity(dl, dp, dt, 0, tab, out, 0, **kw)
        torch.cuda.synchronize()
        _compare(feat.granularity_names(L), out.cpu().numpy(), gr.get_granularity(labels[0], planes[0, 0], **kw))
    with pytest.raises(NotImplementedError):
        engine.granularity(dl, dp, dt, 0, tab, out, 0, image_mask="objects", mask_order=3)
    # through the registry with kwargs, as the reference binds them (loaders.py:71-73), against the oracle's per-object loop
    from aliby_amd.extraction.extract import extract_tree, process_tree_masks
    from oracle import aliby_extract as ox

    f = synth.make_fov(2, 78, shape=(128, 128), n_target=6)
    kw = {"granularity": {"granular_spectrum_length": 4, "element_size": 4}}
    tree = {0: {"max": ["granularity"]}, 1: {"max": ["granularity", "mean"]}}
    inst, res = process_tree_masks(tree, f["cells"], f["pixels"][None], extract_tree, cp_measure_kwargs=kw)
    inst_o, res_o = ox.process_tree_masks(tree, f["cells"], f["pixels"][None], ox.extract_tree, cp_measure_kwargs=kw)
    assert [tuple(i) for i in inst] == [tuple(i) for i in inst_o] and len(res) == len(res_o)
    for a, b in zip(res, res_o):
        if isinstance(b, dict):
            assert list(a) == list(b)
            for k2 in b:
                assert np.allclose(a[k2], b[k2], rtol=1e-4, atol=1e-9, equal_nan=True), k2
        else:
            assert np.allclose(a, b, rtol=1e-4, atol=1e-9, equal_nan=True)
    with pytest.raises(NotImplementedError):
        process_tree_masks(tree, f["cells"], f["pixels"][None], extract_tree, cp_measure_kwargs={"granularity": {"image_mask": "objects"}})


def test_radial_zernikes_of_several_channels_in_one_launch(engine):
    """aliby_features_radial_zernikes_multi (the channel-independent Zernike basis evaluated once per pixel) gives the bits of
    one aliby_features_zernike(weighted) launch per channel, for every channel count and for float pixels."""
    import torch
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16

    fovs = [synth.make_fov(2, 60 + i, shape=(256, 288), n_channels=5, n_target=20) for i in range(2)]
    labels = np.stack([f["nuclei"] for f in fovs])
    labels[1][labels[1] == 3] = 0  # an absent label: NaN row
    planes_u16 = np.stack([f["pixels"][:, 0] for f in fovs])
    for planes in (planes_u16, (planes_u16 / 65535.0).astype(np.float32)):
        dl = to_device_u16(labels)
        dp, dt = to_device_planes(planes)
        tab = engine.object_table(dl)
        for chans in ((0, 1), (4, 2, 0), (0, 1, 2, 3), (3, 1, 4, 0, 2), (0, 1, 2, 3, 4, 0, 2), (1,), (0, 1, 2, 3, 4, 1)):
            ref = engine.new_output(tab.n_obj, 60 * len(chans) + 3)
            got = engine.new_output(tab.n_obj, 60 * len(chans) + 3)
            ref.fill_(-3.0)
            got.fill_(-3.0)
            cols = [1 + 60 * k for k in range(len(chans))]
            for ch, c0 in zip(chans, cols):
                engine.zernike(dl, dp, dt, ch, tab, ref, c0, True)
            engine.radial_zernikes_multi(dl, dp, dt, chans, tab, got, cols)
            torch.cuda.synchronize()
            a, b = ref.cpu().numpy(), got.cpu().numpy()
            assert np.array_equal(np.nan_to_num(a, nan=-9.0), np.nan_to_num(b, nan=-9.0)), chans
            assert np.isnan(b).any() and (b[:, 0] == -3.0).all() and (b[:, -2:] == -3.0).all()
