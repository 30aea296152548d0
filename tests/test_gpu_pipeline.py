"""
GPU parity of the host-facing API (tiler, reduce_z, relabel, process_tree_masks, format_extraction)
against the CPU oracle, through the C ABI.
"""

import numpy as np
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu


def test_tiler_monotile_and_padded_tiles(engine):
    from aliby_amd.tile.tiler import ImageArray, Tiler, TilerParameters
    from oracle import tiler_ref

    f = synth.make_fov(4, 0, shape=(200, 232), n_channels=2, n_z=3, n_target=12)
    tczyx = f["pixels"][None]  # T=1
    mono = Tiler.from_image(ImageArray(tczyx), TilerParameters(tile_size=None))
    out = mono.run_tp(0)
    assert out["pixels"].shape == (1, 2, 3, 200, 232) and out["pixels"].dtype == np.uint16
    assert np.array_equal(out["pixels"][0], f["pixels"])
    assert np.array_equal(mono.get_fczyx(0), out["pixels"])
    # tiles: inside, partially outside (median pad) and mostly outside (NaN tile)
    centres = [(100, 100), (40, 60), (150, 160), (30, 60), (100, 199)]
    tiled = Tiler.from_image(ImageArray(tczyx), TilerParameters(tile_size=65), trap_locations=centres)
    out = tiled.run_tp(0)
    assert len(tiled.tile_locs) == 3  # the last two are dropped: too close to the edge (tiler.py:686-691)
    ranges = [t.as_range(0) for t in tiled.tile_locs]
    want = tiler_ref.get_fczyx(f["pixels"], ranges)
    assert out["pixels"].shape == want.shape
    assert np.array_equal(out["pixels"], want)
    # stage drift pushes tiles over the border: median pad and (>25% outside) NaN tile
    tiled.tile_locs.drifts[0] = [20.4, -15.9]
    ranges = [t.as_range(0) for t in tiled.tile_locs]
    want = tiler_ref.get_fczyx(f["pixels"], ranges)
    got = tiled.get_fczyx(0)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got.astype(float)), np.isnan(want.astype(float)))
    assert np.array_equal(np.nan_to_num(got.astype(float)), np.nan_to_num(want.astype(float)))
    tiled.tile_locs.drifts[0] = [-40.0, 0.0]
    ranges = [t.as_range(0) for t in tiled.tile_locs]
    want = tiler_ref.get_fczyx(f["pixels"], ranges)
    got = tiled.get_fczyx(0)
    assert np.isnan(want).any() and got.dtype == np.float64
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want))


def test_crop_pad_median_semantics(engine):
    """Drive the stager directly with rectangles that leave the image on every side."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr
    from oracle import tiler_ref

    rng = np.random.default_rng(11)
    stack = rng.integers(0, 65535, size=(2, 2, 90, 110), dtype=np.uint16)
    h, w = 48, 40
    rects = np.array([[-10, 5, h, w], [50, 80, h, w], [-12, -9, h, w], [60, -8, h, w], [-20, 0, h, w], [0, 0, h, w]], np.int32)
    dev = torch.from_numpy(stack).cuda()
    out = torch.zeros((len(rects), 2, 2, h, w), dtype=torch.uint16, device="cuda")
    flags = np.zeros(len(rects), np.int32)
    _lib.check(engine.lib.aliby_crop_pad_u16(engine.ctx.handle, _ptr(dev), 2, 2, 90, 110, _ptr(rects), len(rects), h, w,
                                             _ptr(out), _ptr(flags), _stream_ptr()))
    got = out.cpu().numpy()
    ranges = [(slice(r[0], r[0] + h), slice(r[1], r[1] + w)) for r in rects]
    for f, rg in enumerate(ranges):
        for c in range(2):
            want = tiler_ref.if_out_of_bounds_pad(stack[c], rg)
            if np.isnan(want).any():
                assert flags[f] == 1
            else:
                assert flags[f] == 0
                assert np.array_equal(got[f, c], want.astype(np.uint16)), (f, c)
    # rect 3 pads 18 rows below: 18/0.25 > w=40 -> NaN through the reference's (padding/0.25 > tile_shape) broadcast
    assert flags.tolist() == [0, 0, 0, 1, 1, 0]


@pytest.mark.parametrize("op,name", [(0, "max"), (1, "add"), (2, "div")])
def test_reduce_z(engine, op, name):
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    rng = np.random.default_rng(5)
    a = rng.integers(1, 60000, size=(3, 4, 5, 33, 47), dtype=np.uint16)  # [F,C,Z,Y,X]
    dev = torch.from_numpy(a).cuda()
    ufunc = {"max": np.maximum, "add": np.add, "div": np.divide}[name]
    want = ufunc.reduce(a, axis=2)
    out_dt = _lib.U16 if op == 0 else _lib.F32
    out = torch.empty((3, 4, 33, 47), dtype=torch.uint16 if op == 0 else torch.float32, device="cuda")
    _lib.check(engine.lib.aliby_reduce_z(engine.ctx.handle, _ptr(dev), _lib.U16, 12, 5, 33 * 47, op, _ptr(out), out_dt, _stream_ptr()))
    got = out.cpu().numpy()
    if op == 0:
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want.astype(np.float64), rtol=1e-6)
    with pytest.raises(Exception):
        _lib.check(engine.lib.aliby_reduce_z(engine.ctx.handle, _ptr(dev), _lib.U16, 12, 5, 33 * 47, 7, _ptr(out), out_dt, _stream_ptr()))
    # the host-facing reduce_z (distributors.py:6-24) hands back NumPy's own result dtype and bits: uint16 / uint64 / float64
    from aliby_amd.extraction.functions import reduce_z

    for axis in (0, 1):
        got = reduce_z(a[0], ufunc, axis=axis)  # [C,Z,Y,X] reduced over C or over Z
        want = ufunc.reduce(a[0], axis=axis)
        assert got.dtype == want.dtype and np.array_equal(got, want), (name, axis, got.dtype, want.dtype)
    f = rng.random((4, 21, 17), dtype=np.float32) + 0.5
    got = reduce_z(f, ufunc)
    assert got.dtype == np.float32 and np.array_equal(got, ufunc.reduce(f, axis=0))
    assert reduce_z(f[0], ufunc) is not None and reduce_z(f[0], ufunc).shape == (21, 17)
    with pytest.raises(Exception, match="invalid reducer"):
        reduce_z(a[0], np.mean)


def test_relabel_sequential(engine):
    import torch
    from oracle import tiler_ref

    f = synth.make_fov(1, 3, shape=(128, 160), n_target=14)
    lab = f["cells"].astype(np.uint16) * 7
    lab[lab == 21] = 0
    lab2 = np.stack([lab, np.zeros_like(lab), (f["nuclei"] * 2).astype(np.uint16)])
    dev = torch.from_numpy(lab2.copy()).cuda()
    n = engine.relabel_sequential(dev)
    got = dev.cpu().numpy()
    for i in range(3):
        want = tiler_ref.relabel_sequential(lab2[i])
        assert np.array_equal(got[i], want)
        assert n[i] == want.max()


def test_process_tree_masks_matches_reference_structure(engine):
    """Same tree as tests/test_cellpose_cpmeasure_minimal.py:68-81 of the reference."""
    from aliby_amd.extraction.extract import extract_tree, format_extraction, process_tree_masks
    from oracle import aliby_extract as ox

    f = synth.make_fov(1, 0, shape=(256, 256), n_target=25)
    masks = f["nuclei"]  # not a list: "hacky fix when tile level is not provided"
    pixels = f["pixels"][None]
    tree = {"None": {"None": ["sizeshape"]}, 0: {"max": ["intensity"]}}
    inst, res = process_tree_masks(tree, masks, pixels, extract_tree, ncores=None)
    inst_o, res_o = ox.process_tree_masks(tree, masks, pixels, ox.extract_tree)
    assert inst == inst_o
    assert len(res) == len(res_o)
    for a, b in zip(res, res_o):
        assert list(a.keys()) == list(b.keys()) or set(a.keys()) == set(b.keys())
        for k in b:
            if k == "Orientation" and abs(abs(a[k][0]) - 45.0) < 1e-9 and abs(abs(b[k][0]) - 45.0) < 1e-9:
                continue  # isotropic inertia tensor: the sign is floating-point noise in the float restatement
            assert np.allclose(a[k], b[k], rtol=1e-4, atol=1e-9, equal_nan=True), k
    # wide table: fast columnar path == generic pivot over the oracle's results
    t_fast = format_extraction((inst, res))
    t_ref = format_extraction((inst_o, res_o))
    assert t_fast.column_names == t_ref.column_names
    assert t_fast.num_rows == t_ref.num_rows == int(masks.max())
    for name in t_ref.column_names:
        if name.endswith("Orientation"):
            continue
        assert np.allclose(np.asarray(t_fast[name].to_numpy(zero_copy_only=False), float),
                           np.asarray(t_ref[name].to_numpy(zero_copy_only=False), float), rtol=1e-4, atol=1e-9, equal_nan=True)


def test_empty_and_ragged_inputs(engine):
    from aliby_amd.extraction.extract import extract_tree, format_extraction, process_tree_masks

    tree = {"None": {"None": ["sizeshape"]}, 0: {"max": ["intensity"]}}
    empty = np.zeros((64, 64), np.uint16)
    pixels = np.zeros((1, 1, 1, 64, 64), np.uint16)
    inst, res = process_tree_masks(tree, [empty], pixels, extract_tree)
    assert inst == () and len(res) == 0
    t = format_extraction((inst, res))
    assert t.num_rows == 0 and t.column_names == ["tile", "label"]
    # a label gap: labels {1,3} -> rows for 1..3, the absent one is NaN (the reference's all-False mask)
    gap = np.zeros((64, 64), np.uint16)
    gap[5:15, 5:15] = 1
    gap[30:40, 30:50] = 3
    px = np.random.default_rng(0).integers(0, 5000, size=(1, 1, 1, 64, 64), dtype=np.uint16)
    inst, res = process_tree_masks(tree, [gap], px, extract_tree)
    assert len(res) == 3 * 2
    assert res[0]["Area"][0] == 100 and np.isnan(res[2]["Area"][0]) and res[4]["Area"][0] == 200
    # single-pixel and full-frame objects
    odd = np.zeros((2, 40, 40), np.uint16)
    odd[0, 7, 9] = 1
    odd[1] = 1
    px = np.random.default_rng(1).integers(1, 5000, size=(2, 1, 1, 40, 40), dtype=np.uint16)
    inst, res = process_tree_masks(tree, [odd[0], odd[1]], px, extract_tree)
    from oracle import aliby_extract as ox
    inst_o, res_o = ox.process_tree_masks(tree, [odd[0], odd[1]], px, ox.extract_tree)
    for a, b in zip(res, res_o):
        for k in b:
            assert np.allclose(a[k], b[k], rtol=1e-4, atol=1e-9, equal_nan=True), (k, a[k], b[k])


def test_cell_py_metrics_match_reference_restatement(engine):
    """The reference's in-repo metrics (cell.py) through process_tree_masks; the oracle for them is
    pinned against the reference module itself (tests/test_oracle_golden.py)."""
    from aliby_amd.extraction.extract import extract_tree, format_extraction, process_tree_masks
    from oracle import aliby_extract as ox

    f = synth.make_fov(1, 6, shape=(224, 256), n_target=22)
    masks = [f["cells"]]
    pixels = f["pixels"][None]
    tree = {"None": {"None": ["area", "centroid_x", "centroid_y", "volume", "conical_volume", "spherical_volume", "eccentricity"]},
            1: {"max": ["mean", "median", "std", "total", "total_squared", "max2p5pc", "max5px_median", "moment_of_inertia"]}}
    inst, res = process_tree_masks(tree, masks, pixels, extract_tree)
    inst_o, res_o = ox.process_tree_masks(tree, masks, pixels, ox.extract_tree)
    assert inst == inst_o and len(res) == len(res_o)
    for i, (a, b) in enumerate(zip(res, res_o)):
        assert isinstance(a, float)
        assert np.isclose(a, float(b), rtol=1e-4, atol=1e-9, equal_nan=True), (inst[i], a, b)
    t = format_extraction((inst, res))
    assert "None/None/area/area" in t.column_names and "1/max/max2p5pc/max2p5pc" in t.column_names
    assert t.num_rows == int(masks[0].max())
    with pytest.raises(KeyError):
        process_tree_masks({"None": {"None": ["no_such_metric"]}}, masks, pixels, extract_tree)
    with pytest.raises(Exception, match="invalid reducer"):
        process_tree_masks({0: {"mean": ["intensity"]}}, masks, pixels, extract_tree)


def _blob_frames(seed, n_frames=4, shape=(96, 128), n_cells=9):
    """Label images of drifting / appearing / vanishing discs (labels renumbered per frame like a segmenter would)."""
    rng = np.random.default_rng(seed)
    cy = rng.uniform(12, shape[0] - 12, n_cells); cx = rng.uniform(12, shape[1] - 12, n_cells); r = rng.uniform(5, 9, n_cells)
    yy, xx = np.mgrid[: shape[0], : shape[1]]
    frames = []
    for t in range(n_frames):
        alive = rng.random(n_cells) > 0.15
        lab = np.zeros(shape, np.uint16)
        order = rng.permutation(n_cells)
        k = 0
        for i in order:
            if not alive[i]:
                continue
            m = (yy - cy[i]) ** 2 + (xx - cx[i]) ** 2 <= r[i] ** 2
            if (lab[m] > 0).any():
                continue
            k += 1
            lab[m] = k
        frames.append(lab)
        cy += rng.normal(0, 1.5, n_cells); cx += rng.normal(0, 1.5, n_cells)
    return frames


def test_track_stitch_matches_oracle_over_a_time_lapse(engine):
    """IoU stitcher through the C ABI vs the CPU restatement, tile-batched, state carried like the reference's track step."""
    from aliby_amd.track.stitch import StitchTracker
    from oracle.track_restated import stitch_rois as oracle_rois

    tiles = [_blob_frames(s) for s in (1, 2, 3)]  # 3 tiles x 4 frames
    trk = StitchTracker(engine=engine)
    info_gpu = info_cpu = None
    first = trk([[tile[0]] for tile in tiles])
    assert all(first[k]["labels"] == list(range(1, int(tiles[k][0].max()) + 1)) for k in range(3))
    for t in range(1, 4):
        masks = [[tile[t - 1], tile[t]] for tile in tiles]
        info_gpu = trk(masks, info_gpu)
        info_cpu = oracle_rois(masks, info_cpu)
        assert dict(info_gpu) == info_cpu, t
        assert info_gpu["track_info"] is info_gpu  # what the engine's passed_data lookup relies on
    # an identical frame pair keeps every label
    same = trk([[tiles[0][1], tiles[0][1]]])
    assert same[0]["labels"] == list(range(1, int(tiles[0][1].max()) + 1))


def test_track_stitch_edge_cases(engine):
    from aliby_amd.track.stitch import StitchTracker
    from oracle.track_restated import stitch_rois as oracle_rois

    trk = StitchTracker(engine=engine)
    z = np.zeros((32, 32), np.uint16)
    a = z.copy(); a[2:10, 2:10] = 1; a[20:30, 20:30] = 3  # label 2 absent from the frame
    b = z.copy(); b[3:11, 3:11] = 2; b[0:2, 20:30] = 1
    for masks in ([[z, z]], [[z, a]], [[a, z]], [[a, b]], [[b, a]]):
        assert dict(trk(masks)) == oracle_rois(masks), masks
    info = {0: {"labels": [5, 0, 9], "max_label": 12}}
    assert dict(trk([[a, b]], info)) == oracle_rois([[a, b]], info)
    with pytest.raises(ValueError):
        StitchTracker(stitch_threshold=0.005, engine=engine)([[a, b]])  # (below the 0.01 the reference's 3-D branch uses)


def test_track_step_wired_through_the_engine(tmp_path, engine):
    """The engine's own wiring (pipe_core.py:188-205): the `track` step gets the last two timepoints regrouped per tile
    and, through passed_data, its own previous result as `track_info`."""
    from aliby_amd.pipe import init_step
    from aliby_amd.pipe_core import run_pipeline_return_state
    from oracle.track_restated import stitch_rois as oracle_rois

    tiles = [_blob_frames(s, n_frames=3) for s in (7, 8)]
    clock = {"tp": 0}

    def fake_segmenter(**kw):  # per-tile label images of the current timepoint, like a multi-tile segmenter's output
        t = clock["tp"]
        clock["tp"] += 1
        return [tile[t] for tile in tiles]

    def init(step_name, parameters, other_steps=None):
        if step_name == "segment_cells":
            return fake_segmenter
        return init_step(step_name, parameters, other_steps)

    pipeline = {
        "ntps": 3,
        "steps": {"segment_cells": {}, "track": {"kind": "stitch", "stitch_threshold": 0.25}},
        "passed_data": {"track": [("masks", "segment_cells"), ("track_info", "track")]},
        "retain": {"segment_cells": 2},
    }
    state = run_pipeline_return_state(pipeline, tmp_path, init)
    got = state["data"]["track"]
    want_info = None
    for t in range(3):
        masks = [[tile[t]] for tile in tiles] if t == 0 else [[tile[t - 1], tile[t]] for tile in tiles]
        if t == 0:
            want = {k: {"labels": list(range(1, int(tiles[k][0].max()) + 1)), "max_label": int(tiles[k][0].max())} for k in range(2)}
        else:
            want = oracle_rois(masks, want_info)
        assert dict(got[t]) == want, t
        want_info = want


def test_drift_on_device_matches_fixture_and_moves_the_tiles(engine):
    """Phase cross-correlation on the GPU vs the scikit-image fixture / the oracle, and the Tiler using it: with
    calculate_drift the tile window follows the sample (tiles.py: centre - cumulative drift)."""
    import json
    from pathlib import Path

    from aliby_amd.tile.drift import phase_cross_correlation
    from aliby_amd.tile.tiler import Tiler, TilerParameters
    from oracle.drift_restated import phase_cross_correlation as oracle_pcc
    from tests.golden.make_golden import drift_cases

    fx = json.loads((Path(__file__).parent / "golden" / "skimage_drift.json").read_text())
    for (applied, ref, mov), rec in zip(drift_cases(), fx["cases"]):
        assert phase_cross_correlation(ref, mov, normalization=None).tolist() == rec["shift"]
        assert phase_cross_correlation(ref, mov).tolist() == oracle_pcc(ref, mov).tolist() == rec["shift"]
    # a time lapse whose content moves by (+2, -3) px per frame
    base = synth.make_fov(4, 0, shape=(160, 192), n_channels=1, n_z=1, n_target=10)["pixels"][:, 0]  # [C,Y,X]
    T = 3
    tczyx = np.stack([np.roll(base, (2 * t, -3 * t), axis=(1, 2)) for t in range(T)])[:, :, None]  # [T,C,1,Y,X]
    tiler = Tiler(tczyx, {}, TilerParameters.default(tile_size=64, ref_channel=0), trap_locations=[(80, 96)])
    tiler.calculate_drift = True
    crops = [tiler.run_tp(t)["pixels"] for t in range(T)]
    assert tiler.tile_locs.drifts == [[0.0, 0.0], [-2.0, 3.0], [-2.0, 3.0]]
    # drift-corrected tiles show the same piece of the sample at every timepoint
    assert all(np.array_equal(crops[0], c) for c in crops[1:])


def test_c_caller_runs_the_stager_without_python(tmp_path, engine):
    """examples/c_abi_demo.c: stage, crop with median padding and max-project Z from a plain C program."""
    import subprocess

    from tests.test_cpu_host import build_c_demo

    out = subprocess.run([str(build_c_demo(tmp_path))], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "tile0[0,0]=210 (expect 210)" in out.stdout and out.stdout.strip().endswith("ok")


def test_overlapping_mask_extraction_matches_oracle(engine):
    """§8f-4 second half: BABY-style layered masks — objects that overlap sit in different planes of a [stack, Y, X] array per
    tile, labels arrive non-sequential (extract.py:456-517, 156-197, 602-682).  process_tree_masks_overlap + extract_tree
    (overlap=True) against the oracle restatement; format_extraction_overlap restores the original labels."""
    from aliby_amd.extraction.extract import extract_tree, format_extraction, format_extraction_overlap, process_tree_masks_overlap
    from aliby_amd.pipe_core import _init_extract
    from functools import partial
    from oracle import aliby_extract as ox

    rng = np.random.default_rng(3)
    Y, X = 96, 117
    yy, xx = np.mgrid[:Y, :X]

    def disc(cy, cx, r):
        return (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r

    tile0 = np.zeros((2, Y, X), np.int32)
    tile0[0][disc(30, 30, 12)] = 7      # overlaps label 12 of the other plane
    tile0[0][disc(60, 80, 10)] = 3
    tile0[1][disc(36, 38, 11)] = 12
    tile0[1][disc(20, 95, 6)] = 40
    tile1 = np.zeros((3, Y, X), np.int32)
    tile1[0][disc(50, 50, 15)] = 2
    tile1[2][disc(55, 58, 14)] = 5      # plane 1 is empty
    masks = [tile0, tile1]
    pixels = rng.integers(200, 4000, size=(2, 2, 3, Y, X)).astype(np.uint16)  # [F, C, Z, Y, X]
    tree = {"None": {"None": ["sizeshape", "area"]}, 1: {"max": ["intensity", "mean"]}, 0: {"add": ["intensity"]}}

    inst, res = process_tree_masks_overlap(tree, masks, pixels, partial(extract_tree, overlap=True))
    inst_o, res_o, inv_o = ox.process_tree_masks_overlap(tree, masks, pixels)
    assert inst == inst_o
    objs = list(dict.fromkeys(t[0] for t in inst))
    assert objs == [(0, 0, 1), (0, 0, 2), (0, 1, 1), (0, 1, 2), (1, 0, 1), (1, 2, 1)]
    assert {k: dict(v) for k, v in res.inverse_mappings.items()} == inv_o
    assert res.inverse_mappings[0, 0][2] == 7 and res.inverse_mappings[0, 1][2] == 40 and res.inverse_mappings[1, 2][1] == 5
    assert len(res) == len(res_o) == len(inst)
    for a, b in zip(res, res_o):
        if isinstance(b, dict):
            for k in b:
                assert np.allclose(a[k], b[k], rtol=1e-4, atol=1e-8, equal_nan=True), k
        else:
            assert np.isclose(a, b, rtol=1e-4)
    # the engine's pivot (get_profiles_from_state calls format_extraction): keyed by (tile, relabelled id)
    t = format_extraction((inst, res)).to_pandas()
    assert sorted(zip(t["tile"], t["label"])) == [(0, 1), (0, 2), (1, 1)]  # ids collide across planes, as in the reference
    # the overlap-aware pivot: keyed by (tile, ORIGINAL label), one row per object
    t2 = format_extraction_overlap((inst, res, res.inverse_mappings)).to_pandas()
    assert list(zip(t2["metadata_tile"], t2["metadata_label"])) == [(0, 3), (0, 7), (0, 12), (0, 40), (1, 2), (1, 5)]
    want_area = [int((tile0[0] == 3).sum()), int((tile0[0] == 7).sum()), int((tile0[1] == 12).sum()), int((tile0[1] == 40).sum()),
                 int((tile1[0] == 2).sum()), int((tile1[2] == 5).sum())]
    assert t2["None/None/sizeshape/Area"].tolist() == want_area and t2["None/None/area/area"].tolist() == want_area
    # the step the BABY flavour's init_step builds (pipe_baby.py:79-80 -> pipe_core._init_extract(overlap=True))
    step = _init_extract("extract_cells", {"tree": tree}, overlap=True)
    inst2, res2 = step(masks=masks, pixels=pixels)
    assert inst2 == inst and all(np.allclose(list(a.values())[0] if isinstance(a, dict) else a, list(b.values())[0] if isinstance(b, dict) else b,
                                             equal_nan=True) for a, b in zip(res2, res))


def test_ratio_and_trap_functions_match_the_reference_module(engine):
    """cell.ratio, trap.imBackground and trap.background_max5 on the GPU against values produced by importing the reference's
    own modules (tests/golden/reference_leaves.json: the pinned part of the oracle), plus edge cases against the oracle."""
    import json
    from pathlib import Path

    from aliby_amd.extraction import functions as fn
    from aliby_amd.extraction.engine import to_device_planes, to_device_u16
    from oracle import cell_metrics as cm

    g = Path(__file__).parent / "golden"
    ref = json.loads((g / "reference_leaves.json").read_text())
    inputs = np.load(g / "inputs_c1_256.npz")
    lab = inputs["nuclei"]
    stack = np.stack([inputs["pixels"][0, 0], inputs["pixels"][1, 0]], -1).astype(float)
    want = [float("nan") if v is None else float(v) for v in ref["cell_ratio"]]
    got = [fn.ratio(lab == l, stack) for l in range(1, 6)]
    assert np.allclose(got, want, rtol=1e-12, equal_nan=True)
    masks3 = np.stack([lab == l for l in range(1, 4)], axis=2)
    img = inputs["pixels"][0, 0]
    assert fn.imBackground(masks3, img) == ref["trap"]["imBackground"]
    assert fn.background_max5(masks3, img) == ref["trap"]["background_max5"]
    # a 2-D plane (what the extraction tree hands over): NaN, as in the reference
    assert np.isnan(fn.ratio(lab == 1, img))
    # a zero in the denominator anywhere in the cell -> NaN; an even pixel count -> mean of the two middle ratios
    s2 = stack.copy()
    yy, xx = np.nonzero(lab == 2)
    s2[yy[0], xx[0], 1] = 0
    assert np.isnan(fn.ratio(lab == 2, s2)) and np.isnan(cm.ratio(lab == 2, s2))
    # batched forms over a label stack, every object / tile at once
    labels = to_device_u16(np.stack([lab, np.roll(lab, 5, 1)]))
    planes, dt = to_device_planes(np.stack([inputs["pixels"][:2, 0], inputs["pixels"][:2, 0]]))
    tab = engine.object_table(labels)
    r = engine.cell_ratio(labels, planes, dt, 0, 1, tab).cpu().numpy()
    n = int(lab.max())
    for l in (1, 3, n):
        assert np.isclose(r[l - 1], cm.ratio(lab == l, stack), rtol=1e-12, equal_nan=True)
    bg = engine.trap_background(labels, planes, dt, 1).cpu().numpy()
    for f, lf in enumerate((lab, np.roll(lab, 5, 1))):
        m = (lf > 0)[..., None]
        assert bg[f, 0] == cm.imBackground(m, inputs["pixels"][1, 0]) and bg[f, 1] == cm.background_max5(m, inputs["pixels"][1, 0])
    # float pixels, no masks at all, a fully covered tile
    fimg = (inputs["pixels"][0, 0] / 65535.0).astype(np.float32)
    assert np.isclose(fn.imBackground(np.zeros((0,)), fimg), np.median(fimg)) and np.isclose(fn.background_max5([], fimg), np.sort(fimg.ravel())[-5:].mean())
    assert np.isnan(fn.imBackground(np.ones(img.shape + (1,), bool), img))
    # "ratio" through the feature tree: a NaN column (the reference's behaviour on 2-D planes)
    from aliby_amd.extraction.extract import extract_tree, process_tree_masks

    inst, res = process_tree_masks({0: {"max": ["ratio", "mean"]}}, lab, inputs["pixels"][None, :, None][:, :, :, 0] if False else inputs["pixels"][None], extract_tree)
    assert np.isnan(res[0]) and np.isfinite(res[1])


@pytest.mark.gpu
def test_gather_rows_through_rccl_on_a_one_rank_group(engine):
    """The end-of-run exchange (aliby_amd/parallel.py gather_rows: all_gather of the row counts, padded gather of float64 rows and
    int64 metadata) on backend "nccl" = RCCL with device tensors.  A one-GPU box cannot host two RCCL ranks, so the group has one;
    the collectives still go through RCCL (communicator, kernels, dtypes), which the gloo world-size-2 test cannot show."""
    import socket

    import torch
    import torch.distributed as dist

    from aliby_amd import parallel

    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        vals = torch.arange(12, dtype=torch.float64, device="cuda").reshape(4, 3) / 7
        meta = torch.arange(16, dtype=torch.int64, device="cuda").reshape(4, 4)
        v, m = parallel.gather_rows(vals, meta, always_collective=True)
        assert v.is_cuda and torch.equal(v, vals) and torch.equal(m, meta)
        v0, m0 = parallel.gather_rows(vals[:0], meta[:0], always_collective=True)  # a rank without rows
        assert v0.shape == (0, 3) and m0.shape == (0, 4)
        parallel.barrier()
    finally:
        dist.destroy_process_group()
