"""
End-to-end on the GPU: build_pipeline_steps -> run_pipeline_and_post on a synthetic position, the
caller pattern of examples/01_cell_painting_tiff.py (two object sets, 5 channels, intensity without
edges + sizeshape, 10 channel pairs x 4 colocalisation metrics).  Checks the profile table against the
CPU oracle run in the reference's structure and the on-disk layout (profiles/<name>.parquet,
steps/<name>/<step>/<tp:04d>.npz).
"""

from copy import deepcopy

import numpy as np
import pyarrow.parquet
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu


def _position(shape=(192, 224), nt=12):
    f = synth.make_fov(2, 7, shape=shape, n_channels=5, n_target=nt)
    return f


def test_example01_pipeline_matches_oracle(tmp_path, engine):
    import torch
    from aliby_amd.pipe import run_pipeline_and_post
    from aliby_amd.pipe_builder import build_pipeline_steps
    from aliby_amd.extraction.extract import format_extraction
    from oracle import aliby_extract as ox
    from oracle import cellpose_restated as cr

    f = _position()
    flows = {k: synth.analytic_flows(f[k]) for k in ("nuclei", "cells")}

    def override_for(name):
        dP, prob = flows[name]
        return lambda x: (torch.from_numpy(dP[None]).cuda(), torch.from_numpy(prob[None]).cuda())

    base = build_pipeline_steps(
        channels_to_segment={"nuclei": 0, "cell": 3},
        channels_to_extract=[0, 1, 2, 3, 4],
        features_to_extract=("intensity", "sizeshape"),
        cp_measure_feature_kwargs={"intensity": {"edge_measurements": False}},
    )
    assert list(base["steps"]) == ["tile", "segment_nuclei", "segment_cell", "extract_nuclei", "extract_cell",
                                   "extractmulti_nuclei", "extractmulti_cell"]
    pipeline = deepcopy(base)
    pipeline["steps"]["tile"]["image_kwargs"] = {"source": f["pixels"][None]}  # T=1
    pipeline["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override_for("nuclei"))
    pipeline["steps"]["segment_cell"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override_for("cells"))
    profiles, post = run_pipeline_and_post(pipeline=pipeline, pipeline_name="A01__1", output_path=tmp_path, overwrite=False)

    # --- oracle: same steps in the reference's structure
    pixels = f["pixels"][None]
    masks = {"nuclei": cr.finish_labels(cr.compute_masks(*flows["nuclei"])),
             "cell": cr.finish_labels(cr.compute_masks(*flows["cells"]))}
    n_rows = int(masks["nuclei"].max()) + int(masks["cell"].max())
    assert profiles.num_rows == n_rows
    # the prose shape of examples/01:156-158: 4 metadata + 6*78 + 5*16 + 10*8 = 632 columns
    assert len(profiles.column_names) == 632
    assert {"metadata_tile", "metadata_label", "metadata_object", "metadata_tp"} <= set(profiles.column_names)
    assert "0/max/intensity/Intensity_IntegratedIntensity" in profiles.column_names
    assert "(0, 3)/None/max/pearson/Correlation_Pearson" in profiles.column_names
    kw = {"intensity": {"edge_measurements": False}}
    got = profiles.to_pandas().set_index(["metadata_object", "metadata_label"]).sort_index()
    for obj in ("nuclei", "cell"):
        t1 = format_extraction(ox.process_tree_masks(base["steps"][f"extract_{obj}"]["tree"], masks[obj], pixels,
                                                     ox.extract_tree, cp_measure_kwargs=kw))
        t2 = format_extraction(ox.process_tree_masks(base["steps"][f"extractmulti_{obj}"]["tree"], masks[obj], pixels,
                                                     ox.extract_tree_multi, cp_measure_kwargs=kw))
        want = t1.to_pandas().merge(t2.to_pandas(), on=["tile", "label"]).set_index("label").sort_index()
        sub = got.loc[obj]
        assert len(sub) == len(want)
        for col in want.columns:
            if col == "tile":
                continue
            a, b = sub[col].to_numpy(float), want[col].to_numpy(float)
            if col.endswith("Orientation"):
                flip = np.isclose(np.abs(a), 45.0) & np.isclose(np.abs(b), 45.0)
                a, b = a[~flip], b[~flip]
            assert np.allclose(a, b, rtol=1e-4, atol=1e-8, equal_nan=True), (obj, col, a[:3], b[:3])

    # --- on-disk layout
    pq = tmp_path / "profiles" / "A01__1.parquet"
    assert pq.exists()
    assert pyarrow.parquet.read_table(pq).num_rows == n_rows
    for step, key in (("segment_nuclei", "nuclei"), ("segment_cell", "cell")):
        with np.load(tmp_path / "steps" / "A01__1" / step / "0000.npz") as z:
            assert list(z.keys()) == ["arr_0"]
            assert np.array_equal(z["arr_0"], masks[key])
    # resume-by-skip (pipe_core.py:408,446-448)
    again, post2 = run_pipeline_and_post(pipeline=deepcopy(pipeline), pipeline_name="A01__1", output_path=tmp_path,
                                         overwrite=False)
    assert again is None and post2 is None


def test_default_builder_pipeline_runs_all_families(tmp_path, engine):
    """The builder's default feature list (pipe_builder.py:49-56) on one object set: every family present."""
    import torch
    from aliby_amd.pipe import run_pipeline_and_post
    from aliby_amd.pipe_builder import build_pipeline_steps

    f = _position(shape=(160, 176), nt=9)
    dP, prob = synth.analytic_flows(f["nuclei"])
    pipeline = build_pipeline_steps(channels_to_segment={"nuclei": 0}, channels_to_extract=[0, 1])
    pipeline["steps"]["tile"]["image_kwargs"] = {"source": f["pixels"][None, :2]}
    pipeline["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(
        flows_override=lambda x: (torch.from_numpy(dP[None]).cuda(), torch.from_numpy(prob[None]).cuda()))
    profiles, _ = run_pipeline_and_post(pipeline=pipeline, pipeline_name="B02__1", output_path=tmp_path)
    cols = profiles.column_names
    per_channel = 60 + 21 + 2 + 52 + 12 + 30  # radial_zernikes, intensity, feret, texture, radial_distribution, zernike
    assert len(cols) == 4 + 78 + 2 * per_channel + 1 * 8
    assert profiles.num_rows == int(f["nuclei"].max())
    for key in ("0/max/texture/Contrast_3_00_256", "1/max/zernike/Zernike_9_9",
                "0/max/radial_distribution/RadialDistribution_FracAtD_1of4",
                "1/max/radial_zernikes/RadialDistribution_ZernikePhase_2_2", "None/None/sizeshape/Area"):
        assert key in cols, key


def test_timelapse_zstack_pipeline_config4_like(tmp_path, engine):
    """Config-4-like (BASELINE.json configs[3]): T time points, Z=5, one channel, monotile; every tp is tiled,
    segmented, written and measured; profiles carry metadata_tp; retain trims the history
    (pipe_core.py:245-249)."""
    import torch
    from aliby_amd.pipe import run_pipeline_and_post
    from aliby_amd.extraction.extract import format_extraction
    from oracle import aliby_extract as ox
    from oracle import cellpose_restated as cr

    T = 3
    fovs = [synth.make_fov(4, t, shape=(160, 192), n_channels=1, n_z=5, n_target=8 + 2 * t) for t in range(T)]
    tczyx = np.stack([f["pixels"] for f in fovs])  # [T,1,5,Y,X]
    flows = [synth.analytic_flows(f["nuclei"]) for f in fovs]
    calls = {"n": 0}

    def override(x):
        t = calls["n"]
        calls["n"] += 1
        return torch.from_numpy(flows[t][0][None]).cuda(), torch.from_numpy(flows[t][1][None]).cuda()

    tree = {"None": {"None": ["sizeshape"]}, 0: {"max": ["intensity"]}}
    pipeline = {
        "ntps": T,
        "steps": {
            "tile": {"image_kwargs": {"source": tczyx}, "tile_size": None},
            "segment_cells": {"segmenter_kwargs": {"kind": "cellpose", "setup_params": {"flows_override": override}},
                              "channel_to_segment": 0},
            "extract_cells": {"tree": tree},
        },
        "passed_data": {"extract_cells": [("masks", "segment_cells"), ("pixels", "tile")]},
        "passed_methods": {"segment_cells": ("tile", "get_fczyx")},
        "save": ("segment_cells",),
        "save_interval": 1,
        "retain": {"tile": 1},
    }
    profiles, _ = run_pipeline_and_post(pipeline=pipeline, pipeline_name="pos001", output_path=tmp_path)
    assert calls["n"] == T
    got = profiles.to_pandas()
    assert sorted(got["metadata_tp"].unique().tolist()) == list(range(T))
    for t in range(T):
        want_mask = cr.finish_labels(cr.compute_masks(*flows[t]))
        with np.load(tmp_path / "steps" / "pos001" / "segment_cells" / f"{t:04d}.npz") as z:
            assert np.array_equal(z["arr_0"], want_mask)
        want = format_extraction(ox.process_tree_masks(tree, want_mask, tczyx[t][None], ox.extract_tree)).to_pandas()
        sub = got[got["metadata_tp"] == t].sort_values("metadata_label")
        assert len(sub) == len(want) == int(want_mask.max())
        for col in want.columns:
            if col in ("tile", "label") or col.endswith("Orientation"):
                continue
            assert np.allclose(sub[col].to_numpy(float), want[col].to_numpy(float), rtol=1e-4, atol=1e-8, equal_nan=True), (t, col)


def test_deep_zstack_projection_config5_like(engine):
    """Config-5-like (BASELINE.json configs[4]): Z=32, 2 channels.  As wired by the reference, pixels are
    max-projected for segmentation (dispatch.py:199-206) and for every feature ("max" reducer), so the
    reference-faithful result is 2-D; true 3-D features are beyond the reference (SURVEY.md §8d, C5 note)."""
    import torch
    from aliby_amd.extraction.extract import extract_tree, process_tree_masks
    from aliby_amd.segment.dispatch import dispatch_segmenter
    from oracle import aliby_extract as ox
    from oracle import cellpose_restated as cr

    f = synth.make_fov(5, 0, shape=(128, 160), n_channels=2, n_z=32, n_target=8)
    dP, prob = synth.analytic_flows(f["nuclei"])
    seen = {}

    def override(x):
        seen["x"] = x.cpu().numpy()
        return torch.from_numpy(dP[None]).cuda(), torch.from_numpy(prob[None]).cuda()

    segment = dispatch_segmenter(kind="cellpose", channel_to_segment=0, setup_params=dict(flows_override=override))
    pixels = f["pixels"][None]  # [1,2,32,Y,X]
    labels = segment(pixels)
    assert np.array_equal(seen["x"][0], pixels[0, 0].max(axis=0))
    assert np.array_equal(labels, cr.finish_labels(cr.compute_masks(dP, prob)))
    tree = {1: {"max": ["intensity", "radial_distribution"]}, 0: {"add": ["intensity"]}}
    inst, res = process_tree_masks(tree, labels, pixels, extract_tree)
    inst_o, res_o = ox.process_tree_masks(tree, labels, pixels, ox.extract_tree)
    assert inst == inst_o
    for a, b in zip(res, res_o):
        for k in b:
            assert np.allclose(a[k], b[k], rtol=1e-4, atol=1e-8, equal_nan=True), k


def test_config5_from_a_blosc_zarr_group_equals_the_array_source(tmp_path, engine):
    """BASELINE config 5 as stated: a Zarr-backed Z-stack through the pipeline.  The store is a zarr v2 group with zarr's default
    compressor (Blosc lz4 + byte shuffle; written by c-blosc in the build container, tests/golden/make_blosc_fixtures.py), opened
    by ImageZarr (reference: src/aliby/io/image.py:236-264) and decoded by csrc/ingest.hip: profiles and masks come out bit for
    bit as from the same stack handed over as an array."""
    from pathlib import Path

    import pyarrow.parquet as pq
    import torch

    from aliby_amd.pipe import run_pipeline_and_post
    from aliby_amd.pipe_builder import build_pipeline_steps

    store = Path(__file__).parent / "golden" / "blosc"
    with np.load(store / "c5_expected.npz") as z:
        pixels, nuclei = z["pixels"], z["nuclei"]
    dP, prob = synth.analytic_flows(nuclei)

    def override(x):
        return torch.from_numpy(dP[None]).cuda(), torch.from_numpy(prob[None]).cuda()

    def pipeline(image_kwargs):
        p = build_pipeline_steps(channels_to_segment={"nuclei": 0}, channels_to_extract=[0, 1], features_to_extract=("intensity", "texture"))
        p["steps"]["tile"]["image_kwargs"] = image_kwargs
        p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override)
        return p

    got, _ = run_pipeline_and_post(pipeline=pipeline({"source": {"path": store / "c5.zarr", "key": "0"}, "capture_order": "TCZYX"}), pipeline_name="zarr", output_path=tmp_path)
    want, _ = run_pipeline_and_post(pipeline=pipeline({"source": pixels}), pipeline_name="array", output_path=tmp_path)
    assert got.num_rows == want.num_rows == int(nuclei.max()) > 0 and got.schema.equals(want.schema)
    for c in got.column_names:
        a, b = got[c].to_numpy(zero_copy_only=False), want[c].to_numpy(zero_copy_only=False)
        assert np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")), c
    assert (tmp_path / "profiles" / "zarr.parquet").read_bytes() == (tmp_path / "profiles" / "array.parquet").read_bytes()
    with np.load(tmp_path / "steps" / "zarr" / "segment_nuclei" / "0000.npz") as za, np.load(tmp_path / "steps" / "array" / "segment_nuclei" / "0000.npz") as zb:
        assert np.array_equal(za["arr_0"], zb["arr_0"]) and int(za["arr_0"].max()) == int(nuclei.max())
    assert pq.read_table(tmp_path / "profiles" / "zarr.parquet").num_columns == got.num_columns


def test_cell_painting_example_script_runs_from_tiffs(tmp_path, engine, monkeypatch, capsys):
    """examples/cell_painting_tiff.py: TIFF directory -> DatasetDir positions -> run_positions, the caller pattern of the
    reference's examples/01 with the joblib loop replaced; the script itself checks that every synthetic nucleus is one row."""
    import runpy
    import sys
    from pathlib import Path

    script = Path(__file__).resolve().parents[1] / "examples" / "cell_painting_tiff.py"
    monkeypatch.setattr(sys, "argv", [str(script), "--wells", "3", "--fields", "2", "--size", "256", "--out", str(tmp_path / "out"), "--batch-size", "4"])
    runpy.run_path(str(script), run_name="__main__")
    out = capsys.readouterr().out
    assert "6 parquet files" in out and "every synthetic nucleus is one row" in out
    assert len(list((tmp_path / "out" / "profiles").glob("*.parquet"))) == 6


def test_timelapse_example_script_runs_from_a_zarr_store(tmp_path, engine, monkeypatch, capsys):
    """examples/yeast_timelapse_zarr.py: zarr store -> DatasetZarr positions -> trap tiles + drift -> per-tile segmentation (an
    intensity-gradient flow field through `flows_override`) -> stitch tracker -> features, the data side of the reference's
    examples/03 with this build's segmenter in BABY's place."""
    import runpy
    import sys
    from pathlib import Path

    script = Path(__file__).resolve().parents[1] / "examples" / "yeast_timelapse_zarr.py"
    monkeypatch.setattr(sys, "argv", [str(script), "--positions", "2", "--tps", "3", "--out", str(tmp_path / "out")])
    runpy.run_path(str(script), run_name="__main__")
    out = capsys.readouterr().out
    assert "pos000:" in out and "pos001:" in out and "9 trap tiles" in out
    files = sorted((tmp_path / "out" / "profiles").glob("*.parquet"))
    assert [f.stem for f in files] == ["pos000", "pos001"]
    import pyarrow.parquet as pq

    t = pq.read_table(files[0]).to_pandas()
    assert sorted(t["metadata_tp"].unique().tolist()) == [0, 1, 2] and len(t) > 0

