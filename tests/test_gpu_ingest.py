"""
Image ingest on the GPU: TIFF planes decoded by csrc/ingest.hip's host pool into pinned memory and
uploaded plane by plane, behind the reference's ImageList / DatasetDir interface and its examples/01 calling pattern.
"""

from copy import deepcopy
from pathlib import Path

import numpy as np
import pytest

from aliby_amd import synth

pytestmark = pytest.mark.gpu

REGEX, ORDER = ".*/([^/]+)/.+_([0-9]{6})_([A-Za-z0-9]+)_(?:.*_)?([0-9]+).tif", "FTCZ"


def _write_position(root: Path, pos: str, tczyx: np.ndarray, channels):
    (root / pos).mkdir(parents=True)
    k = 0
    for t in range(tczyx.shape[0]):
        for c, ch in enumerate(channels):
            for z in range(tczyx.shape[2]):
                synth.write_tiff(root / pos / f"expt_{t:06d}_{ch}_{z + 1:03d}.tif", tczyx[t, c, z],
                                 compression="deflate" if k % 2 else None, rows_per_strip=32 if k % 3 else 50)
                k += 1


def test_tiler_reads_time_points_through_the_device_ingest(tmp_path, engine):
    from aliby_amd.io.dataset import DatasetDir
    from aliby_amd.io.image import ImageList
    from aliby_amd.tile.tiler import Tiler, TilerParameters

    rng = np.random.default_rng(3)
    tczyx = rng.integers(0, 65535, (4, 2, 3, 150, 130)).astype(np.uint16)
    _write_position(tmp_path, "pos007", tczyx, ("Brightfield", "GFP"))
    position = DatasetDir(tmp_path, regex=REGEX, capture_order=ORDER).get_position_ids()[0]
    image = ImageList(source=position, regex=REGEX, capture_order=ORDER)
    assert image.data.shape == tczyx.shape
    dev = image.data.read_device(2, engine.ctx.handle, None)
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), tczyx[2])
    tiler = Tiler.from_image(image, TilerParameters(tile_size=None))
    for tp in (0, 1, 2, 3, 1):  # 1, 2, 3 arrive from the helper thread that decoded ahead; the last one is a miss
        out = tiler.run_tp(tp)["pixels"]
        assert out.shape == (1, 2, 3, 150, 130)
        assert np.array_equal(out[0], tczyx[tp]), tp
    assert tiler._ingest_pending is not None and tiler._ingest_pending[0] == 2
    tiler._ingest_pending[1].result()


def test_pipeline_from_tiff_files_equals_pipeline_from_array(tmp_path, engine):
    """examples/01's calling pattern (image_kwargs = {source: {key, path: [files]}, regex, capture_order}) over a
    config-4-like time-lapse; the profile must be the one the in-memory stack gives."""
    import torch
    from aliby_amd.io.dataset import dispatch_dataset
    from aliby_amd.pipe import run_pipeline_and_post

    T = 3
    fovs = [synth.make_fov(4, t, shape=(160, 192), n_channels=1, n_z=5, n_target=8 + 2 * t) for t in range(T)]
    tczyx = np.stack([f["pixels"] for f in fovs])
    flows = [synth.analytic_flows(f["nuclei"]) for f in fovs]
    _write_position(tmp_path / "data", "pos001", tczyx, ("Brightfield",))
    positions = dispatch_dataset(tmp_path / "data", regex=REGEX, capture_order=ORDER).get_position_ids()
    assert [p["key"] for p in positions] == ["pos001"] and len(positions[0]["path"]) == T * 5

    def pipeline_for(image_kwargs):
        calls = {"n": 0}

        def override(x):
            t = calls["n"]
            calls["n"] += 1
            return torch.from_numpy(flows[t][0][None]).cuda(), torch.from_numpy(flows[t][1][None]).cuda()

        return {
            "ntps": T,
            "steps": {
                "tile": {"image_kwargs": image_kwargs, "tile_size": None},
                "segment_cells": {"segmenter_kwargs": {"kind": "cellpose", "setup_params": {"flows_override": override}},
                                  "channel_to_segment": 0},
                "extract_cells": {"tree": {"None": {"None": ["sizeshape"]}, 0: {"max": ["intensity"]}}},
            },
            "passed_data": {"extract_cells": [("masks", "segment_cells"), ("pixels", "tile")]},
            "passed_methods": {"segment_cells": ("tile", "get_fczyx")},
            "save": ("segment_cells",),
            "save_interval": 1,
        }

    from_files, _ = run_pipeline_and_post(
        pipeline=pipeline_for({"source": {"key": positions[0]["key"], "path": positions[0]["path"]}, "regex": REGEX,
                               "capture_order": ORDER}),
        pipeline_name="pos001", output_path=tmp_path / "out_files")
    from_array, _ = run_pipeline_and_post(pipeline=pipeline_for({"source": tczyx}), pipeline_name="pos001",
                                          output_path=tmp_path / "out_array")
    assert from_files.num_rows == from_array.num_rows > 0
    assert from_files.schema.names == from_array.schema.names
    for name in from_files.schema.names:
        a, b = from_files[name].to_numpy(), from_array[name].to_numpy()
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), name
    for t in range(T):
        with np.load(tmp_path / "out_files" / "steps" / "pos001" / "segment_cells" / f"{t:04d}.npz") as a, \
                np.load(tmp_path / "out_array" / "steps" / "pos001" / "segment_cells" / f"{t:04d}.npz") as b:
            assert np.array_equal(a["arr_0"], b["arr_0"])
