"""Randomised differential run of the position-batched runner against one run_pipeline_and_post per position: random numbers of
positions, frame shapes, channels, Z, feature selections, batch sizes, empty positions.  Integer columns and label images exact,
float columns to 1e-9 of the column's scale (a batch sizes its workgroups for its largest object: INTEGRATION.md 2.1).
usage: python tests/fuzz/fuzz_runner.py [first_seed=0] [n=20]     (GPU box)"""
import shutil
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from aliby_amd import synth  # noqa: E402
from aliby_amd.parallel import run_positions  # noqa: E402
from aliby_amd.pipe import run_pipeline_and_post  # noqa: E402
from aliby_amd.pipe_builder import build_pipeline_steps  # noqa: E402
from test_gpu_configs import _keyed_override  # noqa: E402

warnings.simplefilter("ignore")
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
TIMELAPSE = len(sys.argv) > 3 and sys.argv[3] == "timelapse"  # positions with several time points and a tracker step, in lockstep


def compare(seed, names, got, want, tmp, step, ntps):
    for i, nm in enumerate(names):
        g, w = got[i][0], want[i]
        assert g.schema.equals(w.schema) and g.num_rows == w.num_rows, (seed, nm, "shape of the table")
        for c in w.column_names:
            x, y = g[c].to_numpy(zero_copy_only=False), w[c].to_numpy(zero_copy_only=False)
            if x.dtype.kind != "f":
                assert np.array_equal(x, y), (seed, nm, c)
            elif len(x):
                scale = max(float(np.nanmax(np.abs(y))) if np.isfinite(y).any() else 0.0, 1.0)
                assert np.array_equal(np.isnan(x), np.isnan(y)) and np.allclose(x, y, rtol=0, atol=1e-9 * scale, equal_nan=True), (seed, nm, c)
        for t in range(ntps):
            with np.load(tmp / "batched" / "steps" / nm / step / f"{t:04d}.npz") as za, \
                    np.load(tmp / "single" / "steps" / nm / step / f"{t:04d}.npz") as zb:
                assert np.array_equal(za["arr_0"], zb["arr_0"]), (seed, nm, "masks", t)


def timelapse(seed):
    rng = np.random.default_rng(13000 + seed)
    npos, T = int(rng.integers(1, 6)), int(rng.integers(2, 5))
    shape = [(160, 192), (128, 128)][int(rng.integers(0, 2))]
    C, Z = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    frames = [[synth.make_fov(4, 30000 + 101 * seed + 7 * i + t, shape=shape, n_channels=C, n_z=Z, n_target=int(rng.integers(3, 10)))
               for t in range(T)] for i in range(npos)]
    override = _keyed_override([f for pos in frames for f in pos], key_channel=0)
    tree = {"None": {"None": ["sizeshape"]}, 0: {"max": ["intensity"]}}
    thr = float(rng.choice([0.25, 0.1, 0.5]))
    retain = {"tile": 1, "segment_cells": 2} if rng.random() < 0.5 else None

    def pipelines():
        out = []
        for pos in frames:
            p = {"ntps": T,
                 "steps": {"tile": {"image_kwargs": {"source": np.stack([f["pixels"] for f in pos])}, "tile_size": None},
                           "segment_cells": {"segmenter_kwargs": {"kind": "cellpose", "per_tile": True, "setup_params": {"flows_override": override}},
                                             "channel_to_segment": 0},
                           "track": {"kind": "stitch", "stitch_threshold": thr},
                           "extract_cells": {"tree": tree}},
                 "passed_data": {"track": [("masks", "segment_cells"), ("track_info", "track")],
                                 "extract_cells": [("masks", "segment_cells"), ("pixels", "tile")]},
                 "passed_methods": {"segment_cells": ("tile", "get_fczyx")},
                 "save": ("segment_cells",), "save_interval": 1}
            if retain:
                p["retain"] = dict(retain)
            out.append(p)
        return out

    tmp = Path(tempfile.mkdtemp(prefix="aliby_fuzz_"))
    names = [f"t{seed}_{i}" for i in range(npos)]
    want = [run_pipeline_and_post(pipeline=p, pipeline_name=nm, output_path=tmp / "single")[0] for p, nm in zip(pipelines(), names)]
    bs = int(rng.integers(1, npos + 1))
    got = run_positions(pipelines(), names, tmp / "batched", batch_size=bs)
    compare(seed, names, got, want, tmp, "segment_cells", T)
    shutil.rmtree(tmp, ignore_errors=True)
    print(f"seed {seed}: {npos} positions x T={T} of {shape} x {C} ch x Z={Z}, tracker at {thr}, retain {retain}, batches of {bs}: ok", flush=True)


if TIMELAPSE:
    for seed in range(first, first + n):
        timelapse(seed)
    n = 0  # (nothing left for the single-time-point loop below)
FEATS = ("sizeshape", "intensity", "texture", "radial_distribution", "zernike", "feret", "radial_zernikes")
for seed in range(first, first + n):
    rng = np.random.default_rng(11000 + seed)
    npos = int(rng.integers(1, 8))
    shape = [(160, 192), (224, 256), (128, 128)][int(rng.integers(0, 3))]
    C, Z = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    fovs = []
    for i in range(npos):
        f = synth.make_fov(2, 20000 + 37 * seed + i, shape=shape, n_channels=C, n_z=Z, n_target=int(rng.integers(3, 14)))
        if rng.random() < 0.15:  # a position without objects
            f["nuclei"] = np.zeros(shape, np.uint16)
        fovs.append(f)
    seg = int(rng.integers(0, C))
    override = _keyed_override(fovs, key_channel=seg)
    feats = tuple(sorted(set(rng.choice(FEATS, int(rng.integers(1, 5))))))
    chans = sorted(set(int(c) for c in rng.choice(np.arange(C), int(rng.integers(1, C + 1)))))

    def pipelines():
        out = []
        for f in fovs:
            p = build_pipeline_steps(channels_to_segment={"nuclei": seg}, channels_to_extract=chans, features_to_extract=feats)
            p["steps"]["tile"]["image_kwargs"] = {"source": f["pixels"][None]}
            p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override)
            out.append(p)
        return out

    tmp = Path(tempfile.mkdtemp(prefix="aliby_fuzz_"))
    names = [f"s{seed}_{i}" for i in range(npos)]
    want = [run_pipeline_and_post(pipeline=p, pipeline_name=nm, output_path=tmp / "single")[0] for p, nm in zip(pipelines(), names)]
    bs = int(rng.integers(1, npos + 1))
    got = run_positions(pipelines(), names, tmp / "batched", batch_size=bs)
    for i, nm in enumerate(names):
        g, w = got[i][0], want[i]
        assert g.schema.equals(w.schema) and g.num_rows == w.num_rows, (seed, nm, "shape of the table")
        for c in w.column_names:
            x, y = g[c].to_numpy(zero_copy_only=False), w[c].to_numpy(zero_copy_only=False)
            if x.dtype.kind != "f":
                assert np.array_equal(x, y), (seed, nm, c)
            elif len(x):
                scale = max(float(np.nanmax(np.abs(y))) if np.isfinite(y).any() else 0.0, 1.0)
                assert np.array_equal(np.isnan(x), np.isnan(y)) and np.allclose(x, y, rtol=0, atol=1e-9 * scale, equal_nan=True), (seed, nm, c)
        with np.load(tmp / "batched" / "steps" / nm / "segment_nuclei" / "0000.npz") as za, \
                np.load(tmp / "single" / "steps" / nm / "segment_nuclei" / "0000.npz") as zb:
            assert np.array_equal(za["arr_0"], zb["arr_0"]), (seed, nm, "masks")
    shutil.rmtree(tmp, ignore_errors=True)
    print(f"seed {seed}: {npos} positions of {shape} x {C} ch x Z={Z}, features {feats} on channels {chans}, batches of {bs}: ok", flush=True)
