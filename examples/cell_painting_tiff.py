"""
Cell Painting profiling on one MI355X: a directory of TIFFs -> Cellpose segmentation -> cp_measure features -> one parquet per
position, through the step API of the reference (`build_pipeline_steps` dicts) and `aliby_amd.parallel.run_positions`.

The reference's example for this workload (examples/01_cell_painting_tiff.py there) downloads a small fixture and fans
`run_pipeline_and_post` out over positions with joblib, one position per worker.  Neither is possible here (no network; one process
owns the GPU), so this script

  1. writes a synthetic plate — wells x fields x 5 channels, one uncompressed TIFF per channel, named
     <plate>__<well>__<field>__<channel>.tif — unless `--data` points at a directory of your own with that convention;
  2. groups the files into positions with `DatasetDir(regex, capture_order)`, as the reference does;
  3. stamps one pipeline dict per position and hands ALL of them to `run_positions`, which batches the device work
     (`batch_size` positions per launch sequence) and writes the files the per-position call would write.

Segmentation needs Cellpose weights; none are obtainable offline.  `--weights PATH` loads a CPnet checkpoint (cyto3-style state
dict).  Without it the script feeds the dynamics the analytic flow field of the synthetic ground truth (`flows_override`), so
that the masks — and with them the feature table — are those of the plate it wrote; the network still runs (random weights)
and its output is discarded, which is what the benchmark does too.

    python examples/cell_painting_tiff.py [--wells 4] [--fields 2] [--size 512] [--data DIR] [--weights PATH] [--out DIR]
"""

from __future__ import annotations

import argparse
import sys
import time
from copy import deepcopy
from pathlib import Path
from tempfile import mkdtemp

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

REGEX = r".*__([A-Z][0-9]{2})__([0-9]+)__([A-Za-z]+)\.tif"
CAPTURE_ORDER = "WFC"  # well, field, channel
# (channel files of a position are taken in sorted order: the index of a channel is its rank among the names)
CHANNEL_NAMES = sorted(["DNA", "ER", "RNA", "AGP", "Mito"])
CHANNELS = {name: i for i, name in enumerate(CHANNEL_NAMES)}


def write_synthetic_plate(root: Path, wells: int, fields: int, size: int) -> dict:
    """TIFFs of a synthetic plate; returns {position key: ground-truth label image of the nuclei}."""
    from aliby_amd import synth

    truth = {}
    for w in range(wells):
        well = f"{'ABCDEFGH'[w // 12]}{w % 12 + 1:02d}"
        for f in range(1, fields + 1):
            fov = synth.make_fov(2, 1000 * w + f, shape=(size, size), n_channels=len(CHANNEL_NAMES), n_target=max(8, (size // 64) ** 2))
            for name, c in CHANNELS.items():
                synth.write_tiff(root / f"plate__{well}__{f}__{name}.tif", fov["pixels"][c, 0])
            truth[f"{well}__{f}"] = dict(nuclei=fov["nuclei"], key_plane=fov["pixels"][CHANNELS["DNA"], 0])
    return truth


def analytic_flows_for(truth: dict):
    """flows_override for a plate this script wrote: every plane the segmenter sees is looked up by its bytes."""
    import torch

    from aliby_amd import synth

    table = {t["key_plane"].tobytes(): synth.analytic_flows(t["nuclei"]) for t in truth.values()}

    def override(x):  # x: uint16 [N, Y, X] on the device
        host = x.cpu().numpy()
        flows = [table[host[i].tobytes()] for i in range(host.shape[0])]
        return (torch.from_numpy(np.stack([f[0] for f in flows])).cuda(), torch.from_numpy(np.stack([f[1] for f in flows])).cuda())

    return override


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", default="", help="directory of TIFFs named <plate>__<well>__<field>__<channel>.tif (default: a synthetic plate)")
    ap.add_argument("--wells", type=int, default=4)
    ap.add_argument("--fields", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--weights", default="", help="CPnet checkpoint for the segmenter (default: analytic flows of the synthetic truth)")
    ap.add_argument("--out", default="")
    ap.add_argument("--batch-size", type=int, default=16)
    args = ap.parse_args()

    import pyarrow.parquet

    from aliby_amd.io.dataset import DatasetDir
    from aliby_amd.parallel import run_positions
    from aliby_amd.pipe_builder import build_pipeline_steps

    truth = None
    if args.data:
        data = Path(args.data)
    else:
        data = Path(mkdtemp(prefix="aliby_plate_"))
        truth = write_synthetic_plate(data, args.wells, args.fields, args.size)
        print(f"synthetic plate: {len(truth)} positions x {len(CHANNEL_NAMES)} channels of {args.size} x {args.size} under {data}")
    if not args.weights and truth is None:
        sys.exit("--data without --weights: there is nothing to segment with (the analytic flows exist for the synthetic plate only)")

    # 1. positions = groups of files (the reference's DatasetDir.get_position_ids: [{"key": "A01__1", "path": [...]}, ...])
    positions = DatasetDir(data, regex=REGEX, capture_order=CAPTURE_ORDER).get_position_ids()

    # 2. one base pipeline: nuclei on the DNA channel, intensity + sizeshape on every channel, colocalisation of every channel pair
    base = build_pipeline_steps(
        channels_to_segment={"nuclei": CHANNELS["DNA"]},
        channels_to_extract=list(CHANNELS.values()),
        features_to_extract=("intensity", "sizeshape"),
        cp_measure_feature_kwargs={"intensity": {"edge_measurements": False}},
    )
    print("pipeline steps:", list(base["steps"]))
    # net_dtype="bfloat16": the hand-written MFMA network (the default, float32, is the same module through PyTorch's own kernels)
    setup = dict(net_dtype="bfloat16", **(dict(pretrained_model=args.weights) if args.weights
                                          else dict(flows_override=analytic_flows_for(truth), run_network_with_override=True)))

    # 3. stamp it per position
    pipelines, names = [], []
    for pos in positions:
        p = deepcopy(base)
        p["io"] = {"input_path": {"key": pos["key"], "path": pos["path"]}, "capture_order": CAPTURE_ORDER}
        p["steps"]["tile"]["image_kwargs"] = {"source": {"key": pos["key"], "path": pos["path"]}, "regex": REGEX, "capture_order": CAPTURE_ORDER}
        p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = setup
        pipelines.append(p)
        names.append(pos["key"])

    pipelines_again = [deepcopy(p) for p in pipelines]  # (a pipeline dict is consumed by its run, as in the reference)

    # 4. all positions at once: batches of `batch_size` share every device step
    out = Path(args.out) if args.out else Path(mkdtemp(prefix="aliby_cellpainting_out_"))
    import torch

    torch.cuda.init()  # (a fresh machine pages PyTorch and the HIP runtime in here: tens of seconds that are not the pipeline's)
    t0 = time.perf_counter()
    results = run_positions(pipelines, names, out, overwrite=True, batch_size=args.batch_size)
    dt = time.perf_counter() - t0
    files = sorted((out / "profiles").glob("*.parquet"))
    rows = sum(r[0].num_rows for r in results if r[0] is not None)
    print(f"{len(files)} parquet files, {rows} objects, {dt:.2f} s (first call: code objects load, the segmenter is built) -> {out}")
    again = [deepcopy(p) for p in pipelines_again]
    t0 = time.perf_counter()
    run_positions(again, names, out / "again", overwrite=True, batch_size=args.batch_size)
    dt = time.perf_counter() - t0
    print(f"the same plate again: {dt:.2f} s = {len(names) / dt:.1f} positions/s (TIFF decode, upload, segmentation, features, files)")
    table = pyarrow.parquet.read_table(files[0])
    print(f"{files[0].name}: {table.num_rows} rows x {len(table.column_names)} columns, e.g. {table.column_names[4:7]}")
    if truth is not None:
        for name, res in zip(names, results):
            assert res[0].num_rows == int(truth[name]["nuclei"].max()), (name, res[0].num_rows)
        print("every synthetic nucleus is one row of its position's table")


if __name__ == "__main__":
    main()
