"""
Yeast time-lapse on one MI355X: a zarr store of TCZYX positions -> trap tiles (ALCATRAS detection + drift) -> per-tile segmentation
-> IoU tracking -> size / shape / intensity per cell and time point, through the reference's step dicts and
`aliby_amd.parallel.run_positions`.

The reference's example for this workload (examples/03_yeast_timelapse_baby.py there) opens a Zenodo zarr with `DatasetZarr` and
segments with BABY behind a Nahual server — a remote service that is out of scope here.  This script keeps the data
side of that example (zarr store, one TCZYX array per position, `tile_size` = 117, `ref_channel` / `ref_z`) and puts the Cellpose
segmenter of this build in BABY's place:

  1. writes a synthetic store — positions x [T, 1, Z, 512, 512] uint16, a trap grid with budding cells that drifts over time
     (`aliby_amd.synth.make_timelapse`) — unless `--data` points at a zarr store of your own;
  2. lists the positions with `DatasetZarr`, as the reference does;
  3. one pipeline dict per position: tile (traps, drift) -> segment_cells (per tile) -> track (stitch) -> extract_cells;
  4. hands them all to `run_positions`: the positions advance through the time points in lockstep, every time point one device
     batch.

Segmentation needs Cellpose weights; none are obtainable offline.  `--weights PATH` loads a CPnet checkpoint.  Without it a
classical flow field stands in for the network's output (`flows_override`, the hook for "flows from any other model"): the
gradient of the smoothed intensity points to the middle of a bright cell, which is all the flow-following dynamics need.  The
masks of that stand-in are approximate; what the script demonstrates is the data flow.

    python examples/yeast_timelapse_zarr.py [--positions 2] [--tps 6] [--data STORE.zarr] [--weights PATH] [--out DIR]
"""

from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path
from tempfile import mkdtemp

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

TILE_SIZE = 117


def write_zarr_array(store: Path, key: str, a: np.ndarray) -> None:
    """A zarr v2 array (uncompressed, one chunk per plane) under `store/key` — what `ImageZarr` opens as (store, key)."""
    arr = store / key
    arr.mkdir(parents=True, exist_ok=True)
    (store / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
    chunks = (1, 1, 1) + a.shape[3:]
    (arr / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": list(a.shape), "chunks": list(chunks), "dtype": a.dtype.str,
                                             "order": "C", "fill_value": 0, "filters": None, "compressor": None}))
    for t in range(a.shape[0]):
        for c in range(a.shape[1]):
            for z in range(a.shape[2]):
                (arr / f"{t}.{c}.{z}.0.0").write_bytes(np.ascontiguousarray(a[t, c, z]).tobytes())


def gradient_flows(x):
    """(dP, cellprob) from intensity alone, for bright convex cells: x uint16 [N, h, w] on the device."""
    import torch
    import torch.nn.functional as F

    f = x.to(torch.int32).to(torch.float32)
    f = f - f.flatten(1).median(dim=1).values[:, None, None]
    r = torch.arange(-6, 7, device=x.device, dtype=torch.float32)
    k = torch.exp(-0.5 * (r / 2.5) ** 2)
    k = (k / k.sum())[None, None]
    s = F.conv2d(F.conv2d(F.pad(f[:, None], (6, 6, 6, 6), mode="replicate"), k[..., None]), k[:, :, None])[:, 0]
    gy = torch.zeros_like(s)
    gx = torch.zeros_like(s)
    gy[:, 1:-1] = s[:, 2:] - s[:, :-2]
    gx[:, :, 1:-1] = s[:, :, 2:] - s[:, :, :-2]
    norm = torch.sqrt(gy * gy + gx * gx) + 1e-6
    level = 0.35 * s.flatten(1).max(dim=1).values[:, None, None]
    inside = (s > level).to(torch.float32)
    dP = torch.stack([gy / norm, gx / norm], dim=1) * 5.0 * inside[:, None]
    return dP.contiguous(), ((s - level) / (level + 1.0) * 6.0).contiguous()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", default="", help="a zarr store with one TCZYX uint16 array per position (default: a synthetic one)")
    ap.add_argument("--positions", type=int, default=2)
    ap.add_argument("--tps", type=int, default=6)
    ap.add_argument("--weights", default="", help="CPnet checkpoint for the segmenter (default: the intensity-gradient stand-in)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    from aliby_amd import synth
    from aliby_amd.io.dataset import DatasetZarr
    from aliby_amd.parallel import run_positions

    if args.data:
        store = Path(args.data)
    else:
        store = Path(mkdtemp(prefix="aliby_timelapse_")) / "timelapse.zarr"
        for p in range(args.positions):
            write_zarr_array(store, f"pos{p:03d}", synth.make_timelapse(T=args.tps, seed=11 + p)["pixels"])
        print(f"synthetic store: {args.positions} positions of [T={args.tps}, C=1, Z=5, 512, 512] under {store}")

    # 1. positions of the store ([{"path": store, "key": "pos000"}, ...] — the reference's DatasetZarr.get_position_ids)
    positions = sorted(DatasetZarr(store).get_position_ids(), key=lambda p: p["key"])

    # 2. one pipeline per position: the step dicts of the reference's engine (pipe_core.py), written out in full
    setup = dict(pretrained_model=args.weights) if args.weights else dict(flows_override=gradient_flows)
    tree = {"None": {"None": ["sizeshape"]}, 0: {"max": ["intensity"]}}

    def pipeline_for(pos):
        return {
            "ntps": args.tps,
            "steps": {
                "tile": {"image_kwargs": {"source": {"path": str(pos["path"]), "key": pos["key"]}}, "tile_size": TILE_SIZE,
                         "ref_channel": 0, "ref_z": 0, "calculate_drift": True},
                "segment_cells": {"segmenter_kwargs": {"kind": "cellpose", "per_tile": True, "setup_params": dict(setup)},
                                  "channel_to_segment": 0},
                "track": {"kind": "stitch", "stitch_threshold": 0.25},
                "extract_cells": {"tree": tree},
            },
            "passed_data": {"track": [("masks", "segment_cells"), ("track_info", "track")],
                            "extract_cells": [("masks", "segment_cells"), ("pixels", "tile")]},
            "passed_methods": {"segment_cells": ("tile", "get_fczyx")},
            "save": ("segment_cells",), "save_interval": 1, "retain": {"tile": 1, "segment_cells": 2},
        }

    pipelines = [pipeline_for(p) for p in positions]
    names = [p["key"] for p in positions]
    print("pipeline steps:", list(pipelines[0]["steps"]))

    # 3. all positions in lockstep
    out = Path(args.out) if args.out else Path(mkdtemp(prefix="aliby_timelapse_out_"))
    import torch

    torch.cuda.init()
    t0 = time.perf_counter()
    results = run_positions(pipelines, names, out, overwrite=True, batch_size=len(pipelines))
    dt = time.perf_counter() - t0
    for name, (profiles, _) in zip(names, results):
        df = profiles.to_pandas()
        per_tp = df.groupby("metadata_tp")["metadata_label"].count().tolist()
        print(f"{name}: {profiles.num_rows} rows x {profiles.num_columns} columns; cells per time point {per_tp}; "
              f"{df['metadata_tile'].nunique()} trap tiles")
    print(f"{len(names)} positions x {args.tps} time points in {dt:.2f} s (first call: code objects load) -> {out}")


if __name__ == "__main__":
    main()
