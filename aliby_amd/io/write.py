"""
Per-timepoint step outputs on disk, same layout as the reference
(src/aliby/io/write.py:8-74): `<steps_dir>/<step>/<tp:04d>.npz` (numpy.savez_compressed's format: `arr_0`, or `tile_i` keys
for dict results carrying "masks"), profiles as parquet with zstd (pipe_core.py:412-413).

Both files are encoded by libaliby_hip.so (csrc/host_writers.hip: `aliby_parquet_write`, `aliby_npz_write`) — the same
formats, read back by pyarrow / numpy like the reference's — because the reference's two library calls are what a position
costs once the kernels are fast: ~15 CPU-ms for a thousand-column parquet through pyarrow, ~5.5 ms for a label image through
numpy + zlib, against ~2 ms of device time per FOV.  The native calls run with the interpreter lock released, so the
position-batched runner's writer THREADS encode files side by side (no writer processes, no Arrow IPC hop).  Tables or arrays
the native encoders do not cover (null values, nested / object dtypes) go through pyarrow / numpy as before.
"""

from __future__ import annotations

import ctypes as C
import json
import os
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

# zstd level of the profiles files.  pyarrow's default for compression="zstd" is 1; a page here is one column of one position
# (~2 KB), where level 1 spends 3/4 of its time building a Huffman table per page for 4 % of size: -1 (zstd's "fast" mode,
# raw literals) encodes a 1047-column file in 0.3x the time at 1.05x the size
ZSTD_LEVEL = int(os.environ.get("ALIBY_PARQUET_ZSTD_LEVEL", "-1"))


# ------------------------------------------------------------------------------------------------ parquet
class TableLayout:
    """Where the values of a (single-chunk, null-free) Arrow table live: per column the address of row 0 and the item width,
    so that a window of rows is `addr + lo * width` for every column at once.  Built once per table (per device batch in
    aliby_amd/runner.py) and shared by every file written from it; keeps the table alive."""

    def __init__(self, table: pa.Table):
        from aliby_amd import _lib

        self.table = table
        n = table.num_columns
        self.cols = (_lib.aliby_pq_column * n)()
        self._names = []  # the C strings the structs point at
        self.addr = np.zeros(n, np.uint64)
        self.aux = np.zeros(n, np.uint64)
        self.width = np.zeros(n, np.uint64)
        for j, field in enumerate(table.schema):
            t = field.type
            col = table.column(j)
            if col.num_chunks != 1:
                raise _Unsupported("chunked column")
            arr = col.chunk(0)
            if arr.null_count:
                raise _Unsupported("null values")
            bufs = arr.buffers()
            if pa.types.is_float64(t):
                kind, w = _lib.PQ_F64, 8
            elif pa.types.is_int64(t):
                kind, w = _lib.PQ_I64, 8
            elif pa.types.is_uint16(t):
                kind, w = _lib.PQ_U16, 2
            elif pa.types.is_string(t):
                kind, w = _lib.PQ_STR, 4
                self.aux[j] = bufs[2].address if bufs[2] is not None else 0
            else:
                raise _Unsupported(f"type {t}")
            name = field.name.encode("utf-8")
            self._names.append(name)
            self.cols[j].name = name
            self.cols[j].type = kind
            self.addr[j] = (bufs[1].address if bufs[1] is not None else 0) + arr.offset * w
            self.width[j] = w


class _Unsupported(Exception):
    pass


def table_layout(table: pa.Table):
    """TableLayout of `table`, or None when the native encoder does not cover it."""
    try:
        return TableLayout(table)
    except _Unsupported:
        return None


def write_parquet_native(path, segments) -> None:
    """One parquet file from row windows of tables with the same columns: segments = [(TableLayout, first row, rows), ...]."""
    from aliby_amd import _lib

    lib = _lib.load()
    first = segments[0][0]
    n_cols, n_segs = len(first.width), len(segments)
    rows = np.asarray([s[2] for s in segments], np.int64)
    values = np.empty((n_cols, n_segs), np.uint64)
    aux = np.empty((n_cols, n_segs), np.uint64)
    for k, (layout, lo, _) in enumerate(segments):
        values[:, k] = layout.addr + np.uint64(lo) * layout.width
        aux[:, k] = layout.aux
    _lib.check(lib.aliby_parquet_write(os.fsencode(str(path)), C.cast(first.cols, C.c_void_p), n_cols, rows.ctypes.data, n_segs,
                                       values.ctypes.data, aux.ctypes.data, ZSTD_LEVEL))


def _same_columns(a: pa.Schema, b: pa.Schema) -> bool:
    return a.names == b.names and a.types == b.types


def write_profiles(table, path) -> None:
    """The profiles table as the reference writes it (pipe_core.py:412-413: parquet, zstd): one row group, PLAIN pages, no
    dictionaries (float features never profit from them), natively encoded; a table with nulls or other column types than
    float64 / int64 / uint16 / string goes through pyarrow with the same settings."""
    if table.num_rows > 0 and os.environ.get("ALIBY_NATIVE_WRITERS", "1") != "0":
        layout = None
        # a position's table is the concatenation of one slice per object set: chunk k of every column is slice k
        n_chunks = {c.num_chunks for c in table.columns}
        if n_chunks == {1}:
            layout = table_layout(table)
            segments = [(layout, 0, table.num_rows)] if layout is not None else None
        else:
            segments = None
            if len(n_chunks) == 1 and all([len(ch) for ch in c.chunks] == [len(ch) for ch in table.column(0).chunks] for c in table.columns):
                parts = [pa.Table.from_arrays([c.chunk(k) for c in table.columns], schema=table.schema) for k in range(n_chunks.pop())]
                layouts = [table_layout(p) for p in parts]
                if all(lay is not None for lay in layouts):
                    segments = [(lay, 0, p.num_rows) for lay, p in zip(layouts, parts)]
            if segments is None:
                layout = table_layout(table.combine_chunks())
                segments = [(layout, 0, table.num_rows)] if layout is not None else None
        if segments is not None:
            write_parquet_native(path, segments)
            return
    pq.write_table(table, path, compression="zstd", use_dictionary=False)


def write_parquet(result, out_dir, subpath: str, filename: str) -> None:
    this_outdir = Path(out_dir) / subpath
    this_outdir.mkdir(exist_ok=True, parents=True)
    write_profiles(result, this_outdir / f"{filename}.parquet")


# ------------------------------------------------------------------------------------------------ npz
NPZ_LEVEL = int(os.environ.get("ALIBY_NPZ_LEVEL", "6"))  # numpy.savez_compressed deflates at zlib's default level, 6


def write_npz_native(path, members: dict) -> bool:
    """numpy.savez_compressed(path, **members) for plain numeric arrays; False (nothing written) when a member is not one."""
    from aliby_amd import _lib

    arrays, keep = [], []
    for name, a in members.items():
        a = np.asarray(a)
        if a.dtype.kind not in "biufc" or a.dtype.hasobject or not a.dtype.isnative:
            return False
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)
        arrays.append((name.encode(), a.dtype.str.encode(), (C.c_int64 * max(a.ndim, 1))(*a.shape), a))
        keep.append(a)
    lib = _lib.load()
    ms = (_lib.aliby_npy_member * len(arrays))()
    for m, (name, descr, shape, a) in zip(ms, arrays):
        m.name, m.descr, m.shape, m.data, m.ndim, m.itemsize = name, descr, shape, a.ctypes.data, a.ndim, a.dtype.itemsize
    _lib.check(lib.aliby_npz_write(os.fsencode(str(path)), C.cast(ms, C.c_void_p), len(arrays), NPZ_LEVEL))
    return True


def write_ndarray(result, steps_dir, subpath, tp: int) -> None:
    this_step = Path(steps_dir) / subpath
    this_step.mkdir(exist_ok=True, parents=True)
    out_file = this_step / f"{tp:04d}.npz"
    native = os.environ.get("ALIBY_NATIVE_WRITERS", "1") != "0"
    if isinstance(result, dict) and "masks" in result:
        members = {f"tile_{i}": np.array(m) for i, m in enumerate(result["masks"])}
        if not (native and members and write_npz_native(out_file, members)):
            np.savez_compressed(out_file, **members)
        if "metadata" in result:
            (this_step / f"{tp:04d}_meta.json").write_text(json.dumps(result["metadata"]))
    else:
        arr = result if isinstance(result, np.ndarray) else None
        if arr is None and not isinstance(result, dict):
            try:
                arr = np.asarray(result)
            except ValueError:  # ragged lists: numpy pickles them as an object array, like the reference
                arr = None
        if not (native and arr is not None and write_npz_native(out_file, {"arr_0": arr})):
            np.savez_compressed(out_file, np.asarray(result))


def dispatch_write_fn(step_name: str):
    if step_name.startswith("segment") or step_name.startswith("tile"):
        return write_ndarray
    if step_name.startswith("nahual_trackastra"):
        return write_parquet
    raise Exception(f"Writing {step_name} is not supported yet")
