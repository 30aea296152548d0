"""
The native encoders of the step API's two files (csrc/host_writers.hip through aliby_amd/io/write.py): what they write is read
back by pyarrow / numpy exactly like the files the reference's calls produce (pipe_core.py:412-413, io/write.py:25-51).
Host code only: these run without a GPU.
"""
import os
import zipfile

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from aliby_amd.io import write as W


def _profiles(n=37, n_feat=60, seed=0):
    rng = np.random.default_rng(seed)
    cols = {"metadata_tile": pa.array(np.zeros(n, np.int64)), "metadata_label": pa.array(np.arange(1, n + 1))}
    for j in range(n_feat // 2):
        cols[f"{j % 5}/max/intensity/Intensity_Feature_{j:03d}"] = pa.array(rng.normal(1000, 300, n))
    cols["metadata_object"] = pa.array(["nuclei"] * (n // 2) + ["cells"] * (n - n // 2), pa.string())
    cols["metadata_tp"] = pa.array(np.arange(n).astype(np.uint16))
    for j in range(n_feat // 2, n_feat):
        v = rng.random(n)
        v[rng.random(n) < 0.1] = np.nan  # NaN is a value (e.g. NormalizedMoment_0_0), not a null
        cols[f"(0, {j % 5})/None/max/pearson/Correlation_{j:03d}"] = pa.array(v)
    return pa.table(cols)


def _equal_tables(a, b):
    assert a.schema.names == b.schema.names and a.schema.types == b.schema.types and a.num_rows == b.num_rows
    for name in a.column_names:
        x, y = a[name].to_numpy(zero_copy_only=False), b[name].to_numpy(zero_copy_only=False)
        assert x.dtype == y.dtype, name
        assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), name


def test_native_parquet_reads_back_like_the_pyarrow_file(tmp_path):
    t = _profiles()
    W.write_profiles(t, tmp_path / "native.parquet")
    pq.write_table(t, tmp_path / "ref.parquet", compression="zstd")  # the reference's call
    got, want = pq.read_table(tmp_path / "native.parquet"), pq.read_table(tmp_path / "ref.parquet")
    _equal_tables(got, want)
    md = pq.read_metadata(tmp_path / "native.parquet")
    assert md.num_row_groups == 1 and md.num_rows == t.num_rows and md.num_columns == t.num_columns
    assert md.created_by.startswith("aliby_amd")  # i.e. the native encoder wrote it, not the pyarrow route
    assert all(md.row_group(0).column(j).compression == "ZSTD" for j in range(md.num_columns))
    import pandas

    assert pandas.read_parquet(tmp_path / "native.parquet").shape == (t.num_rows, t.num_columns)


def test_native_parquet_row_windows_and_chunked_tables_give_the_same_bytes(tmp_path):
    t = _profiles(n=200)
    layout = W.table_layout(t)
    assert layout is not None
    W.write_parquet_native(tmp_path / "a.parquet", [(layout, 10, 50), (layout, 120, 7)])  # the batched runner's route
    chunked = pa.concat_tables([t.slice(10, 50), t.slice(120, 7)])                          # the single call's route
    W.write_profiles(chunked, tmp_path / "b.parquet")
    assert (tmp_path / "a.parquet").read_bytes() == (tmp_path / "b.parquet").read_bytes()
    _equal_tables(pq.read_table(tmp_path / "a.parquet"), chunked.combine_chunks())
    # unevenly chunked columns still end up in the same file
    odd = pa.Table.from_arrays([pa.chunked_array([c.chunk(0), c.chunk(1)]) if i % 2 else pa.chunked_array([pa.concat_arrays(c.chunks)])
                                for i, c in enumerate(chunked.columns)], schema=chunked.schema)
    W.write_profiles(odd, tmp_path / "c.parquet")
    assert (tmp_path / "c.parquet").read_bytes() == (tmp_path / "a.parquet").read_bytes()


def test_tables_the_native_encoder_does_not_cover_go_through_pyarrow(tmp_path):
    t = _profiles(n=9, n_feat=4)
    with_null = t.append_column("sparse", pa.array([1.0, None] + [2.0] * 7))
    W.write_profiles(with_null, tmp_path / "n.parquet")
    back = pq.read_table(tmp_path / "n.parquet")
    assert back["sparse"].null_count == 1 and not pq.read_metadata(tmp_path / "n.parquet").created_by.startswith("aliby_amd")
    W.write_profiles(t.slice(0, 0), tmp_path / "e.parquet")
    assert pq.read_table(tmp_path / "e.parquet").num_rows == 0
    nested = t.append_column("v", pa.array([[1, 2]] * 9))
    W.write_profiles(nested, tmp_path / "l.parquet")
    assert pq.read_table(tmp_path / "l.parquet")["v"].to_pylist()[0] == [1, 2]


def _labels(shape=(300, 400), n=40, seed=3):
    rng = np.random.default_rng(seed)
    lab = np.zeros(shape, np.uint16)
    for k in range(1, n + 1):
        y, x = rng.integers(0, shape[0] - 20), rng.integers(0, shape[1] - 20)
        yy, xx = np.ogrid[:shape[0], :shape[1]]
        lab[(yy - y - 10) ** 2 + (xx - x - 10) ** 2 < rng.integers(16, 90)] = k
    return lab


def test_native_npz_is_a_savez_compressed_file(tmp_path):
    lab = _labels()
    W.write_ndarray(lab, tmp_path, "segment_nuclei", 3)
    f = tmp_path / "segment_nuclei" / "0003.npz"
    with np.load(f) as z:
        assert list(z.keys()) == ["arr_0"] and z["arr_0"].dtype == np.uint16 and np.array_equal(z["arr_0"], lab)
    with zipfile.ZipFile(f) as zf:
        assert zf.testzip() is None and zf.namelist() == ["arr_0.npy"] and zf.infolist()[0].compress_type == zipfile.ZIP_DEFLATED
        assert zf.infolist()[0].compress_size * 8 < lab.nbytes  # the label encoder took it (runs + row repeats)
    np.savez_compressed(tmp_path / "ref.npz", lab)
    with np.load(tmp_path / "ref.npz") as z:
        assert np.array_equal(z["arr_0"], lab)
    # the reference's dict form: one member per tile, metadata beside it
    W.write_ndarray({"masks": [lab[:100, :50], lab[5:9].astype(np.int32)], "metadata": {"a": 1}}, tmp_path, "seg", 4)
    with np.load(tmp_path / "seg" / "0004.npz") as z:
        assert list(z.keys()) == ["tile_0", "tile_1"] and z["tile_1"].dtype == np.int32 and np.array_equal(z["tile_0"], lab[:100, :50])
    assert (tmp_path / "seg" / "0004_meta.json").read_text() == '{"a": 1}'


@pytest.mark.parametrize("case", range(12))
def test_native_npz_array_kinds(tmp_path, case):
    rng = np.random.default_rng(case)
    lab = _labels()
    a = [np.zeros((0, 5), np.float32), np.float64(3.5), np.arange(7) > 3, np.arange(24).reshape(2, 3, 4).astype(np.int8), lab[::2, ::3],
         lab[None].repeat(3, 0), rng.integers(0, 65535, (300, 400), dtype=np.uint16), rng.normal(size=(64, 64)), lab > 0, lab.astype(np.int64),
         lab[:, :1], np.zeros((3, 40000), np.uint16)][case]  # (the last: rows longer than deflate's window)
    W.write_ndarray(a, tmp_path, "x", 0)
    with np.load(tmp_path / "x" / "0000.npz") as z:
        b = z["arr_0"]
        assert b.dtype == np.asarray(a).dtype and b.shape == np.asarray(a).shape and np.array_equal(a, b)


def test_object_results_keep_numpys_pickled_form(tmp_path):
    W.write_ndarray({"drift": [1, 2], "pixels": np.ones((4, 4), np.uint16)}, tmp_path, "tile", 0)  # the tile step's dict (write.py:50)
    got = np.load(tmp_path / "tile" / "0000.npz", allow_pickle=True)["arr_0"].item()  # (this test's own file)
    assert sorted(got) == ["drift", "pixels"]


def test_native_writers_can_be_switched_off(tmp_path, monkeypatch):
    monkeypatch.setenv("ALIBY_NATIVE_WRITERS", "0")
    t = _profiles(n=5, n_feat=4)
    W.write_profiles(t, tmp_path / "p.parquet")
    assert pq.read_metadata(tmp_path / "p.parquet").created_by.startswith("parquet-cpp")
    _equal_tables(pq.read_table(tmp_path / "p.parquet"), t)


def test_host_copy_on_several_threads_is_a_memcpy():
    """aliby_host_copy: sizes around the 8 MB threshold and the page-rounded piece boundaries, 1..16 threads, odd byte counts."""
    import numpy as np

    from aliby_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)
    for nbytes in (0, 1, 4095, (8 << 20) - 1, (8 << 20) + 1, (9 << 20) + 12345):
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)
        for threads in (1, 3, 4, 16, 99):
            dst = np.zeros(nbytes + 8, np.uint8)
            _lib.check(lib.aliby_host_copy(dst.ctypes.data, src.ctypes.data, nbytes, threads))
            assert np.array_equal(dst[:nbytes], src) and not dst[nbytes:].any()


def test_native_writers_fail_loudly_on_a_path_they_cannot_open(tmp_path):
    """Both encoders assemble the file in memory and write it in one piece (csrc/host_writers.hip File): a directory that does
    not exist is an error with the path in it, nothing is left behind, and the next file from the same thread is intact."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq

    from aliby_amd.io import write as W

    table = pa.table({"a": pa.array(np.arange(5.0)), "b": pa.array(np.arange(5, dtype=np.int64))})
    lay = W.table_layout(table)
    missing = tmp_path / "no" / "such" / "dir"
    with pytest.raises(ValueError, match="cannot open"):
        W.write_parquet_native(missing / "x.parquet", [(lay, 0, 5)])
    with pytest.raises(ValueError, match="cannot open"):
        W.write_npz_native(missing / "x.npz", {"arr_0": np.arange(4)})
    assert not missing.exists()
    W.write_parquet_native(tmp_path / "ok.parquet", [(lay, 1, 3)])
    assert pq.read_table(tmp_path / "ok.parquet")["a"].to_pylist() == [1.0, 2.0, 3.0]
    assert W.write_npz_native(tmp_path / "ok.npz", {"arr_0": np.arange(4)})
    assert np.load(tmp_path / "ok.npz")["arr_0"].tolist() == [0, 1, 2, 3]


def test_randomised_tables_and_label_arrays_read_back(monkeypatch, capsys):
    """tests/fuzz/fuzz_writers.py (3300 seeds by hand): random profile-shaped tables — NaN / inf / extreme floats, int64, uint16,
    non-ASCII strings and column names, empty tables — and random label arrays through the native encoders, read back with
    pyarrow / numpy."""
    import runpy
    import sys
    from pathlib import Path

    script = Path(__file__).resolve().parent / "fuzz" / "fuzz_writers.py"
    monkeypatch.setattr(sys, "argv", [str(script), "0", "40"])
    monkeypatch.chdir(script.parents[2])
    runpy.run_path(str(script), run_name="__main__")
    assert "40 seeds ok" in capsys.readouterr().out

