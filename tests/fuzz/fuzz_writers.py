"""Randomised run of the native file writers (host code: no GPU needed): random profile-shaped tables — float64 with NaN / inf /
denormals, int64, uint16, strings with non-ASCII text, 0-400 rows, 1-300 columns, written as row windows — read back with pyarrow;
random label arrays written as .npz and read back with numpy.
usage: python tests/fuzz/fuzz_writers.py [first_seed=0] [n=200]"""
import sys
import tempfile
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, ".")
from aliby_amd.io import write as w  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
tmp = Path(tempfile.mkdtemp(prefix="aliby_fuzz_writers_"))
for seed in range(first, first + n):
    rng = np.random.default_rng(29000 + seed)
    rows, ncols = int(rng.integers(0, 400)), int(rng.integers(1, 300))
    cols = {}
    for j in range(ncols):
        kind = rng.choice(["f", "f", "f", "i", "u", "s"])
        name = f"{j}/max/é{j}" if j % 7 == 0 else f"c{j}/feature_{rng.integers(0, 10**6)}"
        if kind == "f":
            a = rng.standard_normal(rows) * 10.0 ** rng.integers(-300, 300)
            if rows:
                a[rng.random(rows) < 0.05] = np.nan
                a[rng.random(rows) < 0.01] = np.inf
            cols[name] = pa.array(a)
        elif kind == "i":
            cols[name] = pa.array(rng.integers(-2**62, 2**62, rows).astype(np.int64))
        elif kind == "u":
            cols[name] = pa.array(rng.integers(0, 65536, rows).astype(np.uint16))
        else:
            cols[name] = pa.array([["nuclei", "cell", "", "ядро", "x" * int(rng.integers(0, 40))][int(k)] for k in rng.integers(0, 5, rows)], pa.string())
    table = pa.table(cols)
    path = tmp / f"t{seed}.parquet"
    w.write_profiles(table, path)
    back = pq.read_table(path)
    assert back.schema.names == table.schema.names and back.num_rows == rows, (seed, "shape")
    for name in table.schema.names:
        a, b = table[name].to_numpy(zero_copy_only=False), back[name].to_numpy(zero_copy_only=False)
        assert (np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)), (seed, name)
    # label arrays through the .npz writer
    shape = tuple(int(v) for v in rng.integers(1, 90, int(rng.integers(2, 4))))
    lab = (rng.integers(0, 50, shape) * (rng.random(shape) < 0.3)).astype(rng.choice([np.uint16, np.uint8, np.int32]))
    w.write_ndarray(lab, tmp, "steps", seed)
    with np.load(tmp / "steps" / f"{seed:04d}.npz") as z:
        assert list(z.keys()) == ["arr_0"] and z["arr_0"].dtype == lab.dtype and np.array_equal(z["arr_0"], lab), (seed, "npz")
    path.unlink()
    (tmp / "steps" / f"{seed:04d}.npz").unlink()
    if seed % 50 == 0:
        print(f"seed {seed}: {rows} x {ncols}, labels {shape}: ok", flush=True)
print(f"{n} seeds ok")
