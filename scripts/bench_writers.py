"""Thread scaling of the two native file encoders on this machine (no GPU): 64 profiles files (1047 columns x 256 rows, windows of
one batch table) and 64 label images (1024^2 uint16) per round, at 1 / 4 / 12 threads, on the temp directory and on /dev/shm."""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pyarrow as pa

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aliby_amd import synth  # noqa: E402
from aliby_amd.io import write as W  # noqa: E402

ncols, nrows = 1047, 16384
mat = np.random.rand(ncols, nrows)
table = pa.table({f"c{j:04d}": pa.array(mat[j]) for j in range(ncols)})
lay = W.table_layout(table)
labels = [synth.make_fov(2, k, shape=(1024, 1024), n_channels=1, n_z=1, n_target=256)["nuclei"].astype(np.uint16) for k in range(4)]

for root in [tempfile.gettempdir(), "/dev/shm"]:
    out = tempfile.mkdtemp(dir=root)

    def pq(k):
        W.write_parquet_native(f"{out}/{k}.parquet", [(lay, (k % 64) * 256, 256)])

    def npz(k):
        W.write_npz_native(f"{out}/{k}.npz", {"arr_0": labels[k % 4]})

    for name, fn in (("parquet", pq), ("npz", npz)):
        fn(0)
        res = {}
        for nt in (1, 4, 12):
            with ThreadPoolExecutor(nt) as ex:
                t0 = time.perf_counter()
                list(ex.map(fn, range(64)))
                dt = time.perf_counter() - t0
            res[nt] = f"{1e3 * dt:.1f} ms / 64 files = {1e3 * dt * nt / 64:.2f} thread-ms per file"
        print(root, name, res)
    for f in os.listdir(out):
        os.unlink(f"{out}/{f}")
    os.rmdir(out)
