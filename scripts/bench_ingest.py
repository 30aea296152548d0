"""Ingest throughput on the GPU box: one config-2 FOV (5 channels of 1024x1024 uint16, one file per channel) decoded +
uploaded by aliby_ingest_tiff_planes, uncompressed and Deflate, against the plain upload of an in-memory stack."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, ".")

from aliby_amd import synth
from aliby_amd.extraction.engine import FeatureEngine
from aliby_amd.io.image import ImageList

eng = FeatureEngine()
fov = synth.make_fov(2, 0)["pixels"]  # [5,1,1024,1024]
regex, order = ".*__([A-Z][0-9]{2})__([0-9])__([A-Za-z]+).tif", "WFC"
names = ("DNA", "ER", "RNA", "AGP", "Mito")
stream = torch.cuda.Stream()
for comp in (None, "deflate"):
    tmp = Path(tempfile.mkdtemp())
    files = []
    for c, ch in enumerate(names):
        p = tmp / f"plate__A01__1__{ch}.tif"
        synth.write_tiff(p, fov[c, 0], compression=comp, rows_per_strip=64)
        files.append(str(p))
    files.sort()
    size = sum(Path(f).stat().st_size for f in files)
    data = ImageList(source=files, regex=regex, capture_order=order).data
    out = torch.empty(data.shape[1:], dtype=torch.uint16, device="cuda")
    for threads_label in ("pool",):
        data.read_device(0, eng.ctx.handle, stream.cuda_stream, out=out)
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            data.read_device(0, eng.ctx.handle, stream.cuda_stream, out=out)
        dt = (time.perf_counter() - t0) / n
        print(f"compression={comp}: {size / 1e6:.1f} MB on disk, {dt * 1e3:.2f} ms per FOV to HBM "
              f"({out.numel() * 2 / dt / 1e9:.2f} GB/s of pixels, {1 / dt:.0f} FOV/s per ingest thread pool)")
    t0 = time.perf_counter()
    for _ in range(20):
        host = data[0]
    print(f"   host decode only: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per FOV")
src = torch.from_numpy(np.ascontiguousarray(fov)).pin_memory()
dst = torch.empty_like(src, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"pinned in-memory stack upload: {dt * 1e3:.2f} ms per FOV ({src.numel() * 2 / dt / 1e9:.1f} GB/s)")
