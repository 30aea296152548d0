"""Throughput of the IoU stitcher (aliby_track_stitch) at config-2 scale: 32 tiles of 1024^2, ~256 nuclei each, the
newer frame = the older one shifted by (2, 3) px.  Reports frame pairs/s and the fraction of the HBM roofline for the
algorithmic traffic (both label planes read once)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from aliby_amd import synth  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine  # noqa: E402

eng = FeatureEngine()
F = 32
base = [synth.make_fov(2, i, shape=(1024, 1024))["nuclei"] for i in range(4)]
stack = np.stack([base[i % 4] for i in range(F)])
prev = torch.from_numpy(stack).cuda()
cur = torch.from_numpy(np.ascontiguousarray(np.roll(stack, (2, 3), axis=(1, 2)))).cuda()
tp, tc = eng.object_table(prev), eng.object_table(cur)
for _ in range(3):
    tracked, mx = eng.track_stitch(prev, cur, tp, tc, None, None, 0.25)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
t0 = time.perf_counter()
e0.record()
for _ in range(reps):
    tracked, mx = eng.track_stitch(prev, cur, tp, tc, None, None, 0.25)
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
ms = e0.elapsed_time(e1) / reps
alg = 2 * prev.numel() * 2
kept = int((tracked.cpu().numpy() == np.concatenate([np.arange(1, n + 1) for n in np.diff(tc.offsets)])).sum())
print(f"track_stitch: {F} tiles, {tc.n_obj} objects: {ms:.3f} ms per call (wall {wall*1e3:.3f} ms incl. the scalar readback) = "
      f"{F/ms*1e3:.0f} frame pairs/s; algorithmic {alg/1e6:.0f} MB -> {alg/ms/1e6:.0f} GB/s = {alg/ms/1e6/8000:.3f} of 8 TB/s; "
      f"{kept}/{tc.n_obj} labels carried over")
