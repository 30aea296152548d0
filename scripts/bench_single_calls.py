"""What a caller gets who keeps the reference's loop — one run_pipeline_and_post per position — instead of run_positions:
config-2-shaped positions (1024^2, 5 channels, ~256 nuclei, the bench's feature tree), analytic flows, files on disk.
usage: python scripts/bench_single_calls.py [N=16]"""
import shutil
import sys
import tempfile
import time
import warnings
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aliby_amd import synth  # noqa: E402
from aliby_amd.parallel import run_positions  # noqa: E402
from aliby_amd.pipe import run_pipeline_and_post  # noqa: E402
from aliby_amd.pipe_builder import build_pipeline_steps  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fovs = [synth.make_fov(2, 300 + i) for i in range(N)]
table = {f["pixels"][0].max(axis=0).tobytes(): synth.analytic_flows(f["nuclei"]) for f in fovs}
pinned = [torch.from_numpy(f["pixels"][None]).pin_memory() for f in fovs]


def override(x):
    host = x.cpu().numpy()
    fl = [table[host[i].tobytes()] for i in range(host.shape[0])]
    return torch.from_numpy(np.stack([a for a, _ in fl])).cuda(), torch.from_numpy(np.stack([b for _, b in fl])).cuda()


def pipelines():
    out = []
    for i in range(N):
        p = build_pipeline_steps(channels_to_segment={"nuclei": 0}, channels_to_extract=[0, 1, 2, 3, 4])
        p["steps"]["tile"]["image_kwargs"] = {"source": pinned[i].numpy()}
        p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override, run_network_with_override=True)
        out.append(p)
    return out


tmp = Path(tempfile.mkdtemp())
names = [f"P{i:03d}" for i in range(N)]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(2):  # (the first round loads code objects)
        t0 = time.perf_counter()
        for p, nm in zip(pipelines(), names):
            run_pipeline_and_post(pipeline=p, pipeline_name=nm, output_path=tmp / f"single{rep}", overwrite=True)
        torch.cuda.synchronize()
        dt_single = time.perf_counter() - t0
    for rep in range(2):
        t0 = time.perf_counter()
        run_positions(pipelines(), names, tmp / f"batched{rep}", batch_size=N)
        torch.cuda.synchronize()
        dt_batched = time.perf_counter() - t0
print(f"{N} positions: one call each {dt_single:.2f} s = {N / dt_single:.1f} positions/s; run_positions {dt_batched:.2f} s = {N / dt_batched:.1f} positions/s")
shutil.rmtree(tmp, ignore_errors=True)
