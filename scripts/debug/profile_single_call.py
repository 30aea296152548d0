"""cProfile of one run_pipeline_and_post call on a config-2-shaped position (after two warm calls)."""
import cProfile
import pstats
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from aliby_amd import synth  # noqa: E402
from aliby_amd.pipe import run_pipeline_and_post  # noqa: E402
from aliby_amd.pipe_builder import build_pipeline_steps  # noqa: E402

f = synth.make_fov(2, 300)
flows = synth.analytic_flows(f["nuclei"])
pinned = torch.from_numpy(f["pixels"][None]).pin_memory()
override = lambda x: (torch.from_numpy(flows[0][None]).cuda(), torch.from_numpy(flows[1][None]).cuda())  # noqa: E731


def pipe():
    p = build_pipeline_steps(channels_to_segment={"nuclei": 0}, channels_to_extract=[0, 1, 2, 3, 4])
    p["steps"]["tile"]["image_kwargs"] = {"source": pinned.numpy()}
    p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override, run_network_with_override=True)
    return p


tmp = Path(tempfile.mkdtemp())
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for k in range(2):
        run_pipeline_and_post(pipeline=pipe(), pipeline_name=f"w{k}", output_path=tmp, overwrite=True)
    pr = cProfile.Profile()
    pr.enable()
    for k in range(5):
        run_pipeline_and_post(pipeline=pipe(), pipeline_name=f"p{k}", output_path=tmp, overwrite=True)
    pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
