"""Run-to-run determinism of the whole path: the same 16 config-2 positions through run_positions several times (batches of
different sizes, so the kernels see other launch shapes and workgroup schedules): labels and every feature value bit for bit."""
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from aliby_amd import synth  # noqa: E402
from aliby_amd.parallel import run_positions  # noqa: E402
from aliby_amd.pipe_builder import build_pipeline_steps  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fovs = [synth.make_fov(2, 500 + i) for i in range(N)]
table = {f["pixels"][0].max(axis=0).tobytes(): synth.analytic_flows(f["nuclei"]) for f in fovs}


def override(x):
    host = x.cpu().numpy()
    fl = [table[host[i].tobytes()] for i in range(host.shape[0])]
    return torch.from_numpy(np.stack([a for a, _ in fl])).cuda(), torch.from_numpy(np.stack([b for _, b in fl])).cuda()


def pipelines():
    out = []
    for f in fovs:
        p = build_pipeline_steps(channels_to_segment={"nuclei": 0}, channels_to_extract=[0, 1, 2, 3, 4])
        p["steps"]["tile"]["image_kwargs"] = {"source": f["pixels"][None]}
        p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override, run_network_with_override=True)
        out.append(p)
    return out


tmp = Path(tempfile.mkdtemp())
names = [f"D{i:02d}" for i in range(N)]
ref = None
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep, bs in enumerate((16, 16, 8, 5, 16, 3)):
        got = run_positions(pipelines(), names, tmp / f"r{rep}", batch_size=bs)
        masks = [np.load(tmp / f"r{rep}" / "steps" / nm / "segment_nuclei" / "0000.npz")["arr_0"] for nm in names]
        if ref is None:
            ref = (got, masks)
            print(f"rep {rep} batch {bs}: reference, {sum(g[0].num_rows for g in got)} objects, {len(got[0][0].column_names)} columns", flush=True)
            continue
        bad = 0
        for i in range(N):
            if not np.array_equal(masks[i], ref[1][i]):
                bad += 1
                print(f"  position {i}: label images differ")
                continue
            for c in got[i][0].column_names:
                a, b = got[i][0][c].to_numpy(zero_copy_only=False), ref[0][i][0][c].to_numpy(zero_copy_only=False)
                if not np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")):
                    bad += 1
                    print(f"  position {i} column {c}: max |diff| {np.nanmax(np.abs(a.astype(float) - b.astype(float)))}")
                    break
        print(f"rep {rep} batch {bs}: {'identical' if not bad else str(bad) + ' positions differ'}", flush=True)
