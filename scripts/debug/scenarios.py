"""run_positions against one run_pipeline_and_post per position in less usual set-ups (a debugging aid, not a test)."""
import sys, tempfile, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from pathlib import Path
from aliby_amd import synth
from aliby_amd.parallel import run_positions
from aliby_amd.pipe import run_pipeline_and_post
from aliby_amd.pipe_builder import build_pipeline_steps
from test_gpu_configs import _keyed_override
warnings.simplefilter("ignore")
tmp = Path(tempfile.mkdtemp())

def compare(tag, got, want):
    worst = [0.0]
    for i, (g, w) in enumerate(zip(got, want)):
        assert g[0].num_rows == w.num_rows > 0, (tag, i, g[0].num_rows, w.num_rows)
        for c in w.column_names:
            x, y = g[0][c].to_numpy(zero_copy_only=False), w[c].to_numpy(zero_copy_only=False)
            if x.dtype.kind != "f":
                assert np.array_equal(x, y), (tag, i, c)
            elif not np.array_equal(x, y, equal_nan=True):
                # (a batch sizes its workgroups for the largest object in it: sums can differ in the last bits)
                scale = max(float(np.nanmax(np.abs(y))), 1.0)  # (central moments of order 1 are sums that cancel to ~0)
                worst[0] = max(worst[0], float(np.nanmax(np.abs(x - y))) / scale)
                assert worst[0] < 1e-9, (tag, i, c, worst[0])
    print(tag, "ok; largest relative difference of a float column:", worst[0], flush=True)

# (float32 / uint8 SOURCES are refused by the tiler: the stager takes uint16 stacks, as the reference's microscopes deliver)
# 2. two object sets (nuclei + cell), the example-01 pattern, 4 positions
fovs = [synth.make_fov(2, 130 + i, shape=(224, 256), n_channels=3, n_target=8) for i in range(4)]
tab = {}
for f in fovs:
    tab[f["pixels"][0].max(axis=0).tobytes()] = synth.analytic_flows(f["nuclei"])
    tab[f["pixels"][2].max(axis=0).tobytes()] = synth.analytic_flows(f["cells"])
def ov2(x):
    host = x.cpu().numpy(); fl = [tab[host[i].tobytes()] for i in range(host.shape[0])]
    return torch.from_numpy(np.stack([a for a, _ in fl])).cuda(), torch.from_numpy(np.stack([b for _, b in fl])).cuda()
def pipes2():
    out = []
    for f in fovs:
        p = build_pipeline_steps(channels_to_segment={"nuclei": 0, "cell": 2}, channels_to_extract=[0, 1, 2], features_to_extract=("intensity", "sizeshape"))
        p["steps"]["tile"]["image_kwargs"] = {"source": f["pixels"][None]}
        for s in ("segment_nuclei", "segment_cell"):
            p["steps"][s]["segmenter_kwargs"]["setup_params"] = dict(flows_override=ov2)
        out.append(p)
    return out
names = [f"two{i}" for i in range(4)]
want = [run_pipeline_and_post(pipeline=p, pipeline_name=n, output_path=tmp / "s2")[0] for p, n in zip(pipes2(), names)]
compare("two object sets", run_positions(pipes2(), names, tmp / "b2", batch_size=4), want)
