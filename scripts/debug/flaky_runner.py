"""Diagnostic: the position-batched runner against single calls, repeated until the tables differ; prints what differs."""
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from aliby_amd import synth  # noqa: E402
from aliby_amd.parallel import run_positions  # noqa: E402
from aliby_amd.pipe import run_pipeline_and_post  # noqa: E402
from aliby_amd.pipe_builder import build_pipeline_steps  # noqa: E402
from test_gpu_configs import _keyed_override  # noqa: E402

n = 7
fovs = [synth.make_fov(2, 40 + i, shape=(224, 256), n_channels=3, n_target=10 + i) for i in range(n)]
override = _keyed_override(fovs)


def pipelines():
    out = []
    for f in fovs:
        p = build_pipeline_steps(channels_to_segment={"nuclei": 0}, channels_to_extract=[0, 1, 2], cp_measure_feature_kwargs={"texture": {"scale": 2}})
        p["steps"]["tile"]["image_kwargs"] = {"source": f["pixels"][None]}
        p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = dict(flows_override=override)
        out.append(p)
    return out


names = [f"P{i:02d}__1" for i in range(n)]
tmp = Path(tempfile.mkdtemp())
want = [run_pipeline_and_post(pipeline=p, pipeline_name=nm, output_path=tmp / "single")[0] for p, nm in zip(pipelines(), names)]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for bs in (3, 4):
        got = run_positions(pipelines(), names, tmp / f"b{rep}_{bs}", batch_size=bs)
        bad = False
        for i, nm in enumerate(names):
            prof = got[i][0]
            if prof.num_rows != want[i].num_rows:
                print(f"rep {rep} bs {bs} {nm}: rows {prof.num_rows} vs {want[i].num_rows}")
                bad = True
                continue
            def same(c):
                a, b = prof[c].to_numpy(zero_copy_only=False), want[i][c].to_numpy(zero_copy_only=False)
                return np.array_equal(a, b, equal_nan=(a.dtype.kind == "f"))

            cols = [c for c in prof.column_names if not same(c)]
            if cols:
                bad = True
                fam = sorted({c.split("/")[2] if c.count("/") >= 2 else c for c in cols})
                print(f"rep {rep} bs {bs} {nm}: {len(cols)} of {len(prof.column_names)} columns differ; families {fam}")
                c = cols[0]
                a, b = prof[c].to_numpy(zero_copy_only=False), want[i][c].to_numpy(zero_copy_only=False)
                print("   ", c, "got", a[:6], "want", b[:6], "rows differing", int((a != b).sum()), "of", len(a),
                      "same multiset", bool(np.array_equal(np.sort(a), np.sort(b))))
                print("    labels col equal", np.array_equal(prof["metadata_label"].to_numpy(), want[i]["metadata_label"].to_numpy()))
                rows = np.nonzero(a != b)[0]
                area_c = [c for c in prof.column_names if c.endswith("sizeshape/Area")][0]
                lab = prof["metadata_label"].to_numpy()
                mask = np.load(tmp / f"b{rep}_{bs}" / "steps" / nm / "segment_nuclei" / "0000.npz")["arr_0"]
                mask_s = np.load(tmp / "single" / "steps" / nm / "segment_nuclei" / "0000.npz")["arr_0"]
                print("    masks equal (batched vs single):", np.array_equal(mask, mask_s))
                for r in rows:
                    print(f"    row {r} label {lab[r]}: Area got {prof[area_c][int(r)].as_py()} want {want[i][area_c][int(r)].as_py()} true {(mask == lab[r]).sum()}")
        print(f"rep {rep} bs {bs}: {'MISMATCH' if bad else 'ok'}", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
