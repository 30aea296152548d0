#!/bin/bash
# Exclusive (single-stream) kernel times of the non-network part of the step: the default bench runs the feature families on
# four side streams, where rocprofv3's per-kernel durations overlap and over-count.  Run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02x}
export ALIBY_FEATURE_STREAMS=1
# inputs are generated (fork pool) by an UNPROFILED command; the profiled ones load them and never fork
timeout -k 10 300 python3 bench.py --inputs-only --inputs /tmp/aliby_inputs && \
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/prof_${TAG}.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_${TAG}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_normalize99" in r["Kernel_Name"]]
seg = rows[marks[-2]:marks[-1]]
agg = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if k.startswith(("k_conv", "k_first_conv", "k_fused", "k_out_head", "k_maxpool", "k_style", "k_nn", "k_tiles", "k_make_tiles", "k_average")):
        continue
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
with open("gpurun_out/${TAG}_exclusive_non_network.csv", "w") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "calls_per_step", "total_us", "avg_us", "percent"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, a[0], round(a[1] / 1e3, 1), round(a[1] / a[0] / 1e3, 2), round(100 * a[1] / tot, 2)])
print("non-network kernel time per step (single stream):", tot / 1e6, "ms")
PY
