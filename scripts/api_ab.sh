#!/bin/bash
# A/B of the through-the-API leg (run on the GPU box through gpurun): launch-path knobs, one bench line each.
cd "$GRAFT_REPO_ROOT"
python3 bench.py --inputs-only --inputs /tmp/aliby_inputs > /dev/null 2>&1
run() {  # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --inputs /tmp/aliby_inputs > gpurun_out/api_ab_$tag.json 2> gpurun_out/api_ab_$tag.err
  python3 - <<PY
import json
try:
    l = json.loads(open("gpurun_out/api_ab_$tag.json").read().strip().splitlines()[-1])
    m = l["api"]["main_thread"]
    print("$tag", "value", l["value"], "value_api", l["value_api"], "device_steps_s/batch", round(m["device_steps_s"] / m["batches"], 4), "drain", m["drain_writers_s"],
          "arena_wait", m["of_which_waiting_for_a_free_arena_s"], "threads", m["writers"], json.dumps(l["api_split_ms_per_fov"]))
except Exception as e:
    print("$tag failed", e)
PY
}
for t in "$@"; do
  case $t in
    fast) run fast ALIBY_FAST_LAUNCH=1 ;;
    slow) run slow ALIBY_FAST_LAUNCH=0 ;;
    graph) run graph ALIBY_NET_GRAPH=1 ;;
    w8) run w8 ALIBY_WRITERS=8 ;;
    w10) run w10 ALIBY_WRITERS=10 ;;
    w12) run w12 ALIBY_WRITERS=12 ;;
    nodefer) run nodefer ALIBY_DEFER_SUBMITS=0 ;;
    w20) run w20 ALIBY_WRITERS=20 ;;
    sw1) run sw1 ALIBY_SWITCH_INTERVAL=1e-3 ;;
    trace) run trace ALIBY_RUNNER_TRACE=1
           python3 - <<PY
import json
l = json.loads(open("gpurun_out/api_ab_trace.json").read().strip().splitlines()[-1])
tr = l["api"]["main_thread"]["trace"]
# the last two batches: label, ms since the previous mark
idx = [i for i, (lab, t) in enumerate(tr) if lab == "run_batch:arena acquired"]
for (lab, t), (_, t0) in zip(tr[idx[1] - 3:idx[3]], [tr[idx[1] - 4]] + tr[idx[1] - 3:idx[3] - 1]):
    print(f"{lab:40s} +{1e3 * (t - t0):8.2f} ms   t={1e3 * (t - tr[idx[1] - 3][1]):8.1f}")
PY
           ;;
    *) echo "unknown $t" ;;
  esac
done
