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
          "arena_wait", m["of_which_waiting_for_a_free_arena_s"], "threads", m["writers"], "cpu", m.get("cgroup_cpu"), m.get("writer_thread_seconds"), json.dumps(l["api_split_ms_per_fov"]))
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
    nofiles) run nofiles ALIBY_ABLATE=files ;;
    nomulti) run nomulti ALIBY_MULTI_STREAM=0 ;;
    autogc) run autogc ALIBY_MANAGE_GC=0 ;;
    sw1) run sw1 ALIBY_SWITCH_INTERVAL=1e-3 ;;
    trace) run trace ALIBY_RUNNER_TRACE=1
           python3 - <<PY
import json
l = json.loads(open("gpurun_out/api_ab_trace.json").read().strip().splitlines()[-1])
m = l["api"]["main_thread"]
allm = sorted(m["trace"], key=lambda x: x[1])
tr = [(lab, t) for lab, t in allm if "ingest thread" not in lab and not lab.startswith("w:")]
idx = [i for i, (lab, t) in enumerate(tr) if lab == "run_batch:arena acquired"] + [len(tr)]
print("gc_s", m.get("gc_s"), "drain", m["drain_writers_s"])
print([lab for lab, t in allm if lab.startswith("prepare:end")])
cols = ["dynamics:call", "dynamics:returned", "object_table:call", "extract_nuclei:returned", "extractmulti_nuclei:returned", "batch:steps done"]
print("batch  period | " + " | ".join(c[:22] for c in cols) + " | other marks")
for a, b in zip(idx, idx[1:]):
    seg = tr[a:b]
    t0 = seg[0][1]
    at = {lab: t for lab, t in seg}
    nxt = tr[b][1] if b < len(tr) else seg[-1][1]
    extra = [lab for lab, _ in seg if lab.startswith("arena")]
    w = [(lab, t) for lab, t in allm if lab.startswith("w:") and t0 <= t < nxt]
    for kind in ("w:npz begin", "w:npz end", "w:finish begin", "w:pivot there", "w:finish end"):
        ts = [t for lab, t in w if lab == kind]
        if ts:
            extra.append(f"{kind[2:]} {len(ts)}x {1e3 * (min(ts) - t0):.0f}..{1e3 * (max(ts) - t0):.0f}")
    print(f"{1e3 * (nxt - t0):12.1f} | " + " | ".join(f"{1e3 * (at[c] - t0):22.1f}" if c in at else " " * 22 for c in cols) + " | " + ",".join(extra))
PY
           ;;
    *) echo "unknown $t" ;;
  esac
done
