# A/B of one environment switch on the through-the-API leg, interleaved repeats (run on the GPU box): api_ab.sh VAR
VAR=${1:-ALIBY_CHUNKED_SUBMIT}
for rep in 1 2 3; do for v in 0 1; do
env $VAR=$v timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --api-fovs 768 > gpurun_out/ab_${v}_${rep}.json 2>>gpurun_out/ab.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${v}_${rep}.json").read().strip().splitlines()[-1])
m=d["api"]["main_thread"]
print("$VAR=$v rep $rep: value_api", d["value_api"], "device_steps_s", m["device_steps_s"], "drain", m["drain_writers_s"], "arena_wait", m.get("of_which_waiting_for_a_free_arena_s"))
PY
done; done
