# value_api against writer count and job size (through-the-API leg of bench.py)
for w in 8 12; do for n in 256 768; do
ALIBY_WRITERS=$w python bench.py --steps 2 --warmup 1 --no-cpu-baseline --api-fovs $n > gpurun_out/r02f_api_w${w}_n${n}.json 2>>gpurun_out/r02f.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02f_api_w${w}_n${n}.json").read().strip().splitlines()[-1])
print("writers $w n $n", d["value"], d["value_api"], d["api"]["main_thread"])
PY
done; done
