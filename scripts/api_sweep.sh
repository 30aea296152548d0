# value_api against the writer thread / writer process split (through-the-API leg of bench.py); run on the GPU box
for cfg in "4 12" "6 10" "8 8" "8 10" "6 12" "10 6"; do set -- $cfg; w=$1; p=$2
ALIBY_WRITERS=$w ALIBY_WRITER_PROCS=$p timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --api-fovs 768 > gpurun_out/sweep_w${w}_p${p}.json 2>>gpurun_out/sweep.err || break
python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_w${w}_p${p}.json").read().strip().splitlines()[-1])
m=d["api"]["main_thread"]
print("writers $w procs $p: value_api", d["value_api"], "device_steps_s", m["device_steps_s"], "drain", m["drain_writers_s"], {k: round(v, 2) for k, v in m["writer_thread_seconds"].items()})
PY
done
