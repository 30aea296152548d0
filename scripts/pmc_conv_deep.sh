#!/bin/bash
# PMC passes over the conv microbenchmark (run on the GPU box through gpurun); summaries go to gpurun_out/pmc_deep_*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_deep_$name -o run -- python3 scripts/bench_conv_deep.py 288 > gpurun_out/pmc_deep_$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections
name2waves = {}
f2 = glob.glob("gpurun_out/pmc_deep_sq2/**/*counter_collection.csv", recursive=True)
if f2:
    w = collections.defaultdict(list)
    for r in csv.DictReader(open(f2[0])):
        if r["Counter_Name"] == "SQ_WAVES" and "k_conv3x3_deep" in r["Kernel_Name"]: w[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name2waves = {k: sum(v)/len(v) for k, v in w.items()}
for name in ("sq1","sq2","fetch","write"):
    files=glob.glob(f"gpurun_out/pmc_deep_{name}/**/*counter_collection.csv", recursive=True)
    if not files: print(name,"no counter file"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k=r["Kernel_Name"]
        if "k_conv3x3_deep" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in agg.items():
        avg = {c: sum(v)/len(v) for c,v in cs.items()}
        print(name, k[:70], {c: round(v,1) for c,v in avg.items()}, "n=",len(next(iter(cs.values()))))
        if name == "sq1" and avg.get("SQ_WAVE_CYCLES") and name2waves.get(k):
            # persistent waves live for the whole launch: lifetime = SQ_WAVE_CYCLES (quad-cycles) * 4 / waves; 1024 SIMDs
            life = avg["SQ_WAVE_CYCLES"] * 4 / name2waves[k]
            print("   MFMA pipe busy: %.1f %% of the launch (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x %.0f cycles))" % (100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * life), life))
PY
