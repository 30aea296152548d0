#!/bin/bash
# PMC passes over the conv microbenchmark (run on the GPU box through gpurun); summaries go to gpurun_out/pmc_conv_*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_conv_$name -o run -- python3 scripts/bench_conv.py 96 > gpurun_out/pmc_conv_$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections
for name in ("sq1","sq2","fetch","write"):
    files=glob.glob(f"gpurun_out/pmc_conv_{name}/**/*counter_collection.csv", recursive=True)
    if not files: print(name,"no counter file"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k=r["Kernel_Name"]
        if "k_conv3x3" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in agg.items():
        print(name, k[:70], {c: round(sum(v)/len(v),1) for c,v in cs.items()}, "n=",len(next(iter(cs.values()))))
PY
