#!/bin/bash
# SQ counters of the per-object (non-network) kernels of one bench step (run on the GPU box through gpurun).
# Feature families on ONE stream (ALIBY_FEATURE_STREAMS=1) so that a kernel's counters are its own; two counter sets in two
# passes (8 SQ slots per pass); inputs come from a cache written by an unprofiled command: the profiled process never forks.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}
export ALIBY_FEATURE_STREAMS=1
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --host-procs 1 --inputs /tmp/aliby_inputs"
timeout -k 10 300 python3 bench.py --inputs-only --inputs /tmp/aliby_inputs && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS \
  --output-format csv -d gpurun_out/pmc_${TAG}_feat1 -o run -- python3 $ARGS > gpurun_out/pmc_${TAG}_feat1.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS \
  --output-format csv -d gpurun_out/pmc_${TAG}_feat2 -o run -- python3 $ARGS > gpurun_out/pmc_${TAG}_feat2.log 2>&1
echo "pmc exit $?"
python3 scripts/summarise_pmc_features.py ${TAG}
