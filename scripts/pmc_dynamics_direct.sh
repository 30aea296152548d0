#!/bin/bash
# HBM counters of the default bench with the dynamics' direct-gather form (ALIBY_DYN_DIRECT=1), FETCH_SIZE and WRITE_SIZE in their own
# passes (never combined with other trace domains); summary: scripts/summarise_profiles.py <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03v}
export ALIBY_DYN_DIRECT=1
timeout -k 10 300 python3 bench.py --inputs-only --inputs /tmp/aliby_inputs && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/pmc_${TAG}_fetch.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/pmc_${TAG}_write.log 2>&1
echo "exit $?"
