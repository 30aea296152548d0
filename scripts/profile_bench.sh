#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box through gpurun): kernel trace + stats, then the two HBM
# counters in their own passes (never combined with other trace domains).  Summaries: scripts/summarise_profiles.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r01c}
# inputs are generated (fork pool) by an UNPROFILED command; the profiled ones load them and never fork
timeout -k 10 300 python3 bench.py --inputs-only --inputs /tmp/aliby_inputs && \
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/prof_${TAG}.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/pmc_${TAG}_fetch.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/pmc_${TAG}_write.log 2>&1
python3 scripts/summarise_profiles.py ${TAG}
