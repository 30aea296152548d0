#!/bin/bash
# One gpurun call of the round's evidence: GPU test suite, rocprofv3 passes (kernel trace + the two HBM counters, separately),
# exclusive non-network kernel times, the default bench line and a 10-step line.  Outputs under gpurun_out/; the summaries are
# made afterwards with scripts/summarise_profiles.py <tag> and copied into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02b}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1 && \
bash scripts/profile_bench.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1 && \
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api > gpurun_out/${TAG}_bench_steps10.json 2> gpurun_out/${TAG}_bench_steps10.err
echo "exit $?"; tail -3 gpurun_out/${TAG}_gpu_tests.log; tail -2 gpurun_out/${TAG}_profile.log; cut -c1-300 gpurun_out/${TAG}_bench.json; cut -c1-120 gpurun_out/${TAG}_bench_steps10.json
