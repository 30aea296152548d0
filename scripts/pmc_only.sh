cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}
# inputs are generated (fork pool) by an UNPROFILED command; the profiled ones load them and never fork
timeout -k 10 300 python3 bench.py --inputs-only --inputs /tmp/aliby_inputs && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/pmc_${TAG}_fetch.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api --inputs /tmp/aliby_inputs > gpurun_out/pmc_${TAG}_write.log 2>&1
echo "pmc exit $?"
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02d_driver_style.json 2> gpurun_out/r02d_driver_style.err
echo "bench exit $?"; grep -E "Elapsed|Maximum resident" gpurun_out/r02d_driver_style.err; cut -c1-140 gpurun_out/r02d_driver_style.json
