#!/bin/bash
# bench lines of the other BASELINE.json configurations + the K-loop / K-split network error record (one gpurun call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02b}
timeout -k 10 300 python -m pytest tests/test_gpu_segment.py -x -q -k fused_unet_matches -s > gpurun_out/${TAG}_unet_err.log 2>&1 && \
timeout -k 10 400 python bench.py --config 1 > gpurun_out/${TAG}_c1.json 2> gpurun_out/${TAG}_c1.err && \
timeout -k 10 400 python bench.py --config 5 > gpurun_out/${TAG}_c5.json 2> gpurun_out/${TAG}_c5.err && \
timeout -k 10 400 python bench.py --config 4 > gpurun_out/${TAG}_c4.json 2> gpurun_out/${TAG}_c4.err
echo "exit $?"; grep -h "rel l2" gpurun_out/${TAG}_unet_err.log; for c in 1 5 4; do cut -c1-160 gpurun_out/${TAG}_c$c.json; done
