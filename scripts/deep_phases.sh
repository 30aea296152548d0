#!/bin/bash
# Phase elimination on the deep K-loop convolution (timing only, results wrong): one diagnostic build of the library per
# DC_HACK value (compile-time switches: runtime ones wreck the unrolled MFMA loop), then scripts/bench_conv_deep.py on each.
# Part 1 (build container): bash scripts/deep_phases.sh build 0 1 2 ...   Part 2 (GPU box): bash scripts/deep_phases.sh run 0 1 2 ...
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  cd aliby_amd/csrc
  OBJS=$(ls *.o | grep -v nn_conv_deep.o)
  for h in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $( [[ $h == ls* ]] && echo -DDL_HACK=${h#ls} || echo -DDC_HACK=${h%%_*} ) $( [[ $h == *_* ]] && echo -DDC_REQ_AT=${h##*_} ) -c nn_conv_deep.hip -o /tmp/deep_hack_$h.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaliby_hip_deephack_$h.so $OBJS /tmp/deep_hack_$h.o -lz -ldl -lpthread
  done
else
  for h in "$@"; do
    echo "DC_HACK=$h"
    ALIBY_DEEP_LS=$( [[ $h == ls* ]] && echo 1 || echo 0 ) ALIBY_HIP_LIB=$PWD/aliby_amd/libaliby_hip_deephack_$h.so timeout -k 10 120 python3 scripts/bench_conv_deep.py 288 2>/dev/null | grep -E "deep conv"
  done
fi
