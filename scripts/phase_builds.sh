#!/bin/bash
# Phase elimination by diagnostic builds (timing only, results wrong): one copy of the library per value of a compile-time macro
# of one source file, then a microbenchmark on each through ALIBY_HIP_LIB.
#   build container:  bash scripts/phase_builds.sh build nn_conv.hip CP_HACK 0 1 2 4 ...
#   GPU box:          bash scripts/phase_builds.sh run CP_HACK "python3 scripts/bench_conv_pair.py 288" 0 1 2 4 ...
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  src=$1; macro=$2; shift 2
  cd aliby_amd/csrc
  OBJS=$(ls *.o | grep -v "${src%.hip}.o")
  for h in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -D$macro=$h -c $src -o /tmp/phase_${macro}_$h.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaliby_hip_phase_${macro}_$h.so $OBJS /tmp/phase_${macro}_$h.o -lz -ldl -lpthread
  done
else
  macro=$1; cmd=$2; shift 2
  for h in "$@"; do
    echo "$macro=$h"
    ALIBY_HIP_LIB=$PWD/aliby_amd/libaliby_hip_phase_${macro}_$h.so timeout -k 10 120 $cmd 2>/dev/null | grep -E "us|TFLOP" || true
  done
fi
