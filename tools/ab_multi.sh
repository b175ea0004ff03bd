#!/bin/bash
# Frame time (C2, device API, unprofiled) of several builds of librealsr_hip.so, alternating processes:
#   tools/ab_multi.sh ROUNDS A.so B.so C.so ...      (a build = lib/librealsr_hip.so or a tools/build_variant.sh product under lib/exp/)
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    printf "%-60s " "$L"
    RSR_LIB=$L timeout 120 python tools/frame_time.py 2>&1 | tail -1
  done
done
