#!/bin/bash
# A/B two builds of librealsr_hip.so on the GPU box, alternating processes:  tools/ab_perf.sh A.so B.so [rounds]
# (a build = realsr-ncnn-vulkan_amd/lib/librealsr_hip.so or a tools/build_variant.sh product under lib/exp/)
A=$1; B=$2; N=${3:-3}
export RSR_PERF_VARIANTS="${RSR_PERF_VARIANTS:-flow_flags=0}"
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $L"
    RSR_LIB=$L timeout 300 python tools/flow_diag.py perf 2>&1 | grep -E "ms/frame|conv5 by|->32 @1x|192->64"
  done
done
