#!/bin/bash
# round-3 GPU session 2: what bounds the dense-block convs?  DMA ablations (weights / patches / stores) and the Winograd
# instruction-mix builds, per conv class (tools/flow_diag.py perf prints the table)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
export TMPDIR=/tmp
L=$R/realsr-ncnn-vulkan_amd/lib
RSR_PERF_VARIANTS="dbg=0;dbg=64;dbg=128;dbg=4;dbg=68;dbg=1;dbg=0" timeout 600 python tools/flow_diag.py perf > $O/ablate.log 2>&1; echo "ablate rc=$?"
grep -E "ms/frame|->32 @1x|192->64|64->64|64->3 " $O/ablate.log
for v in thin16 thin24; do
  RSR_LIB=$L/exp/$v.so RSR_PERF_VARIANTS="dbg=0;dbg=0" timeout 300 python tools/flow_diag.py perf > $O/$v.log 2>&1; echo "$v rc=$?"
  grep -E "ms/frame|->32 @1x|192->64" $O/$v.log
done
RSR_PERF_VARIANTS="dbg=0" timeout 300 python tools/flow_diag.py perf > $O/base2.log 2>&1; grep -E "ms/frame|->32 @1x|192->64" $O/base2.log
