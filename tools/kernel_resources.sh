#!/bin/bash
# VGPRs / scratch of every conv3x3_flow instantiation (compiler remarks; no GPU needed):  tools/kernel_resources.sh [extra -D flags]
D=$(cd "$(dirname "$0")/../realsr-ncnn-vulkan_amd" && pwd)
make -s -C $D/csrc ../lib/obj/gen/conv_flow_hooks.inc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$D/lib/obj/gen "$@" -Rpass-analysis=kernel-resource-usage \
  -I$D/csrc -c ${RSR_SRC:-$D/csrc/conv_flow.hip} -o /dev/null 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize" | sed -E 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - | sed -E 's/Function Name: _ZN3rsr12conv3x3_flowI//; s/EEvNS_8ConvArgsE//'
