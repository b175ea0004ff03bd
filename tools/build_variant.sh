#!/bin/bash
# Build an experimental variant of librealsr_hip.so:  tools/build_variant.sh NAME FILE "-DRSR_EXPERIMENT -DRSR_FLOW_TRACE ..."
# (-DRSR_EXPERIMENT admits the instrumented / ablation code paths that the product build compiles out: conv_flow.hip RSR_ABL)
# FILE = kernels | conv_flow.  -> realsr-ncnn-vulkan_amd/lib/exp/NAME.so   (load with RSR_LIB=<path>; lib/ is git-ignored
# but travels to the GPU box)
set -e
NAME=$1; FILE=$2; shift; shift
D=$(cd "$(dirname "$0")/../realsr-ncnn-vulkan_amd" && pwd)
make -s -C $D/csrc ../lib/librealsr_hip.so
mkdir -p $D/lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$D/lib/obj/gen "$@" -c $D/csrc/$FILE.hip -o $D/lib/exp/$NAME.$FILE.o
OBJS=""
for o in kernels conv_flow engine capi group model; do
  if [ "$o" = "$FILE" ]; then OBJS="$OBJS $D/lib/exp/$NAME.$FILE.o"; else OBJS="$OBJS $D/lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib/exp/$NAME.so $OBJS -ldl
rm -f $D/lib/exp/$NAME.$FILE.o
ls -la $D/lib/exp/$NAME.so
