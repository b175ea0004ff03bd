#!/bin/bash
# Build an experimental variant of librealsr_hip.so:  tools/build_variant.sh NAME "-DRSR_EXP_FOO=1 ..."
# -> realsr-ncnn-vulkan_amd/lib/exp/NAME.so   (load with RSR_LIB=<path>; lib/ is git-ignored but travels to the GPU box)
set -e
NAME=$1; shift
D=$(cd "$(dirname "$0")/../realsr-ncnn-vulkan_amd" && pwd)
make -s -C $D/csrc ../lib/librealsr_hip.so
mkdir -p $D/lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c $D/csrc/kernels.hip -o $D/lib/exp/$NAME.kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib/exp/$NAME.so $D/lib/exp/$NAME.kernels.o $D/lib/obj/engine.o $D/lib/obj/capi.o $D/lib/obj/model.o
rm -f $D/lib/exp/$NAME.kernels.o
ls -la $D/lib/exp/$NAME.so
