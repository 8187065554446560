#!/bin/bash
# Builds experiment variants of the library next to the shipped one (git-ignored, they travel to the GPU box with gpurun):
#   tools/build_variants.sh prio "-DRG_DECIDE_PRIO"   ->  rafting_amd/libraftgpu_prio.so
# then on the GPU box:  bash tools/exp_ab.sh rafting_amd/libraftgpu.so rafting_amd/libraftgpu_prio.so [bench args]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rafting_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp "$@" -x hip -shared \
    -Wl,-soname,libraftgpu.so -o ../libraftgpu_$NAME.so rg_kernels.hip raftgpu.cpp
ls -la ../libraftgpu_$NAME.so
