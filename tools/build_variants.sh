#!/bin/bash
# Builds experiment variants of the library next to the shipped one (git-ignored, they travel to the GPU box with gpurun):
#   tools/build_variants.sh prio "-DRG_DECIDE_PRIO"   ->  rafting_amd/libraftgpu_prio.so
# then on the GPU box:  bash tools/exp_ab.sh rafting_amd/libraftgpu.so rafting_amd/libraftgpu_prio.so [bench args]
# Known variants: -DRG_DECIDE_PRIO (issue priority for the deciding wavefront), -DRG_FLAG_SYNC (polled LDS counters instead of the barrier),
# -DRG_TIER15, -DRG_SPLIT_NARROW, -DRG_EXP_ONLY_WIDE, -DRG_PROFILE2 [-DRG_PROFILE3], -DRG_HWID, -DRG_EXP_NO_ACK, -DRG_EXP_NO_AE.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rafting_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp "$@" -x hip -shared \
    -Wl,-soname,libraftgpu.so -o ../libraftgpu_$NAME.so rg_kernels.hip raftgpu.cpp
ls -la ../libraftgpu_$NAME.so
