#!/bin/bash
# No GPU needed: compiles the step kernels for F = 4 only and reports the instruction count of step32_kernel's deciding-wavefront loop from its
# header to the branch that skips the general handlers ("spine": tier 1 with its rare blocks inline) — the static proxy for ticks per round.
#   tools/spine.sh [extra hipcc flags]
set -e
OUT=${OUT:-/tmp/spine}; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-sched-strategy=max-ilp -DRG_BUILD_ONLY_F4 "$@" -S --cuda-device-only -o $OUT/rg.s $(dirname $0)/../rafting_amd/csrc/rg_kernels.hip 2>/dev/null
python3 $(dirname $0)/spine.py $OUT/rg.s _ZN2rg13step32_kernelILi4ELb0ELi1ELb0ELi1EEEvNS_10StepParamsE
python3 $(dirname $0)/spine.py $OUT/rg.s _ZN2rg13step32_kernelILi4ELb0ELi1ELb1ELi1EEEvNS_10StepParamsE      # compact outcome rows
