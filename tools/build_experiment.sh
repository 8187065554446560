#!/bin/bash
# Builds an experiment variant of the library from a PATCHED COPY of rafting_amd/csrc (the product sources stay what the evidence passes measured):
#   tools/build_experiment.sh class_ballots -DRG_CLASS_BALLOTS   ->  rafting_amd/libraftgpu_class_ballots.so   (git-ignored; travels with gpurun)
# patches live in tools/experiments/<name>.patch (unified diffs against the repository root)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/rafting_amd $TMP/include
cp -r $ROOT/rafting_amd/csrc $TMP/rafting_amd/csrc
cp $ROOT/include/*.h $TMP/include/
(cd $TMP && patch -p1 -s < $ROOT/tools/experiments/$NAME.patch)
cd $TMP/rafting_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp "$@" -x hip -shared \
    -Wl,-soname,libraftgpu.so -o $ROOT/rafting_amd/libraftgpu_$NAME.so rg_kernels.hip raftgpu.cpp 2>&1 | grep -E "error" || true
ls -la $ROOT/rafting_amd/libraftgpu_$NAME.so
rm -rf $TMP
