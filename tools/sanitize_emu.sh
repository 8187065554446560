#!/bin/bash
# The product's device + host sources on the host emulation (tests/devemu), built with UBSan and then ASan, through the
# emulated-device cases: undefined behaviour (shifts, signed overflow, misaligned access) and memory errors in
# rg_device.hpp / rg_kernels.hip / raftgpu.cpp show up here without a GPU.   usage: tools/sanitize_emu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/rg_sanitize; mkdir -p $OUT/ubsan $OUT/asan
SRC="$ROOT/rafting_amd/csrc/rg_kernels.hip $ROOT/rafting_amd/csrc/raftgpu.cpp $ROOT/tests/devemu/emu_runtime.cpp"
INC="-I$ROOT/tests/devemu -I$ROOT/include"
g++ -O1 -g -std=c++17 -fPIC -shared -w -pthread -fsanitize=undefined -fno-sanitize-recover=undefined $INC -x c++ $SRC -o $OUT/ubsan/libraftgpu_emu.so
g++ -O1 -g -std=c++17 -fPIC -shared -w -pthread -fsanitize=address $INC -x c++ $SRC -o $OUT/asan/libraftgpu_emu.so
cd $ROOT
echo "== UBSan"; RG_LIB=$OUT/ubsan/libraftgpu_emu.so RG_SPLIT=0 RG_ALLOW_HOST_EMULATION=1 python -m pytest tests/devemu/emu_cases.py -q -x -p no:cacheprovider | tail -2
echo "== UBSan, emulated wavefronts (the compact-row kernel: rg_tier1n.hpp's sign words and shifts, the I/O wavefront's tables)"
RG_LIB=$OUT/ubsan/libraftgpu_emu.so RG_SPLIT=1 RG_EMU_WAVES=1 RG_ALLOW_HOST_EMULATION=1 python -m pytest tests/devemu/emu_cases_waves.py -q -x -p no:cacheprovider | tail -2
# the "split kernel is refused" case throws a C++ exception through the preloaded runtime: left out under ASan
echo "== ASan"; LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 RG_LIB=$OUT/asan/libraftgpu_emu.so RG_SPLIT=0 RG_ALLOW_HOST_EMULATION=1 \
    python -m pytest tests/devemu/emu_cases.py -q -x -p no:cacheprovider -k "not refused" | tail -2
echo "== ASan, emulated wavefronts (two-wavefront kernel, counters, timer compaction)"
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 RG_LIB=$OUT/asan/libraftgpu_emu.so RG_SPLIT=1 RG_EMU_WAVES=1 RG_ALLOW_HOST_EMULATION=1 \
    python -m pytest tests/devemu/emu_cases_waves.py -q -x -p no:cacheprovider | tail -2
