#!/bin/bash
# builds tests/devemu/libraftgpu_emu.so (the host emulation of the kernels, test infrastructure) exactly as tests/test_devemu_cpu.py does
cd "$(dirname "$0")/../tests/devemu" && g++ -O1 -g0 -std=c++17 -fPIC -shared -w -pthread -I. -I../../include -x c++ ../../rafting_amd/csrc/rg_kernels.hip ../../rafting_amd/csrc/raftgpu.cpp emu_runtime.cpp -o libraftgpu_emu.so
