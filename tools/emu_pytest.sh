#!/bin/bash
# runs pytest on the host emulation of the kernels the way tests/test_devemu_cpu.py does:  tools/emu_pytest.sh waves|serial [pytest args]
ROOT=$(cd "$(dirname "$0")/.." && pwd); EMU=$ROOT/tests/devemu
MODE=$1; shift
if [ "$MODE" = waves ]; then
  env RG_LIB=$EMU/libraftgpu_emu.so RG_ALLOW_HOST_EMULATION=1 RG_EMU_WAVES=1 RG_SPLIT=1 PYTHONPATH=$ROOT python -m pytest $EMU/emu_cases_waves.py -q -p no:cacheprovider "$@"
else
  env RG_LIB=$EMU/libraftgpu_emu.so RG_ALLOW_HOST_EMULATION=1 RG_SPLIT=0 PYTHONPATH=$ROOT python -m pytest $EMU/emu_cases.py -q -p no:cacheprovider "$@"
fi
