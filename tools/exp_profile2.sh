#!/bin/bash
# Experiment (GPU box): per-round cycle breakdown of step_split_kernel (-DRG_PROFILE2 build in rafting_amd/libraftgpu_prof2.so)
L=$(pwd)/rafting_amd
for ov in "" "leader_frac=0.0;p_timeout=0.0" "leader_frac=1.0;p_higher_term=0.0"; do
  echo "== [$ov]"
  RG_LIB=$L/libraftgpu_prof2.so python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 ${ov:+--override "$ov"} 2>/dev/null | python tools/cyc2.py
done
echo "== config 5, 65536"
RG_LIB=$L/libraftgpu_prof2.so python bench.py --no-cpu-baseline --no-pcie --config 5 --groups-per-gpu 65536 --steps 10 --warmup 2 2>/dev/null | python tools/cyc2.py
