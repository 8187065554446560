# one gpurun call: same-box A/B of the shipped library (two rounds per hand-over) against the same sources built with one (-DRG_ROUNDS_PER_HANDOVER=1), then the compact-row GPU tests
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass"
for i in 1 2 3; do for L in libraftgpu_rpb1.so libraftgpu.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 2>>gpurun_out/r06g_ab.err | tee -a gpurun_out/r06g_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('c3 $L %.4f ms value %.3e golden %s' % (r['avg_kernel_ms'], d['value'], d['golden']))"; done; done
for C in "--config 5 --groups-per-gpu 65536" "--config 2" "--config 2f"; do for L in libraftgpu_rpb1.so libraftgpu.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 $C 2>>gpurun_out/r06g_ab.err | tee -a gpurun_out/r06g_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$C $L %.4f ms value %.3e' % (r['avg_kernel_ms'], d['value']))"; done; done
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r06g_pytest_gpu.log 2>&1; tail -4 gpurun_out/r06g_pytest_gpu.log
