set -x
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r06f_pytest_gpu.log 2>&1; tail -4 gpurun_out/r06f_pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass"
for i in 1 2 3; do for L in libraftgpu.so libraftgpu_pf.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 2>>gpurun_out/r06f_ab.err | tee -a gpurun_out/r06f_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('c3 $L %.4f ms value %.3e' % (r['avg_kernel_ms'], d['value']))"; done; done
for C in "--config 5 --groups-per-gpu 65536" "--config 4"; do for L in libraftgpu.so libraftgpu_pf.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 $C 2>>gpurun_out/r06f_ab.err | tee -a gpurun_out/r06f_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$C $L %.4f ms value %.3e' % (r['avg_kernel_ms'], d['value']))"; done; done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-int64-pass --no-adverse --index-base-batches 0 --no-pmc --tick-batches 110 2>gpurun_out/r06f_tick.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['tick_latency'])"
