# one gpurun call: same-box A/B of a fast exit of the election block for rounds whose election rows are all vote replies that convert nobody (libraftgpu.so) against the commit before (libraftgpu_prev.so), then the GPU suite
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-int64-pass"
line='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s %s %.4f ms value %.3e adverse %s golden %s" % (sys.argv[1], r["library"], r["avg_kernel_ms"], d["value"], d.get("value_adverse_mix"), d["golden"]))'
for i in 1 2 3; do for L in libraftgpu_prev.so libraftgpu.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 2>>gpurun_out/r06v_ab.err | tee -a gpurun_out/r06v_ab.jsonl | python -c "$line" c3; done; done
for C in "--config 5 --groups-per-gpu 65536" "--config 4 --groups-per-gpu 131072" "--config 5 --groups-per-gpu 131072"; do for L in libraftgpu_prev.so libraftgpu.so libraftgpu_prev.so libraftgpu.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --no-adverse --steps 20 --warmup 3 $C 2>>gpurun_out/r06v_ab.err | tee -a gpurun_out/r06v_ab.jsonl | python -c "$line" "$C"; done; done
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06v_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06v_pytest_gpu.log
