# one gpurun call: same-box A/B of prologue / epilogue variants of the step kernel (round 6), and the fixed cost of a launch (rounds sweep)
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass"
line='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s %s %.4f ms value %.3e golden %s" % (sys.argv[1], r["library"], r["avg_kernel_ms"], d["value"], d["golden"]))'
for i in 1 2 3; do for L in libraftgpu.so libraftgpu_ff.so libraftgpu_ta.so libraftgpu_ffta.so libraftgpu_all3.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 2>>gpurun_out/r06i_ab.err | tee -a gpurun_out/r06i_ab.jsonl | python -c "$line" c3; done; done
for C in "--config 5 --groups-per-gpu 65536" "--config 4 --groups-per-gpu 131072"; do for L in libraftgpu.so libraftgpu_ffta.so libraftgpu_all3.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 $C 2>>gpurun_out/r06i_ab.err | tee -a gpurun_out/r06i_ab.jsonl | python -c "$line" "$C"; done; done
for R in 1 2 4 8 16 32; do for L in libraftgpu.so libraftgpu_all3.so; do
  RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 --rounds $R 2>>gpurun_out/r06i_ab.err | tee -a gpurun_out/r06i_rounds.jsonl | python -c "$line" "rounds=$R"; done; done
# what the speculative follower-row loads cost in bytes
RG_LIB=$(pwd)/rafting_amd/libraftgpu_all3.so python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-adverse --no-int64-pass --steps 20 --warmup 3 2>>gpurun_out/r06i_ab.err | tee gpurun_out/r06i_all3_pmc.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('all3 traffic', r['traffic'], r['valu'])"
RG_LIB=$(pwd)/rafting_amd/libraftgpu_all3.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
