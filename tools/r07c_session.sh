# one gpurun call: is 40 ms of copying in front of the warm-up enough? 10 against 100 and 400 iterations of the 1 GB copy, alternating
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass"
line='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s kernel %.4f ms ms/step %.4f value %.3e copy %.0f GB/s" % (sys.argv[1], r["avg_kernel_ms"], d["ms_per_step"], d["value"], r["measured_copy_gbps"]))'
for i in 1 2 3; do for M in 10 100 400; do
  RG_BENCH_COPY_ITERS=$M $B --steps 20 --warmup 3 2>>gpurun_out/r07c.err | tee -a gpurun_out/r07c_copy_iters_ab.jsonl | python -c "$line" "iters=$M"; done; done
