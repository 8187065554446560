# one gpurun call: does the timed region run faster when the plain-copy yardstick (1 GB x 10, ~4 ms of device work) runs right before the warm-up launches
# instead of after the timed region — i.e. are the first timed launches still paying for the idle seconds of host-side staging in front of them?
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass"
line='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s kernel %.4f ms ms/step %.4f value %.3e copy %.0f GB/s golden %s" % (sys.argv[1], r["avg_kernel_ms"], d["ms_per_step"], d["value"], r["measured_copy_gbps"], d["golden"]))'
for i in 1 2 3; do for M in 0 1; do for K in "--steps 10 --warmup 2" "--steps 20 --warmup 3"; do
  RG_BENCH_COPY_FIRST=$M $B $K 2>>gpurun_out/r07a.err | tee -a gpurun_out/r07a_copy_first_ab.jsonl | python -c "$line" "copy_first=$M $K"; done; done; done
