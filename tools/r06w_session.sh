# one gpurun call: (1) the long-lived-groups leg (index bases at 2^40 - 1) over 20 timed launches instead of 4, next to the plain leg of the same run;
# (2) the rounds sweep of profiles/r06i continued to 64, 128, 256 rounds per launch (the fixed cost of a launch as a share of it)
B="python bench.py --no-cpu-baseline --no-pcie --tick-batches 0 --no-pmc --no-adverse --no-int64-pass"
line='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; l=d.get("long_lived_groups") or {}
print("%s %.4f ms value %.3e | long-lived %s ms over %s launches, wide workgroups %s | golden %s" % (sys.argv[1], r["avg_kernel_ms"], d["value"], l.get("avg_kernel_ms"), l.get("launches"), l.get("int64_body_workgroups"), d["golden"]))'
for i in 1 2 3; do
  $B --steps 20 --warmup 3 --index-base-batches 23 2>>gpurun_out/r06w.err | tee -a gpurun_out/r06w_long_lived.jsonl | python -c "$line" "c3+bases"; done
for R in 64 128 256; do
  $B --steps 10 --warmup 2 --index-base-batches 0 --rounds $R 2>>gpurun_out/r06w.err | tee -a gpurun_out/r06w_rounds.jsonl | python -c "$line" "rounds=$R"; done
