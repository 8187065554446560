# one gpurun call: what the fixed cost of bench.py's timed region is made of (interrupt-driven against polled completion signals), 10 and 20 steps
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass --no-copy-bw"
for i in 1 2; do for S in 10 20; do for E in 1 0; do
  HSA_ENABLE_INTERRUPT=$E $B --steps $S --warmup 3 2>>gpurun_out/r06u.err | tee -a gpurun_out/r06u.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('interrupt=$E steps=$S value %.4e ms/step %.4f kernel %.4f fixed %.1f us' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], (d['ms_per_step']-r['avg_kernel_ms'])*$S*1e3))"
done; done; done
