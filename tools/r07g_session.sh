# one gpurun call: the driver's command on the bench.py that carries the long-launch leg (256 rounds per launch beside the 64 of `value`)
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r07g_bench_driver_command.json 2> gpurun_out/r07g_bench_driver_command.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r07g_bench_driver_command.json
python -c "
import json; d=json.load(open('gpurun_out/r07g_bench_driver_command.json')); print(d['long_launches']); print(d['roofline']['traffic'], d['roofline']['traffic_measured_in_this_run'], d['roofline']['valu']['insts_per_launch'])"
( time timeout 600 python bench.py > gpurun_out/r07g_bench_no_flags.json 2> gpurun_out/r07g_bench_no_flags.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r07g_bench_no_flags.json
