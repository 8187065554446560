# one gpurun call: the driver's exact command of round 5 on the committed tree
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r07f_bench_driver_command.json 2> gpurun_out/r07f_bench_driver_command.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r07f_bench_driver_command.json
