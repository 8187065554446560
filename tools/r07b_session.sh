# one gpurun call: the driver's bench command on the final bench.py (copy yardstick in front of every timed leg): no flags, and --steps 20 --warmup 3
( time timeout 600 python bench.py > gpurun_out/r07b_bench_no_flags.json 2> gpurun_out/r07b_bench_no_flags.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r07b_bench_no_flags.json
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r07b_bench_steps20.json 2> gpurun_out/r07b_bench_steps20.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r07b_bench_steps20.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
