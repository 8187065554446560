# one gpurun call: check of the committed tree after the bench legs changed (r06w) — GPU suite, smoke(), the driver's bench command with no flags and with --steps 20 --warmup 3
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06x_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06x_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py > gpurun_out/r06x_bench_no_flags.json 2> gpurun_out/r06x_bench_no_flags.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r06x_bench_no_flags.json
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r06x_bench_steps20.json 2> gpurun_out/r06x_bench_steps20.err ) 2>&1 | tail -3; python tools/benchline.py < gpurun_out/r06x_bench_steps20.json
