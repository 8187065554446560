# one gpurun call: the round's last check of the committed tree — GPU suite, smoke(), a 7-minute differential soak through every step kernel, the driver's bench command
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06t_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06t_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 700 python tools/soak.py 420 > gpurun_out/r06t_soak.log 2>&1; tail -4 gpurun_out/r06t_soak.log
timeout 600 python bench.py > gpurun_out/r06t_bench_no_flags.json 2> gpurun_out/r06t_bench_no_flags.err; python tools/benchline.py < gpurun_out/r06t_bench_no_flags.json
