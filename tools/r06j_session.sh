# one gpurun call: the GPU suite on the library with the one-node tick, then the default bench line (its tick leg times the 4-, 2- and 1-node recordings)
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06j_pytest_gpu.log 2>&1; tail -4 gpurun_out/r06j_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06j_bench_default.json 2> gpurun_out/r06j_bench_default.err; tail -c 400 gpurun_out/r06j_bench_default.err
python tools/benchline.py < gpurun_out/r06j_bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/r06j_bench_default.json')); print(d['tick_latency'])"
