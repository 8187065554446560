# round 6's evidence session (one gpurun call): GPU suite, the evidence pass of tools/prof_r6.sh, the 1000-tick latency distribution
timeout 1300 python -m pytest tests -m gpu -q > gpurun_out/r06q_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06q_pytest_gpu.log
bash tools/prof_r6.sh r06q > gpurun_out/r06q_prof.log 2>&1; tail -3 gpurun_out/r06q_prof.log
python tools/benchline.py < gpurun_out/bench_r06q_default.json
timeout 500 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-int64-pass --no-adverse --index-base-batches 0 --no-pmc --tick-batches 1010 > gpurun_out/r06q_tick_latency_1000.json 2> gpurun_out/r06q_tick.err
python -c "
import json; d=json.load(open('gpurun_out/r06q_tick_latency_1000.json')); print(d['tick_latency'])"
bash tools/r06n_session.sh
