# one gpurun call: the final bench.py once more in its N > 1 form (two ranks on the one GPU, under torch.distributed.run), and the two config-2 views
line='import json,sys
d=json.loads(sys.stdin.read()); print("%s n_gpus %s value %.3e ms/step %.4f per_gpu %s golden %s" % (sys.argv[1], d["n_gpus"], d["value"], d["ms_per_step"], [round(g["avg_kernel_ms"],4) for g in d["per_gpu"]], d["golden"]))'
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --device 0 --steps 10 --warmup 2 2>gpurun_out/r07d_trun.err | tee gpurun_out/r07d_two_ranks_torchrun.json | python -c "$line" torchrun
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass --steps 20 --warmup 3"
for C in "--config 2 --groups-per-gpu 4096" "--config 2f --groups-per-gpu 4096" "--config 5 --groups-per-gpu 65536" "--config 4 --groups-per-gpu 131072" "--config 5 --groups-per-gpu 131072"; do
  $B $C 2>>gpurun_out/r07d.err | tee -a gpurun_out/r07d_other_configs.jsonl | python -c "$line" "$C"; done
