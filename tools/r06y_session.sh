# one gpurun call: the N > 1 forms of the driver's command with both ranks on the one GPU of this box (--device 0: tests only) — self-launched and under
# torch.distributed.run, config 4 and config 5 shards, every rank's first launch against the reference-made digest of its shard — then a differential soak
line='import json,sys
d=json.loads(sys.stdin.read()); print("%s n_gpus %s launcher %s value %.3e ms/step %.4f per_gpu %s golden %s" % (sys.argv[1], d["n_gpus"], d.get("launcher"), d["value"], d["ms_per_step"], [round(g["avg_kernel_ms"],4) for g in d["per_gpu"]], d["golden"]))'
timeout 600 python bench.py --gpus 2 --device 0 --steps 5 --warmup 2 2>gpurun_out/r06y_self.err | tee gpurun_out/r06y_two_ranks_self_launched.json | python -c "$line" self
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --device 0 --steps 5 --warmup 2 2>gpurun_out/r06y_trun.err | tee gpurun_out/r06y_two_ranks_torchrun.json | python -c "$line" torchrun
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --device 0 --steps 5 --warmup 2 --config 5 2>>gpurun_out/r06y_trun.err | tee gpurun_out/r06y_two_ranks_torchrun_config5.json | python -c "$line" torchrun-c5
timeout 1000 python tools/soak.py ${SOAK:-780} > gpurun_out/r06y_soak.log 2>&1; tail -4 gpurun_out/r06y_soak.log
