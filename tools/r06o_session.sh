# one gpurun call: the tick tests on the look-back expiry (bounded time), then the kernel trace of the tick leg
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "resident or once_per_tick or timers or health" 2>&1 | tail -3
bash tools/r06n_session.sh
