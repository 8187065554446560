# one gpurun call: the round's last check of the committed tree — GPU suite and smoke()
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r07h_pytest_gpu.log 2>&1; tail -3 gpurun_out/r07h_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
