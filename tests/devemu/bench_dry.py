"""Runs bench.py unchanged against the host emulation of the kernels (tests/test_devemu_cpu.py): torch's "is a GPU there" calls are
stubbed, everything else — staging, the timed loop, the three host-memory legs, the CPU baseline and the translated-reference check, the
JSON line — is bench.py's own code on tiny sizes. TEST INFRASTRUCTURE: the numbers it prints mean nothing."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
# extra arguments (e.g. --gpus 2 --device 0 under torch.distributed.run: the emulation has one "device") are passed through
sys.argv = ["bench.py", "--groups-per-gpu", "256", "--rounds", "4", "--steps", "2", "--warmup", "1", "--pcie-batches", "3", "--cpu-batches", "1", "--copy-bytes", str(1 << 16), "--adverse-batches", "3", "--index-base-batches", "3", "--tick-batches", "11", "--long-launch-rounds", "8", "--long-launch-batches", "3"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
