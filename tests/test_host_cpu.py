"""CPU-only check of the C++ host mirror's RaftLog stand-in (rafting_amd/host/host_unit.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_memory_log_behaves_like_rockslog():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rafting_amd", "host"), "all"], check=True)
    p = subprocess.run([os.path.join(ROOT, "build", "host_unit")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr
