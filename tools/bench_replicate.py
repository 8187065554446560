#!/usr/bin/env python
"""Measure the N1 send-side kernel (rg::replicate_kernel) on HBM-resident buffers: rows/s and GB/s against the
HBM roofline. usage: python tools/bench_replicate.py [groups=65536] [iters=50]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rafting_amd import abi, engine, workload  # noqa: E402


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    cfg = workload.config(3, groups)
    gen = workload.ReplayGenerator(cfg)
    t = engine.Table(groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    t.load_state(gen.initial_state())
    t.submit(gen.next_batch(4))                                  # some traffic so the state is not pristine
    F = cfg.cluster - 1
    hb = engine.DeviceBuffer.from_host(t, np.ones(groups, dtype=np.uint8))
    fl = engine.DeviceBuffer.from_host(t, np.zeros(groups * F, dtype=np.uint16))
    head = engine.DeviceBuffer(t, groups * abi.SEND_HEAD_DT.itemsize)
    send = engine.DeviceBuffer(t, groups * F * abi.SEND_DT.itemsize)
    L = engine.lib()
    call = lambda: t._check(L.rg_replicate(t._h, groups, None, hb.ptr, fl.ptr, head.ptr, send.ptr, abi.MEM_DEVICE))  # noqa: E731
    for _ in range(5):
        call()
    t.sync()
    t.timing_begin()
    for _ in range(iters):
        call()
    ms = t.timing_end() / iters
    st = t.read_state()
    leaders = int(np.count_nonzero(st.role == abi.LEADER))
    # algorithmic bytes: every row reads its four scalar columns (64 B) and writes head + F sends (48 + 32F);
    # a leader row additionally reads its runs (64 B), F x {lastEpoch,nextIndex} (16F) and the gate inputs (1 + 2F)
    nbytes = groups * (64 + 48 + 32 * F) + leaders * (64 + 16 * F + 1 + 2 * F)
    gbps = nbytes / (ms * 1e-3) / 1e9
    print(json.dumps({"kernel": "rg::replicate_kernel<%d>" % F, "groups": groups, "leaders": leaders, "ms_per_launch": ms,
                      "rows_per_s": groups / (ms * 1e-3), "sends_per_s": leaders * F / (ms * 1e-3),
                      "algorithmic_bytes_per_launch": nbytes, "achieved_gbps": gbps, "frac_of_8TBps": gbps / 8000.0}))


if __name__ == "__main__":
    main()
