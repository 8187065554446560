#!/usr/bin/env python
"""Soak differential fuzz on the GPU box: many seeds x cluster shapes, HIP path vs oracle in lockstep with the hint
protocol, both with and without the fast-path tier. usage: python tools/soak.py [seconds=240]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rafting_amd import abi  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    t0 = time.time()
    seed, runs, rows, misses = 1000, 0, 0, 0
    hist = np.zeros(256, dtype=np.int64)
    shapes = [(3, 0), (3, 2), (5, 0), (5, 3), (2, 0), (4, 1), (6, 2), (7, 6)]
    while time.time() - t0 < budget:
        cluster, self_slot = shapes[runs % len(shapes)]
        pre_vote = (runs // len(shapes)) % 2 == 0
        os.environ["RG_FAST"] = "0" if runs % 5 == 4 else "1"
        # group counts that are not multiples of the wavefront size exercise the shadow lanes of the tail wavefront
        groups, rounds = ((1024, 150), (1000, 150), (257, 400), (65, 600))[runs % 4]
        _, _, _, h, m, _ = T._lockstep(groups, cluster, self_slot, pre_vote, rounds, seed, allow_miss=True)
        hist += h
        misses += m
        rows += groups * rounds
        runs += 1
        seed += 1
    seen = {int(i): int(c) for i, c in enumerate(hist) if c}
    print("soak ok: %d runs, %d rows, %d hinted rows, %.0f s; statuses %s" % (runs, rows, misses, time.time() - t0, seen))


if __name__ == "__main__":
    main()
