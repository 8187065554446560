#!/usr/bin/env python
"""Soak differential fuzz on the GPU box: many seeds x cluster shapes, HIP path vs oracle in lockstep with the hint
protocol, with and without the fast-path tier, through every step kernel in turn: wide rows on the two-wavefront kernel / on the
single-wavefront kernel, compact rows (rg_submit32: step32_kernel's 32-bit body), compact rows forced onto the 64-bit body, and the last two again with compact
outcome rows (rg_submit32c).
usage: python tools/soak.py [seconds=240] [routes, comma separated: default all six]
RG_SOAK_SEED=n starts the seeds at n instead of 1000 (a second soak that repeats the first one's seeds adds nothing); RG_SOAK_BIG=1 adds clusters of 8 .. 15
nodes to the shapes, on the wide-row routes only (the compact formats refuse them)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rafting_amd import abi, engine  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402

WIDE_SUBMIT = engine.Table.submit


def compact_submit(self, batch, out=None, fill=0):      # what tests/test_gpu_parity.py::route_through_compact installs
    if batch.hint is None and abi.batch_fits_32(batch) and self.cluster <= abi.MAX_COMPACT_CLUSTER:
        return self.submit32(batch, out, fill)
    return WIDE_SUBMIT(self, batch, out, fill)


def out32_submit(self, batch, out=None, fill=0):        # route_through_compact(out32=True): rg_submit32c + rg_outcome32_unpack, the raw rows held to their contract
    if batch.hint is None and abi.batch_fits_32(batch) and batch.gid is None and self.cluster <= abi.MAX_COMPACT_CLUSTER:
        before = self.read_state()
        raw = self.submit32c(batch, fill=fill)
        got, _ = engine.unpack32(raw, batch.rounds, batch.count, before.role_epoch)
        T.check_out32_rows(raw, got, before, self.read_state(), batch.rounds, batch.count)
        return got
    return compact_submit(self, batch, out, fill)


ROUTES = ("split", "compact", "single", "compact-forced-wide", "compact-out32", "compact-out32-forced-wide")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    routes = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ROUTES       # e.g. compact-out32,compact-out32-forced-wide
    t0 = time.time()
    seed, runs, rows, misses = int(os.environ.get("RG_SOAK_SEED", "1000")), 0, 0, 0
    seed0 = seed
    hist = np.zeros(256, dtype=np.int64)
    per_route = {}
    shapes = [(3, 0), (3, 2), (5, 0), (5, 3), (2, 0), (4, 1), (6, 2), (7, 6)]
    big = [(9, 4), (11, 10), (15, 0), (8, 7), (13, 6)] if os.environ.get("RG_SOAK_BIG") == "1" else []
    big_runs = 0
    while time.time() - t0 < budget:
        cluster, self_slot = shapes[runs % len(shapes)]
        pre_vote = (runs // len(shapes)) % 2 == 0
        os.environ["RG_FAST"] = "0" if runs % 5 == 4 else "1"
        route = routes[(runs // 3) % len(routes)]
        if big and route in ("split", "single") and runs % 3 == 2:       # every third run of a wide-row route: a cluster above seven nodes
            cluster, self_slot = big[big_runs % len(big)]
            big_runs += 1
        os.environ["RG_SPLIT"] = "0" if route == "single" else "1"
        os.environ["RG_FORCE_WIDE"] = "1" if route.endswith("forced-wide") else "0"
        engine.Table.submit = out32_submit if "out32" in route else (compact_submit if route.startswith("compact") else WIDE_SUBMIT)
        per_route[route] = per_route.get(route, 0) + 1
        # group counts that are not multiples of the wavefront size exercise the shadow lanes of the tail wavefront
        groups, rounds = ((1024, 150), (1000, 150), (257, 400), (65, 600))[runs % 4]
        _, _, _, h, m, _ = T._lockstep(groups, cluster, self_slot, pre_vote, rounds, seed, allow_miss=True)
        hist += h
        misses += m
        rows += groups * rounds
        runs += 1
        seed += 1
    seen = {int(i): int(c) for i, c in enumerate(hist) if c}
    print("soak ok: %d runs (%s; seeds %d .. %d; %d on clusters of 8 .. 15 nodes), %d rows, %d hinted rows, %.0f s; statuses %s"
          % (runs, per_route, seed0, seed - 1, big_runs, rows, misses, time.time() - t0, seen))


if __name__ == "__main__":
    main()
