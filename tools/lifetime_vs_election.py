#!/usr/bin/env python
"""lifetime_vs_election.py hwid_dump.bin [batch_index=11] — per-workgroup lifetime of the deciding wavefronts of a -DRG_PROBE_HWID launch (tools/placement.py's
input: the LAST launch of `bench.py --steps 10 --warmup 2`, i.e. batch 11 of the config-3 stream) against the number of rounds in which that
workgroup's 64 groups hold at least one election row (RV_REQ .. TIMEOUT), recomputed from the stream. No GPU needed once the dump exists."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rafting_amd import workload  # noqa: E402

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:1024]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 11
life = (raw[:, 3].astype(np.int64) - raw[:, 2].astype(np.int64)) / 100.0          # s_memrealtime: 100 MHz
cfg = workload.config(3, 65536)
gen = workload.ReplayGenerator(cfg)
gen.initial_state()
for _ in range(batch + 1):
    b = gen.next_batch(64)
kind = (b.head["hdr"] & 15).reshape(64, 65536)
el = ((kind >= 4) & (kind <= 8)).reshape(64, 1024, 64).any(axis=2).sum(axis=0)
print("1024 workgroups of one 64-round launch: lifetime of the deciding wavefront mean %.1f min %.1f max %.1f us; rounds with an election row per workgroup mean %.1f min %d max %d"
      % (life.mean(), life.min(), life.max(), el.mean(), el.min(), el.max()))
print("correlation(lifetime, election rounds) = %.3f" % np.corrcoef(life, el)[0, 1])
k, c = np.linalg.lstsq(np.vstack([el, np.ones_like(el)]).T, life, rcond=None)[0]
print("least squares: lifetime = %.2f us + %.3f us per election round  =>  %.2f us per plain round, %.2f us per round with the election block" % (c, k, c / 64, c / 64 + k))
for lo, hi in ((0, 30), (30, 40), (40, 50), (50, 65)):
    m = (el >= lo) & (el < hi)
    if m.any():
        print("  workgroups with %2d..%2d such rounds: n = %3d, lifetime mean %.1f us" % (lo, hi - 1, m.sum(), life[m].mean()))
