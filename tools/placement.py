#!/usr/bin/env python
"""placement.py dump.bin [workgroups] — digest of the raw counter slots a -DRG_PROBE_HWID library writes (RG_DUMP_COUNTERS=dump.bin):
where the dispatcher put the deciding and the I/O wavefront of every workgroup of a step32_kernel launch (HW_REG_HW_ID: SIMD / CU / SH / SE,
HW_REG_XCC_ID), how many deciding wavefronts each SIMD got, and how long the wavefronts ran by how many deciders shared their SIMD
(s_memrealtime: 100 MHz). Build: tools/build_variants.sh hwid -DRG_PROBE_HWID."""
import sys
from collections import Counter, defaultdict

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else int((raw[:, 0] != 0).sum())
raw = raw[:wgs]


def where(w):
    hw, xcc = int(w) & 0xFFFFFFFF, (int(w) >> 32) & 0xF
    return (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15), (hw >> 4) & 3, hw & 15       # (xcc, se, sh, cu), simd, wave slot


dec_per_simd, io_per_simd, wg_per_cu = Counter(), Counter(), Counter()
pair = Counter()
for r in raw:
    cu_d, simd_d, _ = where(r[0])
    cu_i, simd_i, _ = where(r[1])
    dec_per_simd[(cu_d, simd_d)] += 1
    io_per_simd[(cu_i, simd_i)] += 1
    wg_per_cu[cu_d] += 1
    pair[(simd_d, simd_i, cu_d == cu_i)] += 1
print("workgroups %d on %d CUs (%d XCCs); workgroups per CU: %s" % (wgs, len(wg_per_cu), len({c[0] for c in wg_per_cu}), dict(Counter(wg_per_cu.values()))))
print("deciding wavefronts per SIMD (SIMDs that got any): %s; SIMDs with none: %d of %d" % (
    dict(sorted(Counter(dec_per_simd.values()).items())), 4 * len(wg_per_cu) - len(dec_per_simd), 4 * len(wg_per_cu)))
print("I/O wavefronts per SIMD: %s" % dict(sorted(Counter(io_per_simd.values()).items())))
print("(SIMD of decider, SIMD of I/O, same CU): %s" % dict(sorted(pair.items())))
by_share = defaultdict(list)
for r in raw:
    cu_d, simd_d, _ = where(r[0])
    by_share[dec_per_simd[(cu_d, simd_d)]].append((int(r[3]) - int(r[2])) / 100.0)
for k in sorted(by_share):
    v = np.array(by_share[k])
    print("deciders sharing a SIMD with %d decider(s) in all: n=%d  lifetime us mean %.1f  min %.1f  max %.1f" % (k, len(v), v.mean(), v.min(), v.max()))
t0 = int(raw[:, 2].min())
print("launch: first decider start -> last decider end %.1f us; decider starts spread %.1f us" % ((int(raw[:, 3].max()) - t0) / 100.0, (int(raw[:, 2].max()) - t0) / 100.0))
