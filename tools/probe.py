#!/usr/bin/env python
"""digest of a bench.py line produced with a -DRG_PROBE library (tools/build_variants.sh probe -DRG_PROBE): ticks per round per workgroup
   I/O wavefront: publish | retire + fetch | barrier      deciding wavefront: event read | tier 1 | slow path + outcome write | barrier"""
import json
import sys
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c = list(d["counters"].values())
    cfg = d["config"]
    wgs = (cfg["groups_per_gpu"] + 63) // 64
    per = d["steps"] * cfg["rounds_per_step"] * wgs
    v = [x / per for x in c]
    print("io: publish %.0f  retire+fetch %.0f  barrier %.0f | decide: read %.0f  tier1 %.0f  slow+write %.0f  barrier %.0f | sum io %.0f decide %.0f | kernel_ms %.4f" % (
        v[0], v[1], v[2], v[4], v[5], v[6], v[7], sum(v[0:3]), sum(v[4:8]), d["roofline"]["avg_kernel_ms"]))
