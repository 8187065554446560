"""Digest for the -DRG_PROFILE2 -DRG_PROFILE3 build: inside the slow-path visits of step_split_kernel (per lane that took one)."""
import json
import sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d["counters"]
        v = max(c["role_conversions"], 1)
        print("lane-visits %d (%.3f%% of rows)  per visit: LDS reads + entry %.0f  tier1.5 %.0f  general %.0f | wave barrier %.0f io-wait %.0f kernel_ms %.4f" % (
            v, 100.0 * v / max(c["rows"], 1), c["commit_advances"] / v, c["asserts"] / v, c["need_host"] / v, 0, 0, d["roofline"]["avg_kernel_ms"]))
