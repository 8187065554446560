"""Digest for the -DRG_PROFILE2 -DRG_PROFILE3 build: inside the slow-path visits of step_split_kernel. Every lane of a wavefront that
makes a visit adds its ticks, so sums are per (wavefront visit x 64 lanes); the number of lanes that went on to the general handlers
is counted per lane."""
import json
import sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d["counters"]
        lanes = max(c["role_conversions"], 1)
        wave_visits = lanes / 64.0
        rounds = d["config"]["groups_per_gpu"] // 64 * d["config"]["rounds_per_step"] * d["steps"]
        print("wavefront visits %.0f (%.1f%% of the wave-rounds)  per visit: tier 1.5 %.0f ticks, general handlers + rest %.0f ticks | rows that went to the general handlers %d (%.4f%% of rows) | kernel_ms %.4f" % (
            wave_visits, 100.0 * wave_visits / rounds, c["asserts"] / lanes, c["need_host"] / lanes, c["commit_advances"], 100.0 * c["commit_advances"] / max(c["rows"], 1),
            d["roofline"]["avg_kernel_ms"]))
