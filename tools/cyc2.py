"""Digest for the -DRG_PROFILE2 experiment build of step_split_kernel (RG_LIB=rafting_amd/libraftgpu_prof2.so python bench.py ... | python tools/cyc2.py):
s_memtime ticks per round per workgroup — deciding wavefront: LDS-read wait, tier 1, tier 2, outcome publish, barrier wait; I/O wavefront: work, barrier wait."""
import json
import sys

for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d["counters"]; wgs = d["config"]["groups_per_gpu"] // 64
        n = wgs * d["config"]["rounds_per_step"] * d["steps"]
        f = lambda k: c[k] / n   # noqa: E731
        print("decide wave: read-wait %.0f  tier1 %.0f  tier2 %.0f  publish %.0f  barrier %.0f | io wave: work %.0f  barrier-wait %.0f | kernel_ms %.4f" % (
            f("role_conversions"), f("commit_advances"), f("asserts"), f("need_host"), f("dropped_stale"), f("replied"), f("log_appends"),
            d["roofline"]["avg_kernel_ms"]))
