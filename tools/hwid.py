"""Digest for the -DRG_HWID build of step_split_kernel (bench.py ... | python tools/hwid.py): where the hardware placed the deciding and the I/O
wavefront of each workgroup (HW_REG_HW_ID: wave slot 3:0, SIMD 5:4). Summed over all launches of the run."""
import json
import sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d["counters"]
        wgs = c["replied"] + c["role_conversions"] + c["commit_advances"] + c["asserts"]
        print("workgroups %d | deciding wavefront on SIMD 0/1/2/3: %.1f%% %.1f%% %.1f%% %.1f%% | I/O wavefront on the next SIMD: %.1f%% | same wave slot: %.1f%% | mean slot of the deciding wavefront %.2f | kernel_ms %.4f" % (
            wgs, 100.0 * c["replied"] / wgs, 100.0 * c["role_conversions"] / wgs, 100.0 * c["commit_advances"] / wgs, 100.0 * c["asserts"] / wgs,
            100.0 * c["need_host"] / wgs, 100.0 * c["dropped_stale"] / wgs, c["log_appends"] / wgs, d["roofline"]["avg_kernel_ms"]))
