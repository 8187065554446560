"""Digest for the -DRG_PROFILE experiment build (RG_LIB=build/libraftgpu_prof.so python bench.py ... | python tools/cyc.py [tiers]).
The build reports s_memtime ticks through three of the counter slots: (drain-wait, issue, decide) per round, or with
-DRG_PROFILE_TIERS and the `tiers` argument (tier 1, tier 2, epilogue) — the split of "decide"."""
import json
import sys

names = ("tier-1", "tier-2", "epilogue") if "tiers" in sys.argv[1:] else ("drain-wait", "issue", "decide")
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d["counters"]; waves = d["config"]["groups_per_gpu"] // 64
        rounds = d["config"]["rounds_per_step"] * d["steps"]
        f = lambda k: c[k] / waves / rounds
        print("per round per wave, s_memtime ticks: %s %.0f  %s %.0f  %s %.0f   | kernel_ms %.4f" % (
            names[0], f("need_host"), names[1], f("dropped_stale"), names[2], f("log_appends"), d["roofline"]["avg_kernel_ms"]))
