import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d["counters"]; waves = d["config"]["groups_per_gpu"] // 64
        rounds = d["config"]["rounds_per_step"] * d["steps"]
        f = lambda k: c[k] / waves / rounds
        print("per round per wave cycles(s_memtime ticks): drain-wait %.0f  issue %.0f  decide %.0f   | kernel_ms %.4f" % (
            f("need_host"), f("dropped_stale"), f("log_appends"), d["roofline"]["avg_kernel_ms"]))
