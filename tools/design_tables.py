#!/usr/bin/env python
"""Prints the measurement tables of DESIGN.md section 6 from the files an evidence pass left under profiles/ (tools/prof_r6.sh + tools/collect_r6.sh):
   python tools/design_tables.py r06k
so that every number in the section is a number of a committed file."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def stats(tag, name):
    f = os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, name))
    if not os.path.exists(f):
        return None
    for row in csv.reader(open(f)):
        if row and "step32_kernel" in row[0]:
            return {"kernel": row[0].split("(")[0].replace("void rg::", ""), "calls": int(row[1]), "avg_ns": float(row[3]), "min_ns": int(row[5]), "max_ns": int(row[6])}
    return None


def traffic(cfg, groups):
    for e in json.load(open(os.path.join(P, "traffic.json")))["entries"]:
        if str(e["config"]) == str(cfg) and e["groups_per_gpu"] == groups and e.get("outcome_format") == "rg_outcome32_t":
            return e
    return None


def main(tag):
    rows = [("config 3, 65 536 groups (bench default)", "n1_default", "config3", 3, 65536),
            ("config 4's shard, 131 072 groups (what `--gpus 8` runs per GPU)", "config4_shard_131072", "config4_shard_131072", 4, 131072),
            ("config 5's shard, 131 072 groups (leader churn)", "config5_shard_131072", "config5_shard_131072", 5, 131072),
            ("config 5, 65 536 groups", "config5_65536", "config5_65536", 5, 65536),
            ("config 2, 4 096 groups x 3, leader view", "config2_4096", None, 2, 4096),
            ("config 2's mirrored follower view", "config2f_4096", None, "2f", 4096)]
    print("| configuration | kernel | decisions/s (1 GPU) | ms/launch: HIP events in `bench.py` / rocprofv3 (calls; min – max) | HBM measured per launch -> TB/s, frac of 8 TB/s | layout estimate | `golden` |")
    print("|---|---|---|---|---|---|---|")
    for label, bname, sname, cfg, groups in rows:
        d = json.load(open(os.path.join(P, "%s_bench_%s.json" % (tag, bname))))
        r = d["roofline"]
        st = stats(tag, sname) if sname else None
        tr = traffic(cfg, groups)
        ms = r["avg_kernel_ms"]
        prof = "%.4f (%d; %.4f – %.4f)" % (st["avg_ns"] / 1e6, st["calls"], st["min_ns"] / 1e6, st["max_ns"] / 1e6) if st else "—"
        if tr and tr.get("lib_sha16") == r.get("lib_sha16", tr.get("lib_sha16")):
            tb = tr["traffic_bytes_per_launch"]
            hbm = "%.1f MB -> %.2f, **%.3f**" % (tb / 1e6, tb / (ms * 1e-3) / 1e12, tb / (ms * 1e-3) / 8e12)
        else:
            hbm = "—"
        g = d.get("golden")
        print("| %s | `%s` | %.3g | %.4f / %s | %s | %.1f MB | %s |" % (label, (st or {}).get("kernel", r["kernel"]), d["value"], ms, prof, hbm, r["layout_estimate_bytes"] / 1e6,
                                                                 (g or {}).get("outcomes", "—") if isinstance(g, dict) else "—"))
    d = json.load(open(os.path.join(P, "%s_bench_n1_default.json" % tag)))
    r = d["roofline"]
    print()
    print("default line:", "value %.4g  ms_per_step %.4f  kernel %.4f  frac %.3f  traffic %.1f MB (in run: %s)  copy %.0f GB/s" % (
        d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r["traffic"] * 1e3, r["traffic_measured_in_this_run"], r["measured_copy_gbps"]))
    v = r["valu"]
    print("valu:", "%.4g per launch, %.0f per 64 groups per round, busy %.3f of one wave64 instruction per SIMD per 4 cycles at %.0f MHz; salu %.3g lds %.3g" % (
        v["insts_per_launch"], v["per_64_groups_per_round"], v["busy_frac"], v["clock_mhz"], v["salu_per_launch"], v["lds_per_launch"]))
    pl = r["per_launch"]
    print("per launch:", "min %.4f median %.4f max %.4f mean %.4f over %d" % (pl["min_ms"], pl["median_ms"], pl["max_ms"], pl["mean_ms"], pl["launches"]))
    print("int64 body: %.4f ms, %.3g/s; long-lived %.3g (%.4f ms); adverse %.3g (%.4f ms, need_host_as_expected %s)" % (
        r["ms_int64_body"], r["value_int64_body"], d["value_long_lived_groups"], d["long_lived_groups"]["avg_kernel_ms"], d["value_adverse_mix"],
        d["adverse_mix"]["avg_kernel_ms"], d["adverse_mix"]["need_host_as_expected"]))
    c = d["cpu_baseline"]
    print("cpu: 1 thread %.3g, 3 threads %.3g, 64 threads %.3g; translated reference %.3g" % (c["value_1_thread"], c["value_3_threads"], c["value_64_threads"], c["reference_translated"]["value_1_thread"]))
    pc = d["pcie_inclusive"]
    print("pcie: serial %.3g, async %.3g, packed %.3g; link %.1f GB/s both ways" % (pc["serial_rg_submit"], pc["pipelined_rg_submit_async"], pc["pipelined_rg_submit_async_packed"], pc["link_gbytes_per_s_both_ways"]["packed"]))
    t = d["tick_latency"]
    for way in ("rg_submit_async_packed", "rg_tick_launch"):
        print("tick %s: p50 %.0f p99 %.0f max %.0f over %d" % (way, t[way]["p50_us"], t[way]["p99_us"], t[way]["max_us"], t[way]["ticks"]))
    print("single-round launch %.1f us; resident tick %.1f us; by nodes %s" % (t["device_us_per_single_round_launch"], t["device_us_per_resident_tick"], t.get("device_us_per_resident_tick_by_graph_nodes")))
    f = os.path.join(P, "%s_tick_latency_1000.json" % tag)
    if os.path.exists(f):
        t = json.load(open(f))["tick_latency"]
        for way in ("rg_submit_async_packed", "rg_tick_launch"):
            print("1000 ticks %s: p50 %.0f p99 %.0f p999 %.0f max %.0f over %d" % (way, t[way]["p50_us"], t[way]["p99_us"], t[way]["p999_us"], t[way]["max_us"], t[way]["ticks"]))


if __name__ == "__main__":
    main(sys.argv[1])
