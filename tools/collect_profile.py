#!/usr/bin/env python
"""Copy what is to be judged from a gpurun_out/prof_<tag>* directory into profiles/ (tracked) under the name <name>:
summary, kernel stats, the traced bench line, and — when present — the traffic entry, merged into profiles/traffic.json
(one entry per workload x kernel; an entry is replaced by a newer one for the same key).
usage: python tools/collect_profile.py gpurun_out/prof_r02c r02c_config3"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src, name):
    dst = os.path.join(ROOT, "profiles")
    if os.path.exists(os.path.join(src, "summary.txt")):
        shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, name + "_rocprofv3_summary.txt"))
    for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, name + "_kernel_stats.csv"))
        break
    for log in ("trace.log",):
        pth = os.path.join(src, log)
        if os.path.exists(pth):
            for ln in open(pth):
                if ln.startswith("{"):
                    open(os.path.join(dst, name + "_bench_traced.json"), "w").write(ln)
    ent = os.path.join(src, "traffic_entry.json")
    if os.path.exists(ent):
        e = json.load(open(ent))
        e["source"] = "profiles/%s_rocprofv3_summary.txt" % name
        tpath = os.path.join(dst, "traffic.json")
        doc = json.load(open(tpath)) if os.path.exists(tpath) else {"note": "HBM bytes per launch of the step kernel from rocprofv3 PMC passes "
                                                                           "(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KB); bench.py quotes an entry only for the same workload, "
                                                                           "kernel and library build (lib_sha16)", "entries": []}
        key = lambda x: (x["config"], x["groups_per_gpu"], x["rounds"], x["kernel"], x.get("outcome_format", "rg_outcome_t"))   # noqa: E731
        doc["entries"] = [x for x in doc["entries"] if key(x) != key(e)] + [e]
        json.dump(doc, open(tpath, "w"), indent=1)
        print("traffic.json:", key(e), "%.1f MB" % (e["traffic_bytes_per_launch"] / 1e6))
    vent = os.path.join(src, "valu_entry.json")
    if os.path.exists(vent):
        e = json.load(open(vent))
        e["source"] = "profiles/%s_rocprofv3_summary.txt" % name
        vpath = os.path.join(dst, "valu.json")
        doc = json.load(open(vpath)) if os.path.exists(vpath) else {"note": "vector (SQ_INSTS_VALU), scalar and LDS instructions per launch of the step kernel from rocprofv3 PMC passes; bench.py "
                                                                           "quotes an entry in roofline.valu only for the same workload, kernel and library build, and only when its own counter pass did not run", "entries": []}
        key = lambda x: (x["config"], x["groups_per_gpu"], x["rounds"], x["kernel"], x.get("outcome_format", "rg_outcome_t"))   # noqa: E731
        doc["entries"] = [x for x in doc["entries"] if key(x) != key(e)] + [e]
        json.dump(doc, open(vpath, "w"), indent=1)
        print("valu.json:", key(e), "%.4g VALU" % e["sq_insts_valu_per_launch"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
