"""Summarise the rocprofv3 CSVs written by tools/prof.sh: per-kernel stats and per-dispatch counter means."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(out):
    lines = []
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
        for row in csv.DictReader(open(f)):
            lines.append("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
                row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "")
                acc[name][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
            lines.append("== counters (%s)" % os.path.relpath(f, out))
            for name, ctrs in acc.items():
                if "step_kernel" not in name and "step_split_kernel" not in name:
                    continue
                lines.append("  %s" % name[:90])
                for c, vals in sorted(ctrs.items()):
                    lines.append("    %-24s mean/dispatch=%.4g  dispatches=%d" % (c, sum(vals) / len(vals), len(vals)))
    text = "\n".join(lines)
    open(os.path.join(out, "summary.txt"), "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1])
