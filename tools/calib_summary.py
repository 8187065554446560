#!/usr/bin/env python
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (KB) on kernels of KNOWN byte counts (build/membench run under the two --pmc passes by
tools/prof_r3.sh): per (kernel, grid size) the mean counter value per dispatch, the bytes the kernel is known to move, and their ratio.
MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half of a wide (16 B per lane) streaming read on gfx950; other widths and WRITE_SIZE are
uncalibrated — this is the calibration for the step kernel's own access shapes (8-byte + 16-byte row loads, 16-byte row stores).
usage: python tools/calib_summary.py gpurun_out/prof_<tag>_calib"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

GIB = float(1 << 30)


def known(name, grid):
    """(read bytes, write bytes) per dispatch, or None"""
    if name.startswith("copy_") or "copy_" in name.split("(")[0]:
        return GIB, GIB
    if "read_only" in name:
        return GIB, 0.0
    if "write_only" in name:
        return 0.0, GIB
    m = re.search(r"rows<(\d), (true|false), (\d+), (\d+)>", name)
    if m and grid:
        mode, waves = int(m.group(1)), int(m.group(4))
        rows = grid / waves * 64.0                     # grid = groups x waves threads; 64 rounds
        write = rows * (16.0 + 16.0 * 0.5)
        if mode == 1:
            return rows * 24.0, write                  # 8 + 16 B per row
        return None, write                             # wide rows: gathered entry-term loads overlap, no exact read figure
    return None


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            try:
                grid = int(float(row.get("Grid_Size", "0") or 0))
            except ValueError:
                grid = 0
            acc[(row.get("Kernel_Name", ""), grid)][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    print("%-58s %10s %12s %12s %7s %12s %12s %7s" % ("kernel", "grid", "FETCH_KB", "read_KB", "ratio", "WRITE_KB", "write_KB", "ratio"))
    for (name, grid), c in sorted(acc.items()):
        kb = known(name, grid)
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) if c.get("FETCH_SIZE") else None
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if c.get("WRITE_SIZE") else None
        rd = kb[0] / 1024.0 if kb and kb[0] is not None else None
        wr = kb[1] / 1024.0 if kb and kb[1] is not None else None
        fmt = lambda x: "%12.1f" % x if x is not None else "%12s" % "-"   # noqa: E731
        ratio = lambda a, b: "%7.3f" % (a / b) if a and b else "%7s" % "-"   # noqa: E731
        print("%-58s %10d %s %s %s %s %s %s" % (name[:58], grid, fmt(f), fmt(rd), ratio(f, rd), fmt(w), fmt(wr), ratio(w, wr)))


if __name__ == "__main__":
    main(sys.argv[1])
