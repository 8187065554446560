#!/bin/bash
# memory-copy + kernel timeline of the host-memory legs of bench.py (is the upload of batch k+1 really overlapping the download of batch k?)
#   tools/prof_pcie.sh <tag> [events to print, default 70]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_pcie_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --pcie-batches 6 > $OUT/run.log 2>&1
cd $ROOT
python - $OUT ${2:-70} <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", "copy "), r.get("Stream_Id", "")))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], r.get("Stream_Id", "")))
ev.sort()
ev = ev[-int(sys.argv[2]):]
t0 = ev[0][0]
for s, e, what, st in ev:
    print("%9.3f .. %9.3f ms  (%6.3f)  stream %-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st, what))
PY
