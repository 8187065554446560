#!/bin/bash
# memory-copy + kernel timeline of the host-memory legs of bench.py (is the upload of batch k+1 really overlapping the download of batch k?)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_pcie_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/run.log 2>&1
cd $ROOT
python - $OUT <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind","?")), int(r.get("Size", r.get("Bytes", 0) or 0))))
rows.sort()
big=[r for r in rows if r[3] > (16<<20)]
t0=big[0][0] if big else 0
for s,e,d,n in big[-40:]:
    print("%9.3f ms .. %9.3f ms  %-28s %7.1f MB  %5.1f GB/s" % ((s-t0)/1e6,(e-t0)/1e6,d,n/1e6,n/max(e-s,1)))
PY
