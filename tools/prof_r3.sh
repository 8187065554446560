#!/bin/bash
# Round-3 evidence pass (run on the GPU box from the repo root): tools/prof_r3.sh <tag>
#   1. default bench line (cpu_baseline + PCIe legs + measured copy)                           -> gpurun_out/bench_<tag>_default.json
#   2. tools/prof.sh <tag>: config 3 (the bench default): kernel trace + SQ passes + FETCH_SIZE / WRITE_SIZE passes
#   3. configs 5 @ 65 536, 4 / 5 shards (131 072), 2 and 2f (4 096): bench line, kernel trace, FETCH_SIZE, WRITE_SIZE (LIGHT=1: no SQ passes)
#   4. build/membench under the FETCH_SIZE / WRITE_SIZE passes: kernels of KNOWN byte counts in the step kernel's own access shapes
#      (8-byte + 16-byte row loads, 16-byte row stores) -> the calibration of the two counters for this access pattern (tools/calib_summary.py)
# Counter passes never share a run with another trace domain. Everything lands under gpurun_out/; tools/collect_r3.sh copies what is judged.
set -u
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_${TAG}_default.json 2> $OUT/bench_${TAG}_default.err
bash tools/prof.sh $TAG > $OUT/prof_$TAG.log 2>&1
for spec in "5 65536" "4 131072" "5 131072" "2 4096" "2f 4096"; do
  set -- $spec
  python bench.py --config $1 --groups-per-gpu $2 --no-cpu-baseline --no-pcie > $OUT/bench_${TAG}_c$1_$2.json 2> $OUT/bench_${TAG}_c$1_$2.err
  LIGHT=1 bash tools/prof.sh ${TAG}_c$1_$2 --config $1 --groups-per-gpu $2 > $OUT/prof_${TAG}_c$1_$2.log 2>&1
done
if [ -x build/membench ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_calib/fetch -o p -- $ROOT/build/membench > $OUT/prof_${TAG}_calib_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_calib/write -o p -- $ROOT/build/membench > $OUT/prof_${TAG}_calib_write.log 2>&1
  cd $ROOT
  python tools/calib_summary.py $OUT/prof_${TAG}_calib > $OUT/prof_${TAG}_calib/summary.txt 2>&1
fi
find $OUT/prof_${TAG}* -name '*agent_info*' -delete 2>/dev/null
find $OUT/prof_${TAG}* -name '*kernel_trace.csv' -size +2M -delete 2>/dev/null
find $OUT/prof_${TAG}* -name '*counter_collection.csv' -size +4M -delete 2>/dev/null
echo done
