#!/bin/bash
# Round-6 evidence pass (round 5's, plus: the default line now carries its own counter passes, per-launch spread and the device-resident tick) (run on the GPU box from the repo root): tools/prof_r6.sh <tag>
#   1. the default bench line as the driver runs it (every leg: cpu_baseline, PCIe, adverse mix, long-lived groups, tick latency)  -> gpurun_out/bench_<tag>_default.json
#   2. config 3 (the bench default, 65 536 groups), config 4's and config 5's shard (131 072), config 5 @ 65 536: per workload a bench line, ONE kernel-trace
#      pass of 24 launches (--stats), a FETCH_SIZE pass and a WRITE_SIZE pass (separate runs, --pmc only); config 3 also the two SQ passes
#   3. config 2 / 2f (4 096 groups): bench lines only
#   4. build/membench under the FETCH_SIZE / WRITE_SIZE passes: the calibration of the two counters on kernels of known byte counts
# Everything lands under gpurun_out/; tools/collect_r6.sh copies what is judged into profiles/.
set -u
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_${TAG}_default.json 2> $OUT/bench_${TAG}_default.err
QUIET="--no-cpu-baseline --no-pcie --no-adverse --index-base-batches 0 --tick-batches 0 --long-launch-rounds 0 --no-pmc"
TRACE_STEPS=20 TRACE_WARMUP=4 bash tools/prof.sh $TAG > $OUT/prof_$TAG.log 2>&1
for spec in "4 131072" "5 131072" "5 65536"; do
  set -- $spec
  python bench.py --config $1 --groups-per-gpu $2 $QUIET --steps 20 --warmup 4 > $OUT/bench_${TAG}_c$1_$2.json 2> $OUT/bench_${TAG}_c$1_$2.err
  TRACE_STEPS=20 TRACE_WARMUP=4 LIGHT=1 bash tools/prof.sh ${TAG}_c$1_$2 --config $1 --groups-per-gpu $2 > $OUT/prof_${TAG}_c$1_$2.log 2>&1
done
for spec in "2 4096" "2f 4096"; do
  set -- $spec
  python bench.py --config $1 --groups-per-gpu $2 $QUIET --steps 20 --warmup 4 > $OUT/bench_${TAG}_c$1_$2.json 2> $OUT/bench_${TAG}_c$1_$2.err
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
