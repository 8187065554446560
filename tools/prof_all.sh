#!/bin/bash
# Round-level evidence pass (run on the GPU box from the repo root): tools/prof_all.sh <tag>
#   1. tools/prof.sh <tag>           config 3 default command: kernel trace + PMC passes
#   2. the same (bench line, kernel trace, PMC passes) for config 2, config 4's shard (131 072 groups/GPU), config 5 at 131 072 and 65 536
#   3. N1 replicate kernel: bench line, kernel trace, FETCH/WRITE passes
# Everything lands under gpurun_out/prof_<tag>*/ ; copy what is to be judged into profiles/.
set -u
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
bash tools/prof.sh $TAG > $OUT/prof_$TAG.log 2>&1
python bench.py --copy-bw > $OUT/bench_${TAG}_c3.json 2> $OUT/bench_${TAG}_c3.err
for spec in "2 4096" "4 131072" "5 131072" "5 65536"; do
  set -- $spec
  python bench.py --config $1 --groups-per-gpu $2 --no-cpu-baseline --no-pcie > $OUT/bench_${TAG}_c$1_$2.json 2> $OUT/bench_${TAG}_c$1_$2.err
  bash tools/prof.sh ${TAG}_c$1_$2 --config $1 --groups-per-gpu $2 > $OUT/prof_${TAG}_c$1_$2.log 2>&1      # kernel trace + the PMC passes (HBM bytes) for this workload too
done
python tools/bench_replicate.py 1048576 50 > $OUT/bench_${TAG}_repl.json 2> $OUT/bench_${TAG}_repl.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_repl/trace -o t -- python $ROOT/tools/bench_replicate.py 1048576 20 > $OUT/prof_${TAG}_repl.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_repl/pmc3 -o p -- python $ROOT/tools/bench_replicate.py 1048576 20 >> $OUT/prof_${TAG}_repl.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_repl/pmc4 -o p -- python $ROOT/tools/bench_replicate.py 1048576 20 >> $OUT/prof_${TAG}_repl.log 2>&1
cd $ROOT
find $OUT/prof_${TAG}* -name '*agent_info*' -delete 2>/dev/null
find $OUT/prof_${TAG}* -name '*kernel_trace.csv' -size +2M -delete 2>/dev/null
echo done
