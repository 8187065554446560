#!/bin/bash
# kernel-trace pass with MORE launches per run (24 instead of 5): the first launch of a process is slower (code load), and with five calls it
# moves the average rocprofv3 reports by several per cent. tools/prof_trace_long.sh <tag>  ->  gpurun_out/prof_<tag>_long_<workload>/
set -u
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for spec in "3 65536" "5 65536" "4 131072" "5 131072"; do
  set -- $spec
  OUT=$ROOT/gpurun_out/prof_${TAG}_long_c$1_$2
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-copy-bw --steps 20 --warmup 4 --config $1 --groups-per-gpu $2 > $OUT/trace.log 2>&1
  grep -h "step32" $OUT/trace/*/t_kernel_stats.csv $OUT/trace/t_kernel_stats.csv 2>/dev/null | head -2
  find $OUT -name '*kernel_trace.csv' -size +2M -delete 2>/dev/null; find $OUT -name '*agent_info*' -delete 2>/dev/null
done
cd $ROOT
