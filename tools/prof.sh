#!/bin/bash
# rocprofv3 passes for the step kernel (run on the GPU box, from the repo root).
#   tools/prof.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{trace,pmc1,pmc2,pmc3,pmc4}/... ; counter passes use --pmc only
# (never combined with trace domains other than the kernel trace needed to name dispatches). LIGHT=1: kernel trace + the two HBM byte passes only.
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (only the timed launches of the default leg may run under the counters: the other legs launch kernels of the same name)
QUIET="--no-cpu-baseline --no-pcie --no-int64-pass --no-adverse --index-base-batches 0 --tick-batches 0 --long-launch-rounds 0 --no-pmc"
BENCH="python $ROOT/bench.py $QUIET --steps 4 --warmup 1 $*"
# the trace pass with TRACE_STEPS launches (default 4; the evidence pass of round 5 uses 20 + 4 warm-up: the first launch of a process is slower)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $QUIET --steps ${TRACE_STEPS:-4} --warmup ${TRACE_WARMUP:-1} $* > $OUT/trace.log 2>&1
if [ -z "${LIGHT:-}" ]; then
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -o p -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2 -o p -- $BENCH > $OUT/pmc2.log 2>&1
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o p -- $BENCH > $OUT/pmc4.log 2>&1
[ -z "${LIGHT:-}" ] && rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc5 -o p -- $BENCH > $OUT/pmc5.log 2>&1
cd $ROOT
python tools/prof_summary.py $OUT
