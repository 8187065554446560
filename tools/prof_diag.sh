#!/bin/bash
# Diagnostic counter passes for the step kernel (GPU box): takes a wish list of SQ / SQC / LDS counters, keeps the ones this
# device offers (rocprofv3 --list-avail), collects them six at a time (--pmc only), prints mean per dispatch of the step kernel.
#   tools/prof_diag.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
WISH="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_EXP_GDS SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACCUM_PREV SQ_WAVES_EQ_64 SQ_WAIT_INST_VMEM SQ_WAIT_INST_VALU GRBM_GUI_ACTIVE"
HAVE=""
for c in $WISH; do grep -qw "$c" $OUT/avail.txt && HAVE="$HAVE $c"; done
echo "available of the wish list:$HAVE" > $OUT/have.txt
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-pcie --steps 4 --warmup 1 $*"
set -- $HAVE
i=0
while [ $# -gt 0 ]; do
  G=""; for k in 1 2 3 4 5 6; do [ $# -gt 0 ] && { G="$G $1"; shift; }; done
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
done
cd $ROOT
python - $OUT <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/p_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "step_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # one row per (dispatch, counter) — or per dimension instance: sum instances per dispatch
    for k, v in sorted(acc.items()):
        print("%-30s sum %.4g over %d rows" % (k, sum(v), len(v)))
PY
