#!/bin/bash
# Experiment (GPU box): how long is a round when a deciding wavefront only carries ONE class of tier-1 code?
#   libraftgpu_noack.so  tier 1 without the ack / client-append blocks, fed an all-follower stream
#   libraftgpu_noae.so   tier 1 without the AppendEntries block, fed an all-leader stream
# against the shipped library on the same two streams and on config 3. Decides whether splitting the deciding wavefront
# by row class (followers' rows / leaders' rows) is worth building. Writes gpurun_out/exp_<tag>_*.json.
TAG=${1:-x}
OUT=gpurun_out
run() { # name lib override
  RG_LIB=$2 python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 ${3:+--override "$3"} > $OUT/exp_${TAG}_$1.json 2> $OUT/exp_${TAG}_$1.err
  python - "$OUT/exp_${TAG}_$1.json" "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%-22s %.4f ms/launch  frac %.3f  %s" % (sys.argv[2], r["avg_kernel_ms"], r["frac"], d["counters"]))
PY
}
L=$(pwd)/rafting_amd
run base_c3        $L/libraftgpu.so        ""
run base_followers $L/libraftgpu.so        "leader_frac=0.0;p_timeout=0.0"
run noack_followers $L/libraftgpu_noack.so "leader_frac=0.0;p_timeout=0.0"
run base_leaders   $L/libraftgpu.so        "leader_frac=1.0;p_higher_term=0.0"
run noae_leaders   $L/libraftgpu_noae.so   "leader_frac=1.0;p_higher_term=0.0"
