#!/bin/bash
# Same-box A/B of two library builds (box-to-box variation of one binary is up to 8 %): tools/exp_ab.sh <libA> <libB> [bench args...]
# alternates A, B, A, B, A, B and prints ms per launch of each run.
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    RG_LIB=$(pwd)/$L python bench.py --no-cpu-baseline --no-pcie --steps ${STEPS:-10} --warmup 2 "$@" | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('%-40s %.4f ms  frac %.3f  %s' % ('$L', r['avg_kernel_ms'], r['frac'], r['kernel']))"
  done
done
