#!/bin/bash
# Static proxy for the step kernel's per-round cost: compiles rg_kernels.hip to gfx950 assembly and reports, for
# step_kernel<4, dense, 64 lanes>, the instruction count and the SGPR-spill traffic inside the round loop.
set -e
OUT=${1:-/tmp/isa}; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I$(dirname $0)/../include ${EXTRA} -S --cuda-device-only -o $OUT/rg.s $(dirname $0)/../rafting_amd/csrc/rg_kernels.hip 2>/dev/null
KERNEL=${KERNEL:-_ZN2rg11step_kernelILi4ELb0ELi64EEEvNS_10StepParamsE}     # KERNEL=_ZN2rg17step_split_kernelILi4ELb0EEEvNS_10StepParamsE for the split kernel
awk -v k="^$KERNEL:" '$0 ~ k {p=1} p{print} /\.end_amdhsa_kernel/{if(p){exit}}' $OUT/rg.s > $OUT/k4.s
python3 - $OUT/k4.s <<'PY'
import sys,re
from collections import Counter
L=open(sys.argv[1]).read().split('\n')
labels={l.split(':')[0]:i for i,l in enumerate(L) if l.startswith('.LBB')}
# the round loop = the widest backward branch; blocks placed after it that jump back inside are part of it
spans=[(i-labels[t],labels[t],i) for i,l in enumerate(L) for t in re.findall(r's_c?branch\S*\s+(\.LBB\d+_\d+)',l) if t in labels and labels[t]<i]
_,first,last=max(spans)
end=last
for i,l in enumerate(L):
    if i>end:
        for t in re.findall(r's_c?branch\S*\s+(\.LBB\d+_\d+)',l):
            if t in labels and first<=labels[t]<=last: end=i
body=[l for l in L[first:end+1] if l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')]
ops=[l.split()[0] for l in body]
c=Counter(ops)
print("loop instructions %d  readlane %d writelane %d s_nop %d  cndmask %d  branches %d" % (len(ops), c['v_readlane_b32'], c['v_writelane_b32'], c['s_nop'], sum(v for k,v in c.items() if k.startswith('v_cndmask')), sum(v for k,v in c.items() if k.startswith('s_cbranch'))))
txt='\n'.join(L)
for k in ('.amdhsa_next_free_sgpr','.amdhsa_next_free_vgpr','.amdhsa_private_segment_fixed_size','.amdhsa_group_segment_fixed_size'):
    m=re.search(re.escape(k)+r'\s+(\d+)',txt)
    if m: print(k, m.group(1))
open(sys.argv[1].replace('k4.s','loop.s'),'w').write('\n'.join(L[first:end+1]))
PY
