#!/usr/bin/env python3
"""spine.py file.s kernel_mangled -> instruction count of step32_kernel's deciding-wavefront round from the loop header to the branch that
skips the general handlers ("spine": tier 1 with its election block and its rare blocks in line). The loop is the first depth-1 loop of the
kernel; its first s_cbranch_vccz skips the election block, its second the general handlers (tools/spine.sh prints what it finds — read the
number as a static proxy, and check the two branches still mean that after a change of the round loop)."""
import re
import sys

L = open(sys.argv[1]).read().split('\n')
k = sys.argv[2]
start = next(i for i, l in enumerate(L) if l.startswith(k + ':'))
end = next(i for i in range(start, len(L)) if '.end_amdhsa_kernel' in L[i])
K = L[start:end]
isinstr = lambda l: l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')   # noqa: E731
hdr = next(i for i, l in enumerate(K) if 'Loop Header: Depth=1' in l)
vccz = [i for i in range(hdr, len(K)) if re.search(r's_cbranch_vccz\s+\.LBB', K[i])][:2]
bar = next(i for i in range(hdr, len(K)) if 's_barrier' in K[i])
loop_end = next((i for i in range(hdr + 1, len(K)) if 'Loop Header: Depth=1' in K[i]), len(K))
spine = [l for l in K[hdr:vccz[1] + 1] if isinstr(l)]
election = [l for l in K[vccz[0] + 1:vccz[1] + 1] if isinstr(l)]
target = re.search(r'(\.LBB\d+_\d+)', K[vccz[1]]).group(1)
t0 = next(i for i, l in enumerate(K) if l.startswith(target + ':'))
tail = []
for l in K[t0 + 1:]:
    if l.startswith('.LBB'):
        break
    if isinstr(l):
        tail.append(l)
movs = sum(1 for l in spine if l.split()[0].startswith('v_mov'))
lanes = sum(1 for l in spine if 'lane_b32' in l.split()[0])
print("spine %d instructions (election block and what follows it: %d; + %d in the block the skip lands in); v_mov %d, read/writelane %d; "
      "first s_barrier %d instructions after the loop header in layout order; loop total %d"
      % (len(spine), len(election), len(tail), movs, lanes, sum(1 for l in K[hdr:bar] if isinstr(l)), sum(1 for l in K[hdr:loop_end] if isinstr(l))))
