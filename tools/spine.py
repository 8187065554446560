#!/usr/bin/env python3
"""spine.py file.s kernel_mangled -> instruction count of the narrow decide loop from its header to the branch that skips the slow path"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
k = sys.argv[2]
start = next(i for i, l in enumerate(L) if l.startswith(k + ':'))
end = next(i for i in range(start, len(L)) if '.end_amdhsa_kernel' in L[i])
K = L[start:end]
isinstr = lambda l: l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')
hdr = next(i for i, l in enumerate(K) if 'Loop Header: Depth=1' in l)          # first depth-1 loop = narrow decide loop
# the tail block: the first block after hdr that contains two ds_write_b128 followed by s_barrier
labels = {l.split(':')[0]: i for i, l in enumerate(K) if l.startswith('.LBB')}
bar = next(i for i in range(hdr, len(K)) if 's_barrier' in K[i])
# walk back from barrier to its block label
tail = max(i for i in range(hdr, bar) if K[i].startswith('.LBB'))
tail_label = K[tail].split(':')[0]
# branches to a block that (directly or via one hop) reaches tail: accept direct branch to tail_label or to a block that ends with s_branch tail_label
hop = set([tail_label])
for lab, i in labels.items():
    j = i + 1
    n = 0
    while j < len(K) and not K[j].startswith('.LBB') and n < 8:
        if isinstr(K[j]):
            n += 1
            m = re.search(r's_branch\s+(\.LBB\d+_\d+)', K[j])
            if m and m.group(1) == tail_label: hop.add(lab)
        j += 1
br = next(i for i in range(hdr, bar) if re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', K[i]) and re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', K[i]).group(1) in hop)
spine = [l for l in K[hdr:br + 1] if isinstr(l)]
tailn = [l for l in K[tail:bar + 1] if isinstr(l)]
movs = sum(1 for l in spine if l.split()[0].startswith('v_mov'))
lanes = sum(1 for l in spine if 'lane_b32' in l.split()[0])
print("spine %d (+ tail %d) instructions; v_mov %d, read/writelane %d; loop total %d" % (len(spine), len(tailn), movs, lanes, sum(1 for l in K[hdr:bar] if isinstr(l))))
