#!/usr/bin/env python3
"""spine_trace.py file.s kernel out_prefix -> <out_prefix>_main.s / _election.s: the instructions of tools/spine.py's two paths, in execution order."""
import re
import sys
sys.path.insert(0, __import__('os').path.dirname(__file__))
import spine

K = spine.kernel_lines(sys.argv[1], sys.argv[2])
hdr = next(i for i, l in enumerate(K) if 'Loop Header: Depth=1' in l)
labels = {l.split(':')[0]: i for i, l in enumerate(K) if l.startswith('.LBB')}


def trace(dec):
    i, d, out = hdr, 0, []
    while len(out) < 5000:
        l = K[i]
        if spine.isinstr(l):
            op = l.split()[0]
            out.append(l)
            if op == 's_barrier':
                return out
            m = re.search(r'(\.LBB\d+_\d+)', l)
            if op in ('s_cbranch_vccz', 's_cbranch_vccnz', 's_cbranch_scc0', 's_cbranch_scc1'):
                t = dec[d] if d < len(dec) else True
                d += 1
                out.append('   ; ---- taken' if t else '   ; ---- not taken')
                if t:
                    i = labels[m.group(1)]
                    continue
            elif op == 's_branch':
                i = labels[m.group(1)]
                continue
        i += 1
    return out


open(sys.argv[3] + '_main.s', 'w').write('\n'.join(trace([True, True, True])))
open(sys.argv[3] + '_election.s', 'w').write('\n'.join(trace([False, True, False, True, True])))
