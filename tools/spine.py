#!/usr/bin/env python3
"""spine.py file.s kernel_mangled -> instruction counts of ONE ROUND of step32_kernel's deciding wavefront, walked along the paths a cluster in
operation takes: from the header of the round loop to its first s_barrier, with the wave-uniform branches (s_cbranch_vccz: {rare block | election block}, inside it
the rare block and the election block, then the general handlers) decided as named and every exec-mask branch (s_cbranch_execz) falling through — its body is executed.
  main     : no rare block, no election row, nothing for the general handlers  (steady replication)
  election : the same with the election block
The static proxy for ticks per round (the deciding wavefront is alone on its SIMD, instructions are what it pays for)."""
import re
import sys


def kernel_lines(path, k):
    L = open(path).read().split('\n')
    start = next(i for i, l in enumerate(L) if l.startswith(k + ':'))
    end = next(i for i in range(start, len(L)) if '.end_amdhsa_kernel' in L[i])
    return L[start:end]


def isinstr(l):
    return l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')


def walk(K, hdr, decisions):
    """decisions: for the successive s_cbranch_vccz / s_cbranch_vccnz met on the path, True = taken"""
    labels = {l.split(':')[0]: i for i, l in enumerate(K) if l.startswith('.LBB')}
    i, n, d, ops = hdr, 0, 0, {}
    while True:
        if n > 20000 or i >= len(K):
            return -1, ops                      # no barrier on this path: the loop's branch structure is not the one this walker knows
        l = K[i]
        if isinstr(l):
            op = l.split()[0]
            n += 1
            ops[op] = ops.get(op, 0) + 1
            if op == 's_barrier':
                return n, ops
            m = re.search(r'(\.LBB\d+_\d+)', l)
            if op in ('s_cbranch_vccz', 's_cbranch_vccnz', 's_cbranch_scc0', 's_cbranch_scc1'):
                taken = decisions[d] if d < len(decisions) else True
                d += 1
                if taken:
                    i = labels[m.group(1)]
                    continue
            elif op == 's_branch':
                i = labels[m.group(1)]
                continue
        i += 1


def main():
    K = kernel_lines(sys.argv[1], sys.argv[2])
    hdr = next(i for i, l in enumerate(K) if 'Loop Header: Depth=1' in l)
    a, opsa = walk(K, hdr, [True, True, True])                   # skip {rare, election}, skip the general handlers
    b, opsb = walk(K, hdr, [False, True, False, True, True])     # enter that branch, skip rare, enter election, skip the general handlers
    # the I/O wavefront's round: the barrier-to-barrier stretches that load a row (two columns) and store up to three (its loop is unrolled by four)
    bars = [i for i, l in enumerate(K) if isinstr(l) and l.split()[0] == 's_barrier']
    ios = []
    for x, y in zip(bars, bars[1:]):
        seg = [l.split()[0] for l in K[x:y] if isinstr(l)]
        if sum(o.startswith('global_load') for o in seg) == 2 and sum(o.startswith('global_store') for o in seg) == 3 and len(seg) < 600:
            ios.append(len(seg))
    ios = ios[:4]                              # (a fifth such stretch belongs to the 64-bit body)
    io = (sum(ios) + len(ios) - 1) // max(len(ios), 1)
    cls = lambda ops, p: sum(v for k, v in ops.items() if k.startswith(p))   # noqa: E731
    print("I/O wavefront: %d instructions per round (mean of the unrolled four: %s)" % (io, ios))
    print("main %d instructions (v_cmp %d, v_cndmask %d, s_and/s_or %d, ds_ %d, v_mov %d); election %d (v_cmp %d, v_cndmask %d, s_and/s_or %d, v_mov %d)"
          % (a, cls(opsa, 'v_cmp'), cls(opsa, 'v_cndmask'), cls(opsa, 's_and') + cls(opsa, 's_or'), cls(opsa, 'ds_'), cls(opsa, 'v_mov'),
             b, cls(opsb, 'v_cmp'), cls(opsb, 'v_cndmask'), cls(opsb, 's_and') + cls(opsb, 's_or'), cls(opsb, 'v_mov')))


if __name__ == '__main__':
    main()
