#!/usr/bin/env python3
"""spine.py file.s kernel_mangled -> instruction counts of ONE ROUND of step32_kernel's deciding wavefront, walked along the paths a cluster in
operation takes: from the header of the round loop to its first s_barrier, with the wave-uniform branches (the ballots: {rare block | election block}, inside it
the rare block and the election block, inside that one block per row class, then the general handlers) decided as named and every exec-mask branch
(s_cbranch_execz) falling through — its body is executed.
  main     : no rare block, no election row, nothing for the general handlers  (steady replication)
  election : the same with the election block and every class block in it; per class on the last line
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


def walk(K, hdr, enters):
    """enters: for the successive wave-uniform ballot branches met on the path (s_cbranch_vccz / vccnz / scc0 / scc1), True = the block behind the
    ballot is entered. The compiler lays a block out either behind a vccz that skips it or at the target of a vccnz that enters it."""
    labels = {l.split(':')[0]: i for i, l in enumerate(K) if l.startswith('.LBB')}
    i, n, d, ops = hdr, 0, 0, {}
    scc_means_zero = True                        # what SCC = 1 says about the ballot last compared: s_cmp_eq_u64 x, 0 -> "no lane", s_cmp_lg_u64 x, 0 -> "some lane"
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
            if op.startswith('s_cmp_eq'):
                scc_means_zero = True
            elif op.startswith('s_cmp_lg') or op.startswith('s_cmp_ne'):
                scc_means_zero = False
            m = re.search(r'(\.LBB\d+_\d+)', l)
            if op in ('s_cbranch_vccz', 's_cbranch_vccnz', 's_cbranch_scc0', 's_cbranch_scc1'):
                enter = enters[d] if d < len(enters) else False
                d += 1
                if op.startswith('s_cbranch_vcc'):
                    taken_means_enter = op.endswith('nz')
                else:                            # (round 5: ballots issued ahead of their branches are tested on the scalar unit: s_cmp + s_cbranch_scc)
                    taken_means_enter = op.endswith('scc1') != scc_means_zero
                if enter == taken_means_enter:
                    i = labels[m.group(1)]
                    continue
            elif op == 's_branch':
                i = labels[m.group(1)]
                continue
        i += 1


def main():
    K = kernel_lines(sys.argv[1], sys.argv[2])
    hdr = next(i for i, l in enumerate(K) if 'Loop Header: Depth=1' in l)
    a, opsa = walk(K, hdr, [])                                   # no {rare, election} block, no general handlers
    # the ballots in program order: {rare | election}, rare, election, then (round 5: one sub-block per row class, each behind its own ballot) vote
    # replies, timeouts, vote requests, the conversion tail; then the general handlers
    b, opsb = walk(K, hdr, [True, False, True, True, True, True, True])       # every class and the conversion tail: the worst round
    c, _ = walk(K, hdr, [True, False, True, True, False, False, False])       # a vote reply that is merely counted: three election rounds of four
    d, _ = walk(K, hdr, [True, False, True, True, False, False, True])        # a vote reply that converts
    e, _ = walk(K, hdr, [True, False, True, False, True, False, True])        # a timeout (always converts or prepares)
    f, _ = walk(K, hdr, [True, False, True, False, False, True, True])        # a vote request that converts
    # the I/O wavefront's round: the barrier-to-barrier stretches that load a row (two columns) and store up to three (its loop is unrolled by four)
    bars = [i for i, l in enumerate(K) if isinstr(l) and l.split()[0] == 's_barrier']
    ios = []
    for x, y in zip(bars, bars[1:]):
        seg = [l.split()[0] for l in K[x:y] if isinstr(l)]
        if sum(o.startswith('global_load') for o in seg) == 2 and sum(o.startswith('global_store') for o in seg) in (2, 3) and len(seg) < 600:      # (two stores: compact outcome rows)
            ios.append(len(seg))
    ios = ios[:4]                              # (a fifth such stretch belongs to the 64-bit body)
    io = (sum(ios) + len(ios) - 1) // max(len(ios), 1)
    cls = lambda ops, p: sum(v for k, v in ops.items() if k.startswith(p))   # noqa: E731
    print("I/O wavefront: %d instructions per round (mean of the unrolled four: %s)" % (io, ios))
    print("main %d instructions (v_cmp %d, v_cndmask %d, s_and/s_or %d, ds_ %d, v_mov %d); election %d (v_cmp %d, v_cndmask %d, s_and/s_or %d, v_mov %d)"
          % (a, cls(opsa, 'v_cmp'), cls(opsa, 'v_cndmask'), cls(opsa, 's_and') + cls(opsa, 's_or'), cls(opsa, 'ds_'), cls(opsa, 'v_mov'),
             b, cls(opsb, 'v_cmp'), cls(opsb, 'v_cndmask'), cls(opsb, 's_and') + cls(opsb, 's_or'), cls(opsb, 'v_mov')))
    print("election rounds by class: vote reply counted %d, vote reply converting %d, timeout %d, vote request converting %d, all classes %d" % (c, d, e, f, b))


if __name__ == '__main__':
    main()
