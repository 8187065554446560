"""The oracle against the REFERENCE'S OWN CODE (CPU suite).

oracle/_ref/libref.so holds the reference's Java decision classes — Follower, Candidate, Leader, Leadership.State,
Membership, RaftMember, TimerTicket, RocksLog's log operations, RaftRoutine's role switch and timers, RaftContext's
switchTo / commitLog — translated token by token into C++ by tools/make_ref.py (sha-pinned source ranges, no hand-edited
output) and compiled against stand-ins for the JDK and the I/O plugins (oracle/ref_shim/).  These tests replay the same
inputs through that library and through the hand-written oracle (oracle/raft_oracle.c) and require identical answers:
every known-answer scenario, >= 10^6 random inputs per pure function, the state-aware lockstep fuzzer over seven
cluster shapes, and the BASELINE replay streams.  Skipped only where the library cannot exist (no /root/reference and no
prebuilt copy)."""
import numpy as np
import pytest

from rafting_amd import abi, workload
from tests import fuzz, kat_scenarios, oracle_lib, ref_lib
from tests.helpers import canonical_state, compare_outcomes, compare_states

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libref.so needs the reference checkout to be built")

N_FUNC = 1_000_000


def mk_ref(groups, cluster, self_slot, pre_vote):
    return ref_lib.RefTable(groups, cluster, self_slot, pre_vote)


@pytest.mark.parametrize("scenario", kat_scenarios.SCENARIOS, ids=lambda f: f.__name__)
def test_reference_code_passes_the_known_answer_scenarios(scenario):
    """the hand-derived KATs of tests/kat_scenarios.py, each citing the Java lines it was read from, hold on the
    reference's own code — so they pin the reference, not just the oracle"""
    scenario(mk_ref)


def test_is_better_matches_the_reference():
    """Membership.isBetter (member/Membership.java:74-108)"""
    rng = np.random.default_rng(1)
    n = N_FUNC
    nr, cr = rng.integers(0, 3, n, dtype=np.int32), rng.integers(0, 3, n, dtype=np.int32)
    ct = rng.integers(-3, 12, n, dtype=np.int64)
    nt = ct + rng.choice(np.array([-1, 0, 0, 0, 1, 1 << 40], dtype=np.int64), n)
    nb, cb = rng.integers(-1, 5, n, dtype=np.int32), rng.integers(0, 5, n, dtype=np.int32)   # a live membership has a ballot unless it is a Follower
    cb[(cr == abi.FOLLOWER) & (rng.random(n) < 0.3)] = abi.NO_NODE
    nb[(nr == abi.CANDIDATE) & (nb == abi.NO_NODE)] = 0        # `ballot.equals` on a null ballot is an NPE in the reference: never built
    a, b = np.zeros(n, np.int32), np.zeros(n, np.int32)
    oracle_lib.lib().orc_is_better_batch(n, *(x.ctypes.data for x in (nr, nt, nb, cr, ct, cb, a)))
    ref_lib.lib().ref_is_better_batch(n, *(x.ctypes.data for x in (nr, nt, nb, cr, ct, cb, b)))
    assert np.array_equal(a, b)
    assert {1, 0, -abi.A_LEADER_UNCHANGED, -abi.A_CAND_BALLOT} <= set(np.unique(a).tolist())


@pytest.mark.parametrize("followers", [1, 2, 3, 4, 5, 6])
def test_major_indices_matches_the_reference(followers):
    """Leadership.State.majorIndices (member/Leadership.java:116-130), incl. its golden comment table"""
    rng = np.random.default_rng(followers)
    n = N_FUNC // 4
    m = rng.integers(0, 50, (n, followers), dtype=np.int64)
    m[: n // 8] = rng.integers(-2**62, 2**62, (n // 8, followers), dtype=np.int64)
    a, b = np.zeros((n, 2), np.int64), np.zeros((n, 2), np.int64)
    oracle_lib.lib().orc_major_indices_batch(n, followers, m.ctypes.data, a.ctypes.data)
    ref_lib.lib().ref_major_indices_batch(n, followers, m.ctypes.data, b.ctypes.data)
    assert np.array_equal(a, b)
    s = np.sort(m, axis=1)
    assert np.array_equal(a[:, 0], s[:, 0]) and np.array_equal(a[:, 1], s[:, followers // 2])


def test_update_index_matches_the_reference():
    """Leadership.State.updateIndex (member/Leadership.java:75-114), incl. Math.round(Math.log(Math.E + recentRejection))"""
    rng = np.random.default_rng(7)
    n = N_FUNC
    last_epoch = rng.integers(0, 40, n, dtype=np.int64)
    match = np.where(rng.random(n) < 0.5, 0, rng.integers(0, 200, n, dtype=np.int64))
    nxt = np.where(match > 0, match + 1, rng.integers(1, 400, n, dtype=np.int64))
    st = np.stack([last_epoch, nxt, match], axis=1).astype(np.int64)
    rej = rng.choice(np.array([0, 1, 2, 9, 10, 30, 31, 87, 88, 241, 242, 662, 663, 4912, 4913, 98713, 98714, 2**31 - 1, -1, -2], dtype=np.int32), n)
    rej[: n // 2] = rng.integers(0, 3000, n // 2, dtype=np.int32)
    pend = (rng.random(n) < 0.2).astype(np.uint8)
    epoch = last_epoch + rng.integers(-2, 3, n, dtype=np.int64)
    index = np.where(rng.random(n) < 0.1, match - 1, match + rng.integers(0, 60, n, dtype=np.int64))
    succ = (rng.random(n) < 0.6).astype(np.uint8)
    snap = (rng.random(n) < 0.25).astype(np.uint8)
    sa, ra, pa, ca = st.copy(), rej.copy(), pend.copy(), np.zeros(n, np.int32)
    sb, rb, pb, cb = st.copy(), rej.copy(), pend.copy(), np.zeros(n, np.int32)
    oracle_lib.lib().orc_update_index_batch(n, *(x.ctypes.data for x in (sa, ra, pa, epoch, index, succ, snap, ca)))
    ref_lib.lib().ref_update_index_batch(n, *(x.ctypes.data for x in (sb, rb, pb, epoch, index, succ, snap, cb)))
    for x, y, what in ((ca, cb, "status"), (sa, sb, "state"), (ra, rb, "rejection"), (pa, pb, "pending")):
        assert np.array_equal(x, y), what
    assert np.count_nonzero(ca == abi.A_MATCH_ROLLBACK) > 1000 and np.count_nonzero(pa != pend) > 1000


def test_rejection_step_table_matches_the_reference():
    """the integer threshold table that replaces the path's only float against the reference's own
    Math.round(Math.log(Math.E + r)) at every threshold, both sides, and over a dense range"""
    import math
    rs = list(range(0, 20000))
    k = 2
    while True:
        lo = math.ceil(math.exp(k - 0.5) - math.e)
        if lo > 2**31 - 1:
            break
        rs += [lo - 1, lo, lo + 1]
        k += 1
    rs += [2**31 - 1, 2**31 - 2]
    for r in rs:
        assert oracle_lib.rejection_step(r) == ref_lib.rejection_step(r), r


def _lockstep(groups, cluster, self_slot, pre_vote, rounds, seed):
    st0 = fuzz.random_initial_state(groups, cluster, self_slot, seed)
    orc = oracle_lib.OracleTable(groups, cluster, self_slot, pre_vote)
    ref = ref_lib.RefTable(groups, cluster, self_slot, pre_vote)
    orc.load_state(st0)
    ref.load_state(st0)
    compare_states(canonical_state(ref.read_state()), canonical_state(orc.read_state()), "loaded")
    fz = fuzz.Fuzzer(groups, cluster, self_slot, seed, allow_miss=True)
    hist = np.zeros(256, dtype=np.int64)
    for r in range(rounds):
        b = abi.Batch(1, groups)
        fz.round(orc.read_state(), b, 0)
        oo, orf = orc.submit(b, fill=0xAB), ref.submit(b, fill=0xAB)
        compare_outcomes(orf, oo, "round %d" % r)
        compare_states(canonical_state(ref.read_state()), canonical_state(orc.read_state()), "round %d" % r)
        hist += np.bincount(oo.status, minlength=256)
    return hist


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(3, 0, True, 11), (5, 2, True, 12), (5, 4, False, 13),
                                                             (2, 1, True, 14), (4, 0, False, 15), (7, 3, True, 16),
                                                             (6, 5, True, 17)])
def test_lockstep_fuzz_oracle_vs_reference_code(cluster, self_slot, pre_vote, seed):
    """every outcome row AND the whole group state after every round, over dirty traffic (stale terms, wrong prevLog,
    conflicting entries, fenced and late responses, assertion triggers)"""
    hist = _lockstep(384, cluster, self_slot, pre_vote, 150, seed)
    seen = set(np.flatnonzero(hist).tolist())
    assert {abi.OK, abi.A_TWO_LEADERS, abi.A_COMMIT_ROLLBACK, abi.A_SAME_TERM_LEADER, abi.NPE_MAJOR_NULL,
            abi.DROPPED_STALE_ROLE, abi.NOT_LEADER, abi.BAD_EVENT} <= seen, seen


@pytest.mark.parametrize("number,groups,rounds", [(2, 4096, 24), (3, 8192, 24), (5, 8192, 24), (-3, 4096, 48)])
def test_baseline_replays_oracle_vs_reference_code(number, groups, rounds):
    """BASELINE configs' synthetic RPC streams (the bench workload): outcomes and final state. -3 = config 3 with 1 % of the rows being a new
    leader's AppendEntries that overwrite uncommitted entries (RocksLog.conflict / truncate / append run in the reference's own code)"""
    import dataclasses
    cfg = workload.config(abs(number), groups)
    if number < 0:
        cfg = dataclasses.replace(cfg, p_conflict=0.01)
    gen = workload.ReplayGenerator(cfg)
    st0, b = gen.initial_state(), gen.next_batch(rounds)
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    ref = ref_lib.RefTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc.load_state(st0)
    ref.load_state(st0)
    compare_outcomes(ref.submit(b), orc.submit(b), "config %d" % number)
    compare_states(canonical_state(ref.read_state()), canonical_state(orc.read_state()), "config %d" % number)


@pytest.mark.slow
def test_ten_million_rows_of_config3_oracle_vs_reference_code():
    """>= 10^7 rows of the metric's configuration (65 536 groups x 5, 160 rounds in five launches) through the reference's own
    code and through the oracle: every outcome row and the final state. About two minutes (the translated reference runs 2.5 x 10^5 rows/s)."""
    cfg = workload.config(3, 65536)
    gen = workload.ReplayGenerator(cfg)
    st0 = gen.initial_state()
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    ref = ref_lib.RefTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc.load_state(st0)
    ref.load_state(st0)
    rows = 0
    for k in range(5):
        b = gen.next_batch(32)
        compare_outcomes(ref.submit(b), orc.submit(b), "config 3, launch %d" % k)
        rows += b.rounds * b.count
    compare_states(canonical_state(ref.read_state()), canonical_state(orc.read_state()), "config 3 after %d rows" % rows)
    assert rows >= 10_000_000


def test_send_side_oracle_vs_reference_code():
    """N1: what Leader.replicateLog itself ships (recorded by the RaftService stand-in) against orc_replicate, on the
    states a fuzzed run leaves behind, heartbeat and command paths, with in-flight gating"""
    G, P = 512, 5
    st0 = fuzz.random_initial_state(G, P, 1, 77)
    orc, ref = oracle_lib.OracleTable(G, P, 1, True), ref_lib.RefTable(G, P, 1, True)
    orc.load_state(st0)
    ref.load_state(st0)
    fz = fuzz.Fuzzer(G, P, 1, 77, allow_miss=True)
    rng = np.random.default_rng(5)
    for r in range(40):
        b = abi.Batch(1, G)
        fz.round(orc.read_state(), b, 0)
        compare_outcomes(ref.submit(b), orc.submit(b), "round %d" % r)
        hb = (rng.random(G) < 0.5).astype(np.uint8)
        infl = np.where(rng.random((P - 1, G)) < 0.15, rng.integers(0, 30, (P - 1, G)), 0).astype(np.uint16)
        (ho, so), (hr, sr) = orc.replicate(heartbeat=hb, in_flight=infl), ref.replicate(heartbeat=hb, in_flight=infl)
        assert np.array_equal(ho, hr), "send heads, round %d" % r
        assert np.array_equal(so, sr), "sends, round %d" % r
        compare_states(canonical_state(ref.read_state()), canonical_state(orc.read_state()), "after replicate %d" % r)


# ---- the stand-ins the translated reference stands ON (oracle/ref_shim/jrt.hpp) against Python's exact integers --------------------------
# a mistake in jrt.hpp would be common-mode with the oracle's author. Java's arithmetic is specified exactly (JLS 15.17-15.19:
# two's-complement wrap-around, >>> on the low 6 / 5 bits of the count; Math.round = floor(x + 1/2) saturating, NaN -> 0; Long.compare,
# Long.hashCode, Integer.compareUnsigned and Arrays.sort(long[]) as documented), so each stand-in is held to that specification written
# with Python's unbounded integers — no C arithmetic on the checking side — on 10^6 inputs including every boundary.
def _jrt():
    import ctypes as C
    L = ref_lib.lib()
    vp, u32 = C.c_void_p, C.c_uint32
    L.ref_jrt_long_ops.argtypes = [u32, C.c_int, vp, vp, vp]
    L.ref_jrt_int_ops.argtypes = [u32, C.c_int, vp, vp, vp]
    L.ref_jrt_math_round.argtypes = [u32, vp, vp]
    L.ref_jrt_arrays_sort.argtypes = [vp, u32]
    for f in (L.ref_jrt_long_ops, L.ref_jrt_int_ops, L.ref_jrt_math_round, L.ref_jrt_arrays_sort):
        f.restype = None
    return L


def _wrap64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def _wrap32(v):
    v &= (1 << 32) - 1
    return v - (1 << 32) if v >> 31 else v


def _edge_longs(rng, n):
    edges = [0, 1, -1, 2, -2, (1 << 31) - 1, 1 << 31, -(1 << 31), (1 << 32) - 1, 1 << 32, (1 << 62), (1 << 63) - 1, -(1 << 63), -(1 << 63) + 1,
             (1 << 63) - 2, 0x5555555555555555, -0x5555555555555556]
    vals = [rng.choice(edges) if rng.random() < 0.2 else _wrap64(rng.getrandbits(64) >> rng.choice([0, 0, 8, 33, 50])) * rng.choice([1, -1]) for _ in range(n)]
    return [_wrap64(v) for v in vals]


def test_jrt_long_and_int_arithmetic_is_javas():
    import random
    L = _jrt()
    rng = random.Random(9)
    n = 125000
    a, b = _edge_longs(rng, n), _edge_longs(rng, n)
    A, B = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
    out = np.zeros(n, dtype=np.int64)
    models = {0: lambda x, y: _wrap64(x + y), 1: lambda x, y: _wrap64(x - y), 2: lambda x, y: _wrap64(x * y),
              3: lambda x, y: _wrap64((x & ((1 << 64) - 1)) >> (y & 63)), 4: lambda x, y: (x > y) - (x < y),
              5: lambda x, y: _wrap32((x ^ ((x & ((1 << 64) - 1)) >> 32)) & 0xFFFFFFFF), 6: max, 7: min, 8: lambda x, y: _wrap64(-x),
              9: lambda x, y: _wrap32(_wrap32(x) + _wrap32(y))}
    for op, f in models.items():
        L.ref_jrt_long_ops(n, op, A.ctypes.data, B.ctypes.data, out.ctypes.data)
        want = [f(x, y) for x, y in zip(a, b)]
        bad = [i for i in range(n) if int(out[i]) != want[i]]
        assert not bad, (op, a[bad[0]], b[bad[0]], int(out[bad[0]]), want[bad[0]])
    ia = np.array([_wrap32(v) for v in a], dtype=np.int32)
    ib = np.array([_wrap32(v) for v in b], dtype=np.int32)
    io = np.zeros(n, dtype=np.int32)
    imodels = {0: lambda x, y: ((x & 0xFFFFFFFF) > (y & 0xFFFFFFFF)) - ((x & 0xFFFFFFFF) < (y & 0xFFFFFFFF)),
               1: lambda x, y: _wrap32((x & 0xFFFFFFFF) >> (y & 31)), 2: lambda x, y: _wrap32(x + y)}
    for op, f in imodels.items():
        L.ref_jrt_int_ops(n, op, ia.ctypes.data, ib.ctypes.data, io.ctypes.data)
        want = [f(int(x), int(y)) for x, y in zip(ia, ib)]
        bad = [i for i in range(n) if int(io[i]) != want[i]]
        assert not bad, (op, int(ia[bad[0]]), int(ib[bad[0]]), int(io[bad[0]]), want[bad[0]])


def test_jrt_math_round_is_javas():
    """Math.round(double) of Java 7+ = floor(x + 1/2) computed EXACTLY, NaN -> 0, saturating (java.lang.Math; Java 6 added in double
    arithmetic and got 0.49999999999999994 and the odd integers of [2^52, 2^53) wrong — so did jrt.hpp until this test existed; the
    reference's one call site, Leadership.java:105, cannot produce such an argument). Checked with fractions.Fraction."""
    import random
    from fractions import Fraction
    import math
    import struct
    rng = random.Random(10)
    xs = [0.0, -0.0, 0.5, -0.5, 1.5, -1.5, 2.5, 0.49999999999999994, -0.49999999999999994, 4503599627370495.5, 4503599627370496.5, 9007199254740993.0,
          9.223372036854775e18, 9.223372036854776e18, -9.223372036854776e18, -9.3e18, 1e300, -1e300, float("inf"), float("-inf"), float("nan")]
    while len(xs) < 250000:
        k = rng.random()
        if k < 0.4:
            xs.append(math.log(math.e + rng.randrange(0, 1 << 31)))                      # the one call site: Leadership.java:105
        elif k < 0.6:
            xs.append(rng.randrange(-(1 << 40), 1 << 40) + rng.choice([0.5, -0.5, 0.25, 0.49999999999999994, 0.5000000000000001]))
        else:
            xs.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0])    # any bit pattern
    X = np.array(xs, dtype=np.float64)
    out = np.zeros(len(xs), dtype=np.int64)
    _jrt().ref_jrt_math_round(len(xs), X.ctypes.data, out.ctypes.data)
    lo, hi = -(1 << 63), (1 << 63) - 1

    def model(x):                        # Java 7+: "the long closest to the argument, ties rounding to positive infinity", saturating
        if x != x:
            return 0
        if x in (float("inf"), float("-inf")):
            return hi if x > 0 else lo
        return max(lo, min(hi, math.floor(Fraction(x) + Fraction(1, 2))))         # exact rational arithmetic
    for i, x in enumerate(xs):
        assert int(out[i]) == model(x), (x, int(out[i]), model(x))


def test_jrt_arrays_sort_is_an_ascending_permutation():
    import random
    rng = random.Random(11)
    for _ in range(2000):
        n = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 50, 500])
        v = _edge_longs(rng, n)
        a = np.array(v, dtype=np.int64)
        _jrt().ref_jrt_arrays_sort(a.ctypes.data, n)
        assert a.tolist() == sorted(v)
