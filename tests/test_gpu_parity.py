"""Parity of the HIP decision path (through the C-ABI, libraftgpu.so) against the CPU oracle.
Everything here needs a real MI355X: run with `pytest -m gpu`.

Bar: bit-exact — every reply row, every conditional effect row that is flagged valid, and the final
state of every group (term, votedFor, role, commitIndex, matchIndex[], log tail ...)."""
import os

import numpy as np
import pytest

from rafting_amd import abi, engine
from tests import fuzz, kat_scenarios, oracle_lib
from tests.helpers import set_group, compare_outcomes, compare_states, make_state, simple_log, check_out32_rows

pytestmark = pytest.mark.gpu


def route_through_compact(monkeypatch, out32=False):
    """Table.submit packs every batch that can travel as compact rows (no hint column, every value in [0, 2^31)) with the library's
    rg_batch32_pack and hands it to rg_submit32 — so whatever a test does through submit() is decided by step32_kernel. out32: dense batches
    go through rg_submit32c instead (compact OUTCOME rows, ABI 4) and come back through rg_outcome32_unpack, with the raw rows held to what
    include/raftgpu.h says about them (helpers.check_out32_rows)."""
    wide_submit = engine.Table.submit

    def submit(self, batch, out=None, fill=0):
        if batch.hint is None and abi.batch_fits_32(batch) and self.cluster <= abi.MAX_COMPACT_CLUSTER:      # (larger clusters: wide rows only, include/raftgpu.h)
            if out32 and batch.gid is None:
                before = self.read_state()
                raw = self.submit32c(batch, fill=fill)
                got, _ = engine.unpack32(raw, batch.rounds, batch.count, before.role_epoch)
                check_out32_rows(raw, got, before, self.read_state(), batch.rounds, batch.count)
                if out is not None:
                    out.reply[:], out.logfx[:], out.persist[:] = got.reply, got.logfx, got.persist
                    return out
                return got
            return self.submit32(batch, out, fill)
        return wide_submit(self, batch, out, fill)
    monkeypatch.setattr(engine.Table, "submit", submit)


@pytest.fixture(autouse=True, params=["split", "single", "compact", "compact-forced-wide", "compact-out32", "compact-out32-forced-wide"])
def step_kernel_variant(request, monkeypatch):
    """Every test of this module runs against every step kernel: `split` = decide + I/O wavefront per 64 groups on wide rows (what the
    library picks up to one wavefront of groups per SIMD), `single` = one wavefront does both (picked beyond that), `compact` = the
    compact-format kernel (32-bit body wherever the values allow it), `compact-forced-wide` = the same kernel made to take its 64-bit
    body from the start (RG_FORCE_WIDE=1), `compact-out32*` = the same two with compact outcome rows (rg_submit32c)."""
    monkeypatch.setenv("RG_SPLIT", "0" if request.param == "single" else "1")
    if request.param.startswith("compact"):
        route_through_compact(monkeypatch, out32="out32" in request.param)
        if request.param.endswith("forced-wide"):
            monkeypatch.setenv("RG_FORCE_WIDE", "1")
    return request.param


def mk_gpu(groups, cluster, self_slot, pre_vote):
    return engine.Table(groups, cluster, self_slot, pre_vote)


@pytest.mark.parametrize("scenario", kat_scenarios.SCENARIOS, ids=lambda f: f.__name__)
def test_kat_on_gpu(scenario):
    """The same hand-derived known answers that pin the oracle, asked of the HIP path."""
    scenario(mk_gpu)


def _subset(b, rows, gids):
    """sparse single-round batch holding `rows` of dense batch b"""
    s = abi.Batch(1, len(rows), gid=gids, hints=True)
    for k, row in enumerate(rows):
        hdr, aux = int(b.head["hdr"][row]), int(b.head["aux"][row])
        n = hdr >> 12
        ents = None
        if (hdr & 0xF) == abi.EV_AE_REQ and n:
            ents = b.entry_terms[aux:aux + n]
        s.put(0, k, hdr & 0xF, slot=(hdr >> 4) & 0xF, flag=(hdr >> 8) & 1, a=int(b.ab["x"][row]), b=int(b.ab["y"][row]),
              c=int(b.cd["x"][row]), d=int(b.cd["y"][row]), aux=aux, entries=ents, n=n)
    return s


def _resolve_need_host(gpu, orc, b, out, state_before):
    """The host half of the NEED_HOST protocol: look the missing terms up in the host's log (the oracle's
    lossless log stands in for RocksLog here) and resubmit those rows, sparse, with hints."""
    rows = np.flatnonzero(out.status == abi.NEED_HOST)
    if len(rows) == 0:
        return 0
    gids = rows.astype(np.uint32)
    s = _subset(b, rows, gids)
    for k, row in enumerate(rows):
        g = int(row)
        kind = int(b.head["hdr"][row]) & 0xF
        if kind == abi.EV_AE_REQ:
            prev = int(b.ab["y"][row])
            n = int(b.head["hdr"][row]) >> 12
            aux = int(b.head["aux"][row])
            terms = b.entry_terms[aux:aux + n]
            eidx = int(state_before.epoch_index[g])
            e0 = prev + 1
            if n and e0 <= eidx:
                skip = min(n, eidx - e0 + 1)
                terms, e0 = terms[skip:], e0 + skip
            pt = orc.log_term(g, prev)
            s.set_hint(k, -1 if pt is None else pt, orc.log_conflict(g, e0, terms) if len(terms) else 0)
        else:
            idx = int(out.logfx["log_from"][row])
            t = orc.log_term(g, idx)
            s.set_hint(k, idx, -1 if t is None else t)
    o2 = gpu.submit(s, fill=0xAB)
    assert not np.any(o2.status == abi.NEED_HOST), "hinted rows must apply"
    out.reply[rows], out.logfx[rows], out.persist[rows] = o2.reply, o2.logfx, o2.persist
    return len(rows)


def _lockstep(groups, cluster, self_slot, pre_vote, rounds, seed, allow_miss):
    st0 = fuzz.random_initial_state(groups, cluster, self_slot, seed)
    gpu = engine.Table(groups, cluster, self_slot, pre_vote)
    orc = oracle_lib.OracleTable(groups, cluster, self_slot, pre_vote)
    gpu.load_state(st0)
    orc.load_state(st0)
    fz = fuzz.Fuzzer(groups, cluster, self_slot, seed, allow_miss=allow_miss)
    batches, outs, hist, misses = [], [], np.zeros(256, dtype=np.int64), 0
    for r in range(rounds):
        cur = gpu.read_state()
        b = abi.Batch(1, groups)
        fz.round(cur, b, 0)
        og = gpu.submit(b, fill=0xAB)
        misses += _resolve_need_host(gpu, orc, b, og, cur)
        oo = orc.submit(b, fill=0xAB)
        compare_outcomes(oo, og, "round %d" % r)
        hist += np.bincount(oo.status, minlength=256)
        batches.append(b)
        outs.append(oo)
    compare_states(orc.read_state(), gpu.read_state(), "final")
    return st0, batches, outs, hist, misses, gpu


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(3, 0, True, 11), (5, 2, True, 12), (5, 4, False, 13),
                                                             (2, 1, True, 14), (4, 0, False, 15), (7, 3, True, 16),
                                                             (6, 5, True, 17)])
def test_fuzz_lockstep_with_hints(cluster, self_slot, pre_vote, seed):
    """Round-by-round differential replay; cache misses are resolved through the hint protocol."""
    _, _, _, hist, misses, gpu = _lockstep(384, cluster, self_slot, pre_vote, 120, seed, allow_miss=True)
    seen = set(np.flatnonzero(hist).tolist())
    assert {abi.OK, abi.A_COMMIT_ROLLBACK, abi.DROPPED_STALE_ROLE, abi.NOT_LEADER} <= seen
    c = gpu.counters()
    assert c[0] > 0 and c[1] > 0 and c[2] > 0


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(9, 4, True, 61), (11, 10, False, 62), (15, 0, True, 63), (8, 7, True, 64)])
def test_fuzz_lockstep_on_clusters_above_seven_nodes(step_kernel_variant, cluster, self_slot, pre_vote, seed):
    """ABI 5: Leadership.majorIndices sorts any number of followers (member/Leadership.java:116-130) and so does the table — clusters of
    8 .. 15 nodes on the wide-row kernels, the same lockstep differential as the small ones (hints included); the compact formats say no with a message."""
    if step_kernel_variant not in ("split", "single"):
        pytest.skip("clusters above %d nodes are decided by the wide-row kernels" % abi.MAX_COMPACT_CLUSTER)
    _, _, _, hist, misses, gpu = _lockstep(256, cluster, self_slot, pre_vote, 90, seed, allow_miss=True)
    seen = set(np.flatnonzero(hist).tolist())
    assert {abi.OK, abi.DROPPED_STALE_ROLE, abi.NOT_LEADER} <= seen
    c = gpu.counters()
    assert c[0] > 0 and c[1] > 0 and c[2] > 0 and c[3] > 0
    b = abi.Batch(1, 256)
    with pytest.raises(engine.EngineError, match="wide rows"):
        gpu.submit32(b)
    with pytest.raises(engine.EngineError, match="wide rows"):
        gpu.submit32c(b)
    head, send = gpu.replicate()                          # the send side and the readiness gate take every cluster size
    assert send.shape == (256, cluster - 1) and gpu.ready(1, 0, 0).shape == (256,)


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(5, 0, True, 41), (3, 2, False, 42)])
def test_fuzz_lockstep_general_handlers_only(monkeypatch, cluster, self_slot, pre_vote, seed):
    """Same differential replay with RG_FAST=0: with the fast-path tier on, most rows never reach the general
    handlers, so they get their own dirty-traffic coverage here."""
    monkeypatch.setenv("RG_FAST", "0")
    _, _, _, hist, _, _ = _lockstep(384, cluster, self_slot, pre_vote, 100, seed, allow_miss=True)
    assert hist[abi.OK] > 0 and hist[abi.DROPPED_STALE_ROLE] > 0


def test_fuzz_lockstep_large():
    """a longer run on the most common shape (5 peers): more rare-path interleavings through both tiers"""
    _, _, _, hist, misses, _ = _lockstep(4096, 5, 3, True, 160, 43, allow_miss=True)
    seen = set(np.flatnonzero(hist).tolist())
    assert {abi.OK, abi.A_TWO_LEADERS, abi.A_COMMIT_ROLLBACK, abi.NPE_MAJOR_NULL, abi.DROPPED_STALE_ROLE, abi.NOT_LEADER,
            abi.BAD_EVENT} <= seen, seen


def test_fuzz_multi_round_launch():
    """The same replay as ONE multi-round launch on HBM-resident buffers must equal the row-by-row result."""
    G, P = 1024, 5
    st0, batches, outs, _, misses, _ = _lockstep(G, P, 1, True, 64, 21, allow_miss=False)
    assert misses == 0
    big = fuzz.concat_batches(batches)
    ref = fuzz.concat_outcomes(outs)
    gpu = engine.Table(G, P, 1, True)
    gpu.load_state(st0)
    db = engine.DeviceBatch(gpu, big)
    gpu.timing_enable(True)
    gpu.submit_device(db)
    gpu.sync()
    n, ms = gpu.timing_read()
    assert n == 1 and ms > 0
    compare_outcomes(ref, db.outcome(), "multi-round")
    orc = oracle_lib.OracleTable(G, P, 1, True)
    orc.load_state(st0)
    orc.submit(big)
    compare_states(orc.read_state(), gpu.read_state(), "multi-round final")
    c = gpu.counters()
    kinds = big.head["hdr"] & 0xF
    assert c[0] == int(np.count_nonzero(kinds))
    assert c[1] == int(np.count_nonzero(ref.reply["flags"] & abi.F_REPLIED))
    assert c[2] == int(np.count_nonzero(ref.reply["flags"] & abi.F_ROLE_CHANGED))
    assert c[3] == int(np.count_nonzero(ref.reply["flags"] & abi.F_COMMIT))
    db.free()


def test_need_host_blocks_later_rounds():
    """A cache miss inside a multi-round launch leaves the row unapplied and skips the group's later rows."""
    gpu = engine.Table(2, 3, 0, True)
    st = abi.GroupState(2, 3, runs_total=12)
    for g in range(2):
        st.role[g], st.current_term[g] = abi.FOLLOWER, 9
        st.run_count[g], st.run_offset[g] = 6, 6 * g
        st.first_index[g], st.last_index[g] = 1, 60
        for k in range(6):
            st.run_start[6 * g + k], st.run_term[6 * g + k] = 1 + 10 * k, 1 + k
    gpu.load_state(st)
    got = gpu.read_state()
    assert list(got.run_count) == [4, 4] and int(got.run_start[0]) == 21      # newest 4 runs cached
    b = abi.Batch(3, 2)
    b.put(0, 0, abi.EV_AE_REQ, slot=1, a=9, b=15, c=2, d=0)                   # prev in a forgotten run -> miss
    b.put(1, 0, abi.EV_AE_REQ, slot=1, a=9, b=60, c=6, d=0)
    b.put(0, 1, abi.EV_AE_REQ, slot=1, a=9, b=25, c=3, d=0)                   # cached: fine
    b.put(2, 1, abi.EV_AE_REQ, slot=1, a=9, b=60, c=6, d=30)
    out = gpu.submit(b)
    stt = out.status.reshape(3, 2)
    assert stt[0, 0] == abi.NEED_HOST and int(out.logfx["log_from"][0]) == 15
    assert stt[1, 0] == abi.SKIPPED_AFTER_NEED_HOST and stt[2, 0] == abi.OK   # NONE rows stay OK
    assert stt[0, 1] == abi.OK and stt[2, 1] == abi.OK and out.success[1] and out.success[5]
    after = gpu.read_state()
    assert int(after.current_leader[0]) == abi.NO_NODE and int(after.commit_index[1]) == 30
    # host answers: term(15) == 2 -> the row applies
    s = abi.Batch(1, 1, gid=[0], hints=True)
    row = s.put(0, 0, abi.EV_AE_REQ, slot=1, a=9, b=15, c=2, d=0)
    s.set_hint(row, 2, 0)
    o2 = gpu.submit(s)
    assert o2.status[0] == abi.OK and o2.success[0]
    assert int(gpu.read_state().current_leader[0]) == 1


def test_sparse_rows_only_touch_their_groups():
    G, P = 300, 3
    st0 = fuzz.random_initial_state(G, P, 0, 5)
    gpu, orc = engine.Table(G, P), oracle_lib.OracleTable(G, P)
    gpu.load_state(st0)
    orc.load_state(st0)
    gids = np.array([0, 7, 63, 64, 65, 128, 299], dtype=np.uint32)
    b = abi.Batch(1, len(gids), gid=gids)
    for k in range(len(gids)):
        b.put(0, k, abi.EV_TIMEOUT)
    compare_outcomes(orc.submit(b, fill=0xAB), gpu.submit(b, fill=0xAB), "sparse")
    compare_states(orc.read_state(), gpu.read_state(), "sparse final")
    with pytest.raises(engine.EngineError):
        bad = abi.Batch(1, 2, gid=np.array([5, 5], dtype=np.uint32))
        gpu.submit(bad)


def fenced_timeouts_case(G=192):
    """RG_OPT_REQUIRE_FENCED_TIMEOUTS (ABI 4): with the option a TIMEOUT row whose aux is 0 is RG_BAD_EVENT and changes nothing, a row
    that names the participant whose ticket fired is decided as ever, a row that names a replaced one is RG_DROPPED_STALE_ROLE; without the option
    aux = 0 still means "whoever is current" (context/RaftRoutine.java:65-77 checks the ticket's participant on the timer thread, before the loop)."""
    st = abi.GroupState(G, 3)
    for g in range(G):
        set_state_follower(st, g, term=4, leader=1, last=20 + g)
        st.role_epoch[g] = 5 + g % 3
        if g % 4 == 1:
            st.role[g], st.voted_for[g], st.current_leader[g] = abi.CANDIDATE, 0, abi.NO_NODE
        if g % 4 == 2:
            st.role[g], st.voted_for[g], st.current_leader[g] = abi.LEADER, 0, abi.NO_NODE
    b = abi.Batch(3, G)
    for g in range(G):
        b.put(0, g, abi.EV_TIMEOUT, aux=0)                                   # un-fenced
        b.put(1, g, abi.EV_TIMEOUT, aux=int(st.role_epoch[g]) + (7 if g % 5 == 0 else 0))     # fenced: the live participant, or one that is gone
        b.put(2, g, abi.EV_TIMEOUT, aux=0)
    for required in (True, False):
        gpu, orc = engine.Table(G, 3, 0, True), oracle_lib.OracleTable(G, 3, 0, True)
        for t in (gpu, orc):
            t.set_option(abi.OPT_REQUIRE_FENCED_TIMEOUTS, required)
            t.load_state(st)
        ref, got = orc.submit(b), gpu.submit(b)
        compare_outcomes(ref, got, "fenced timeouts (required=%s)" % required)
        compare_states(orc.read_state(), gpu.read_state(), "fenced timeouts (required=%s)" % required)
        status = abi.flags_status(got.reply["flags"]).reshape(3, G)
        if required:
            assert np.all(status[0] == abi.BAD_EVENT) and np.all(status[2] == abi.BAD_EVENT)
            assert not np.any(got.reply["flags"].reshape(3, G)[0] & (abi.F_PERSIST | abi.F_ROLE_CHANGED | abi.F_RESET_TIMER | abi.F_EMIT_MASK))
            stale = np.arange(G) % 5 == 0                                    # (round 0 changed nothing: the epochs of round 1 are judged against the loaded ones)
            assert np.all(status[1][stale] == abi.DROPPED_STALE_ROLE) and np.all(status[1][~stale] == abi.OK)
        else:
            assert np.all(status[0] == abi.OK) and np.all(status[2] == abi.OK)
        gpu.close()
        orc.close()


def test_unfenced_timeouts_are_refused_where_the_table_requires_fences():
    fenced_timeouts_case()


def abi4_misuse_case():
    """ABI 4's entry points refuse what their header says they refuse, with a message (every check below happens on the host, before any device call)."""
    import ctypes as C
    G = 64
    t = engine.Table(G, 3)
    L = engine.lib()
    with pytest.raises(engine.EngineError, match="unknown option"):
        t.set_option(12345, 1)
    with pytest.raises(engine.EngineError, match="negative"):
        t.set_index_base(np.array([5, -1, 7], dtype=np.int64))
    with pytest.raises(engine.EngineError, match="groups"):
        t.set_index_base(np.zeros(8, dtype=np.int64), first=G - 4)
    b = abi.Batch(1, G)
    b32 = engine.pack32(b)
    out = abi.Outcome32(G, wide=True)
    cb, co = b32.as_struct(), out.as_struct()
    co.wide.logfx = None                                   # two of the three overflow columns
    assert L.rg_submit32c(t._h, C.byref(cb), C.byref(co), abi.MEM_HOST) < 0 and b"all three or none" in L.rg_last_error(t._h)
    gid = np.arange(G, dtype=np.uint32)
    sparse = engine.pack32(abi.Batch(1, G, gid=gid))
    cs, co = sparse.as_struct(), abi.Outcome32(G, wide=False).as_struct()
    assert L.rg_submit32c(t._h, C.byref(cs), C.byref(co), abi.MEM_HOST) < 0 and b"dense batches only" in L.rg_last_error(t._h)
    co = abi.Outcome32(G, wide=False).as_struct()
    co.persist = None
    assert L.rg_submit32c(t._h, C.byref(cb), C.byref(co), abi.MEM_HOST) < 0 and b"required" in L.rg_last_error(t._h)
    h = C.c_void_p()
    assert L.rg_tick_create(t._h, None, None, C.byref(h)) < 0 and not h.value and b"null batch" in L.rg_last_error(t._h)
    n = C.c_uint64()
    assert L.rg_wide_body_workgroups(t._h, None, 0) < 0
    # the host-side converters refuse missing columns without a table
    assert L.rg_outcome32_unpack(None, 1, 1, None, None) == -1
    assert L.rg_batch32_pack_rel(None, None, None, None, None) == -1
    t.close()


def test_abi4_misuse_is_reported(step_kernel_variant):
    if step_kernel_variant != "split":
        pytest.skip("host-side checks: once")
    abi4_misuse_case()


def test_api_misuse_is_reported():
    gpu = engine.Table(8, 3)
    with pytest.raises(engine.EngineError):
        gpu.submit(abi.Batch(1, 7))                                         # dense batch of the wrong width
    st = abi.GroupState(8, 3)
    st.role[3] = 7
    with pytest.raises(engine.EngineError):
        gpu.load_state(st)
    st = abi.GroupState(8, 3)
    st.set_log(0, 5, [(5, 1)], 9)                                           # first index not adjacent to epoch (0,0)
    with pytest.raises(engine.EngineError):
        gpu.load_state(st)
    with pytest.raises(engine.EngineError):
        engine.Table(8, abi.MAX_CLUSTER + 1)                                 # (8 .. 15 nodes are clusters like any other since ABI 5)
    with pytest.raises(engine.EngineError):
        engine.Table(8, 1)


def test_copy_bandwidth_reports_something_sane():
    gpu = engine.Table(8, 3)
    gbps = gpu.copy_bandwidth(1 << 28, 5)
    assert 500.0 < gbps < 8000.0


def test_replicate_matches_oracle_on_fuzzed_state():
    """N1: the send-side kernel against the oracle's Leader.replicateLog on states reached by random traffic."""
    G, P = 2048, 5
    st0 = fuzz.random_initial_state(G, P, 1, 31)
    gpu, orc = engine.Table(G, P, 1, True), oracle_lib.OracleTable(G, P, 1, True)
    gpu.load_state(st0)
    orc.load_state(st0)
    fz = fuzz.Fuzzer(G, P, 1, 31, allow_miss=False)
    rng = np.random.default_rng(5)
    seen = np.zeros(5, dtype=np.int64)
    for r in range(40):
        b = abi.Batch(1, G)
        cur = gpu.read_state()
        fz.round(cur, b, 0)
        og = gpu.submit(b, fill=0xAB)
        _resolve_need_host(gpu, orc, b, og, cur)
        compare_outcomes(orc.submit(b, fill=0xAB), og, "round %d" % r)
        if r % 4 == 3:
            hb = rng.integers(0, 2, G).astype(np.uint8)
            fl = rng.choice([0, 1, 2, 3, 20, 21], size=(G, P - 1)).astype(np.uint16)
            gids = None if r % 8 == 3 else np.flatnonzero(rng.random(G) < 0.3).astype(np.uint32)
            sel = slice(None) if gids is None else gids
            hg, sg = gpu.replicate(gids, hb[sel], fl[sel])
            ho, so = orc.replicate(gids, hb[sel], fl[sel])
            need = sg["kind"] == abi.SEND_NEED_HOST                # cache miss: the host would look prevLogTerm up itself
            assert np.array_equal(hg, ho)
            for f in ("prev_index", "last_index", "count"):
                assert np.array_equal(sg[f], so[f]), f
            assert np.array_equal(sg["kind"][~need], so["kind"][~need]) and np.array_equal(sg["prev_term"][~need], so["prev_term"][~need])
            assert np.all(so["kind"][need] == abi.SEND_APPEND)
            seen += np.bincount(so["kind"].reshape(-1), minlength=5)[:5]
            compare_states(orc.read_state(), gpu.read_state(), "after replicate %d" % r)
    assert seen[abi.SEND_NONE] and seen[abi.SEND_APPEND] and seen[abi.SEND_GATED] and seen[abi.SEND_SNAPSHOT]


def test_timer_driven_replay_matches_oracle():
    """N4 in a closed loop: expired timers (ballot-compacted on the device) become TIMEOUT rows, every batch's replies
    re-arm the deadlines; lists, outcomes and deadlines must equal the oracle's round after round."""
    G, P = 4096, 5
    st0 = fuzz.random_initial_state(G, P, 2, 77)
    gpu, orc = engine.Table(G, P, 2, True), oracle_lib.OracleTable(G, P, 2, True)
    fz = fuzz.Fuzzer(G, P, 2, 77, allow_miss=False)
    for t in (gpu, orc):
        t.load_state(st0)
        t.timers_configure(900, 300, 1234)
        t.timers_arm(10_000)
    assert np.array_equal(gpu.timers_read(), orc.timers_read())
    fired = 0
    for r in range(60):
        now = 10_000 + 150 * r
        eg, ng = gpu.timers_expired(now, capacity=G if r % 7 else 97)     # a short buffer now and then
        eo, no = orc.timers_expired(now, capacity=G if r % 7 else 97)
        assert ng == no and np.array_equal(eg, eo) and np.all(np.diff(eg.astype(np.int64)) > 0)
        fired += len(eg)
        b = abi.Batch(1, G)
        cur = gpu.read_state()
        fz.round(cur, b, 0)
        for g in eg:                                                       # the expired groups get their onTimeout instead
            b.head[int(g)] = (int(abi.hdr_make(abi.EV_TIMEOUT)), 0)
        og = gpu.submit(b, fill=0xAB)
        _resolve_need_host(gpu, orc, b, og, cur)                           # an ack's quorum index may lie below the cached runs
        oo = orc.submit(b, fill=0xAB)
        compare_outcomes(oo, og, "round %d" % r)
        gpu.timers_update(1, G, og.reply, [now])
        orc.timers_update(1, G, oo.reply, [now])
        assert np.array_equal(gpu.timers_read(), orc.timers_read()), r
    assert fired > G // 4
    # multi-round update in one call == round by round
    big = abi.Batch(3, G)
    for k in range(3):
        big.put(k, 5 + k, abi.EV_TIMEOUT)
        big.put(k, 900, abi.EV_TIMEOUT)
    og, oo = gpu.submit(big), orc.submit(big)
    compare_outcomes(oo, og, "multi-round")
    nows = [30_000, 30_100, 30_250]
    gpu.timers_update(3, G, og.reply, nows)
    orc.timers_update(3, G, oo.reply, nows)
    assert np.array_equal(gpu.timers_read(), orc.timers_read())


def test_health_replay_matches_oracle():
    """N4b in a closed loop: acks fold into requestSuccess / recentFailure through rg_health_update (derived from the reply
    rows), RPC failures through rg_health_failure; statistics and Leader.isReady must equal the oracle, which keeps them
    inside its ack handler where the reference does."""
    G, P = 4096, 5
    rng = np.random.default_rng(5)
    st0 = fuzz.random_initial_state(G, P, 1, 91)
    gpu, orc = engine.Table(G, P, 1, True), oracle_lib.OracleTable(G, P, 1, True)
    fz = fuzz.Fuzzer(G, P, 1, 91, allow_miss=False)
    for t in (gpu, orc):
        t.load_state(st0)
    ready_seen = np.zeros(2, dtype=np.int64)
    for r in range(50):
        now = 50_000 + 40 * r
        b = abi.Batch(1, G)
        cur = gpu.read_state()
        fz.round(cur, b, 0)
        og = gpu.submit(b, fill=0xAB)
        _resolve_need_host(gpu, orc, b, og, cur)
        gpu.health_update(b, og.reply, [now])
        oo = orc.submit(b, fill=0xAB, now=[now])
        compare_outcomes(oo, og, "round %d" % r)
        n = int(rng.integers(0, G // 2))                                   # RPC errors / timeouts, repeats of a (group, peer) included
        fg = rng.integers(0, G, n).astype(np.uint32)
        fs = rng.integers(0, P, n).astype(np.uint8)                        # the self slot shows up too: ignored
        ff = rng.integers(0, 4, n).astype(np.uint8)
        gpu.health_failure(fg, fs, ff, now + 7)
        orc.health_failure(fg, fs, ff, now + 7)
        for a, c in zip(gpu.health_read(), orc.health_read()):
            assert np.array_equal(a, c), r
        compare_states(orc.read_state(), gpu.read_state(), "after failures %d" % r)
        for cp, cd in ((0, 0), (1, 0), (0, 60), (2, 100), (1, 10 ** 9)):
            rg_, ro_ = gpu.ready(now + 20, cp, cd), orc.ready(now + 20, cp, cd)
            assert np.array_equal(rg_, ro_), (r, cp, cd)
            ready_seen += np.bincount(ro_, minlength=2)[:2]
    assert ready_seen[0] and ready_seen[1]


def test_health_multi_round_and_sparse_fold():
    """rg_health_update over a multi-round batch (clock per round) and over a sparse batch == the oracle."""
    G, P = 256, 4
    st = make_state(P, G, role=abi.LEADER, term=5, voted_for=0, role_epoch=3, repl_prepared=1, log=simple_log(50, 5),
                    peers=[(0, 51, 0, 0, 0)] * 3)
    gpu, orc = engine.Table(G, P, 0, True), oracle_lib.OracleTable(G, P, 0, True)
    for t in (gpu, orc):
        t.load_state(st)
    big = abi.Batch(3, G)
    for g in range(G):
        big.put(0, g, abi.EV_AE_ACK, slot=1 + g % 3, flag=1, a=5, b=0, c=40, aux=3)
        if g % 4 == 1:
            big.put(1, g, abi.EV_AE_ACK, slot=1 + (g + 1) % 3, flag=g % 8 == 1, a=5 + (g % 16 == 5), b=0, c=45, aux=3)
        if g % 2 == 0:
            big.put(2, g, abi.EV_IS_ACK, slot=3, flag=0, a=5, b=0, aux=3 - (g % 6 == 0))
    nows = [90_000, 90_010, 90_005]                                        # the third clock runs behind: increaseMono
    og = gpu.submit(big, fill=0xAB)
    gpu.health_update(big, og.reply, nows)
    oo = orc.submit(big, fill=0xAB, now=nows)
    compare_outcomes(oo, og, "multi-round")
    for a, c in zip(gpu.health_read(), orc.health_read()):
        assert np.array_equal(a, c)
    assert gpu.health_read()[0].max() == 90_010
    gids = np.arange(3, G, 5, dtype=np.uint32)
    sp = abi.Batch(1, len(gids), gid=gids)
    for k in range(len(gids)):
        sp.put(0, k, abi.EV_AE_ACK, slot=2, flag=1, a=5, b=0, c=50, aux=3)
    og = gpu.submit(sp, fill=0xAB)
    gpu.health_update(sp, og.reply, [95_000])
    oo = orc.submit(sp, fill=0xAB, now=[95_000])
    compare_outcomes(oo, og, "sparse")
    for a, c in zip(gpu.health_read(), orc.health_read()):
        assert np.array_equal(a, c)
    assert np.array_equal(gpu.ready(95_001, 1, 10), orc.ready(95_001, 1, 10)) and gpu.ready(95_001, 1, 10).any()
    first, count = 17, 100                                                 # ranged read
    for a, c in zip(gpu.health_read(first, count), orc.health_read(first, count)):
        assert np.array_equal(a, c)


@pytest.mark.gpu
def test_pipelined_host_submissions_match_the_oracle():
    """rg_submit_async / rg_submit_wait: a stream of host-memory batches, two in flight (upload of batch k+1 overlapping the
    kernel and the download of batch k), dense and sparse rows, page-locked and pageable caller buffers — every outcome and
    the final state equal to the oracle's sequential replay; a synchronous call in between drains the pipeline first."""
    from rafting_amd import workload
    cfg = workload.config(3, 4096)
    gen = workload.ReplayGenerator(cfg)
    st0 = gen.initial_state()
    gpu = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    gpu.load_state(st0)
    orc.load_state(st0)
    batches = [gen.next_batch(4) for _ in range(7)]
    refs = [orc.submit(b, fill=0xAB) for b in batches]
    owners, outs = [], []
    for k, b in enumerate(batches):
        out = abi.Outcome(b.rounds * b.count, 0xAB)
        if k % 3 != 2:                                   # two of three batches live in page-locked memory, the third is pageable
            b.entry_terms = np.ascontiguousarray(b.entry_terms[:max(b.entry_count, 1)])
            for obj, names in ((b, ("head", "ab", "cd", "entry_terms")), (out, ("reply", "logfx", "persist"))):
                for nm in names:
                    view, own = engine.pinned_like(gpu, getattr(obj, nm))
                    setattr(obj, nm, view)
                    owners.append(own)
        outs.append(out)
    assert gpu.submit_wait() is None                     # nothing in flight yet
    for k in range(5):
        gpu.submit_async(batches[k], outs[k])            # the third call waits for the first batch by itself
    got = []
    while True:
        o = gpu.submit_wait()
        if o is None:
            break
        got.append(o)
    assert len(got) == 2                                 # three batches had already been waited for inside submit_async
    gpu.submit_async(batches[5], outs[5])
    mid = gpu.read_state()                               # a synchronous entry point drains the pipeline first
    assert gpu.submit_wait() is None and int(mid.role_epoch.sum()) > 0
    gpu.submit_async(batches[6], outs[6])
    gpu.sync()
    for k in range(7):
        compare_outcomes(refs[k], outs[k], "pipelined batch %d" % k)
    compare_states(orc.read_state(), gpu.read_state(), "after the pipeline")
    for own in owners:
        own.free()


@pytest.mark.gpu
def test_packed_pipelined_submissions_match_the_oracle():
    packed_pipeline_case(4096, 6)


def packed_pipeline_case(groups, rounds):
    """rg_submit_async_packed: event fields uploaded as int32 and widened on the device, logfx / persist returned as row-ordered packed
    lists written by the device into page-locked memory — rebuilt into dense outcomes they equal the oracle's, batch for batch
    (dense rounds, a sparse batch, a list capacity that is too small, mixed with the wide entry point), and so does the final state."""
    from rafting_amd import workload
    cfg = workload.config(5, groups)                     # churn: plenty of persist items as well
    gen = workload.ReplayGenerator(cfg)
    st0 = gen.initial_state()
    gpu = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    gpu.load_state(st0)
    orc.load_state(st0)
    batches = [gen.next_batch(rounds) for _ in range(6)]
    refs = [orc.submit(b) for b in batches]
    packed = [engine.PackedBatch(gpu, b) for b in batches[:5]]
    wide_out = abi.Outcome(batches[5].rounds * batches[5].count, 0xAB)
    for pb in packed[:4]:
        gpu.submit_async_packed(pb)                      # the third call waits for the first batch by itself
    gpu.submit_async_packed(packed[4])
    gpu.submit_async(batches[5], wide_out)               # the two entry points share one pipeline
    gpu.sync()
    for k, pb in enumerate(packed):
        nl, npers = int(pb.counts[0]), int(pb.counts[1])
        assert 0 < npers < nl < pb.rows                  # the lists really are shorter than the dense columns
        compare_outcomes(refs[k], pb.unpack(), "packed batch %d" % k)
    compare_outcomes(refs[5], wide_out, "wide batch after packed ones")
    compare_states(orc.read_state(), gpu.read_state(), "after the packed pipeline")

    # a sparse batch, and lists that do not fit: counts still report what there was, the items that fit are the first ones
    gid = np.arange(0, cfg.groups, 3, dtype=np.uint32)
    sb = abi.Batch(1, len(gid), gid=gid)
    for i in range(len(gid)):
        sb.put(0, i, abi.EV_TIMEOUT)
    made = []
    for caps in (None, (0, 5)):
        g2 = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        o2 = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        g2.load_state(st0)
        o2.load_state(st0)
        ref = o2.submit(sb)
        n_per = int(abi.has_persist(ref.reply["flags"]).sum())
        assert n_per > 8
        pb = engine.PackedBatch(g2, sb) if caps is None else engine.PackedBatch(g2, sb, logfx_cap=caps[0], persist_cap=caps[1])
        g2.submit_async_packed(pb)
        g2.sync()
        assert int(pb.counts[1]) == n_per and int(pb.counts[0]) == int(abi.has_logfx(ref.reply["flags"]).sum())
        if caps is None:
            compare_outcomes(ref, pb.unpack(), "sparse packed batch")
        else:
            assert np.array_equal(pb.persist[:5], ref.persist[abi.has_persist(ref.reply["flags"])][:5])
        compare_states(o2.read_state(), g2.read_state(), "after the sparse packed batch")
        made.append((pb, g2))
    for pb in packed:
        pb.free()
    for pb, g2 in made:
        pb.free()
        g2.close()

    # pageable list memory is refused (the device could not write it), and so is a value the narrow format cannot carry
    bad = engine.PackedBatch(gpu, batches[0])
    bad.c_out.counts = np.zeros(2, dtype=np.uint32).ctypes.data
    with pytest.raises(engine.EngineError, match="page-locked"):
        gpu.submit_async_packed(bad)
    bad.free()
    big = abi.Batch(1, cfg.groups)
    big.put(0, 0, abi.EV_AE_REQ, slot=1, a=1 << 40, b=1, c=1, d=0)
    assert not abi.batch_fits_32(big)


# ---- compact rows (rg_batch32_t / rg_submit32 / step32_kernel) ---------------------------------------------------------------------
def numpy_pack32(batch):
    """what rg_batch32_pack must produce, written independently (numpy + a python loop over the AppendEntries rows)"""
    rows = batch.rounds * batch.count
    head = batch.head.copy()
    head["hdr"] &= np.uint32(~(abi.HDR_HINT_BIT | abi.HDR_SAME_TERM | (1 << 11)) & 0xFFFFFFFF)
    q = np.zeros(rows, dtype=abi.QUAD32_DT)
    q["a"], q["b"], q["c"], q["d"] = batch.ab["x"], batch.ab["y"], batch.cd["x"], batch.cd["y"]
    terms = []
    kind, n = head["hdr"] & 0xF, head["hdr"] >> 12
    for r in np.flatnonzero((kind == abi.EV_AE_REQ) & (n > 0)):
        aux, k = int(head["aux"][r]), int(n[r])
        if aux + k > batch.entry_count:
            head["aux"][r] = 0xFFFFFFFF
            continue
        e = batch.entry_terms[aux:aux + k]
        if np.all(e == e[0]):
            head["hdr"][r] |= abi.HDR_SAME_TERM
            head["aux"][r] = int(e[0])
        else:
            head["aux"][r] = len(terms)
            terms.extend(int(x) for x in e)
    return head, q, np.array(terms, dtype=np.int32)


def compact_multi_round_case(G, P, rounds):
    """ONE multi-round launch of step32_kernel on HBM-resident compact rows against the oracle (outcomes, final state, counters); then the
    ways a workgroup leaves the 32-bit domain — a group state at 2^30 when the launch starts, a row field reaching 2^30 in the middle of
    the launch, a state value pushed towards 2^31 by client appends — each compared with the oracle and with neighbours that stay inside."""
    st0, batches, outs, _, misses, _ = _lockstep(G, P, 1, True, rounds, 21, allow_miss=False)
    assert misses == 0
    big, ref = fuzz.concat_batches(batches), fuzz.concat_outcomes(outs)
    gpu = engine.Table(G, P, 1, True)
    gpu.load_state(st0)
    db = engine.DeviceBatch32(gpu, big)
    emu_fallbacks()
    gpu.submit_device(db)
    gpu.sync()
    fb = emu_fallbacks()
    assert fb in (None, 0) or os.environ.get("RG_FORCE_WIDE"), "a fuzz stream of small values must stay in the 32-bit body (%r fallbacks)" % fb
    compare_outcomes(ref, db.outcome(), "compact multi-round")
    orc = oracle_lib.OracleTable(G, P, 1, True)
    orc.load_state(st0)
    orc.submit(big)
    compare_states(orc.read_state(), gpu.read_state(), "compact multi-round final")
    c = gpu.counters()
    assert c[0] == int(np.count_nonzero(big.head["hdr"] & 0xF))
    assert c[1] == int(np.count_nonzero(ref.reply["flags"] & abi.F_REPLIED))
    assert c[2] == int(np.count_nonzero(ref.reply["flags"] & abi.F_ROLE_CHANGED))
    assert c[3] == int(np.count_nonzero(ref.reply["flags"] & abi.F_COMMIT))
    db.free()
    gpu.close()

    # followers whose log tail sits just below / far below / above 2^30, fed AppendEntries with 4 entries per round: the groups of the
    # second workgroup cross 2^30 in round 3 (their rows leave the domain), one group of the third starts beyond it, the first never does
    G2, R2, LIM = 192, 8, 1 << 30
    base = np.full(G2, 1000, dtype=np.int64)
    base[64:128] = LIM - 10 - np.arange(64)
    base[130] = LIM + 12345
    st = abi.GroupState(G2, 3)
    for g in range(G2):
        set_state_follower(st, g, term=5, leader=1, last=int(base[g]))
    b = abi.Batch(R2, G2)
    last = base.copy()
    for r in range(R2):
        for g in range(G2):
            n = 4 if (g + r) % 3 else 0
            b.put(r, g, abi.EV_AE_REQ, slot=1, a=5 + (1 if (r == 4 and g % 7 == 0) else 0), b=int(last[g]), c=5 if r < 5 or g % 7 else 5,
                  d=int(last[g]) - 1, entries=[5] * n if n else None)
            last[g] += n
    _compact_vs_oracle(G2, 3, 0, st, b, "tails crossing 2^30", expect_fallbacks=2)

    # a leader appending 2^20 - 1 commands per round from just below 2^30: state grows past the row limit and on towards 2^31
    G3, R3 = 64, 640
    st = abi.GroupState(G3, 3)
    for g in range(G3):
        set_state_follower(st, g, term=3, leader=abi.NO_NODE, last=1000 + g)
        st.role[g], st.voted_for[g] = abi.LEADER, 0
    st.last_index[7] = LIM - (1 << 21)
    b = abi.Batch(R3, G3)
    for r in range(R3):
        for g in range(G3):
            if g == 7:
                b.put(r, g, abi.EV_CLIENT_APPEND, n=(1 << 20) - 1)
            elif r % 2 == 0:
                b.put(r, g, abi.EV_CLIENT_APPEND, n=1 + (g % 3))
            else:
                b.put(r, g, abi.EV_TIMEOUT)
    _compact_vs_oracle(G3, 3, 0, st, b, "client appends towards 2^31", expect_fallbacks=1)

    # round 4: the small fields the sign words of rg_tier1n.hpp compare have a domain too. Workgroup 0: leaders and candidates whose role epoch lies
    # at 2^30 + k from the start (the state load leaves the domain), answered by acks / vote replies that name it, and that do not; workgroup 1:
    # the same with small epochs but rows whose `aux` — a role epoch there — is 2^30 or 2^31 + 5 (the row leaves the domain: s_ne() would read an
    # operand that differs in bit 31 as EQUAL); workgroup 2: the same traffic entirely inside the domain, which must stay in the 32-bit body.
    G4, R4 = 192, 6
    st = abi.GroupState(G4, 5)
    log = (1, [(1, 4), (51, 5)], 100)
    for g in range(G4):
        big = g < 64
        ep = (LIM + g) if big else 3 + (g % 5)
        if g % 2:
            set_group(st, g, role=abi.CANDIDATE, term=6, voted_for=0, role_epoch=ep, votes=1 + (g % 3), log=log, commit=40)
        else:
            set_group(st, g, role=abi.LEADER, term=5, voted_for=0, role_epoch=ep, repl_prepared=1, log=log, peers=[(0, 101, 90, 0, 0)] * 4, commit=40)
    b = abi.Batch(R4, G4)
    for r in range(R4):
        for g in range(G4):
            ep = int(st.role_epoch[g])
            aux = ep if (g + r) % 3 else ep + 1                          # mostly the live participant, sometimes a fenced one
            if 64 <= g < 128 and r == 3:
                aux = LIM if g % 4 else (1 << 31) + 5                    # a row out of the domain in the middle of the launch
            if g % 2:
                b.put(r, g, abi.EV_RV_REPLY, slot=1 + (r % 4), flag=int(r % 2 == 0), a=6, aux=aux)
            else:
                b.put(r, g, abi.EV_AE_ACK, slot=1 + (r % 4), flag=1, a=5, b=0, c=100, aux=aux)
    _compact_vs_oracle(G4, 5, 0, st, b, "role epochs at and above 2^30", expect_fallbacks=2)


def set_state_follower(st, g, term, leader, last):
    st.role[g], st.current_term[g], st.voted_for[g], st.current_leader[g] = abi.FOLLOWER, term, leader, leader
    st.commit_index[g] = max(last - 2, 0)
    st.set_log(g, 1, [(1, term)], last)


def emu_fallbacks():
    """host emulation only: workgroups of step32_kernel that took the 64-bit body since the last call (None on the GPU)"""
    L = engine.lib()
    if not hasattr(L, "rg_emu_fallbacks"):
        return None
    L.rg_emu_fallbacks.restype = __import__("ctypes").c_long
    return int(L.rg_emu_fallbacks(1))


def _compact_vs_oracle(G, P, self_slot, st, batch, where, expect_fallbacks=None):
    emu_fallbacks()
    gpu = engine.Table(G, P, self_slot, True)
    orc = oracle_lib.OracleTable(G, P, self_slot, True)
    gpu.load_state(st)
    orc.load_state(st)
    ref = orc.submit(batch)
    db = engine.DeviceBatch32(gpu, batch)
    gpu.submit_device(db)
    gpu.sync()
    compare_outcomes(ref, db.outcome(), where)
    compare_states(orc.read_state(), gpu.read_state(), where + " final")
    fb = emu_fallbacks()
    if fb is not None and expect_fallbacks is not None and not os.environ.get("RG_FORCE_WIDE"):
        assert fb == expect_fallbacks, "%s: %d workgroups fell back to 64-bit arithmetic, expected %d" % (where, fb, expect_fallbacks)
    db.free()
    gpu.close()
    orc.close()


def compact_workload_case(groups, rounds):
    """BASELINE streams (config 3, config 5, config 3 with conflicting AppendEntries) as compact rows: C packer == numpy packer, and
    step32_kernel == oracle on outcomes and final state, two launches in a row"""
    import dataclasses
    from rafting_amd import workload
    for number in (3, 5, -3):
        cfg = workload.config(abs(number), groups)
        if number < 0:
            cfg = dataclasses.replace(cfg, p_conflict=0.008, name=cfg.name + " + conflicts")
        gen = workload.ReplayGenerator(cfg)
        st0 = gen.initial_state()
        gpu = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        gpu.load_state(st0)
        orc.load_state(st0)
        for k in range(2):
            b = gen.next_batch(rounds)
            b32 = engine.pack32(b)
            h, q, t = numpy_pack32(b)
            assert np.array_equal(b32.head, h) and np.array_equal(b32.abcd, q) and np.array_equal(b32.entry_terms[:b32.entry_count], t)
            assert b32.entry_count < max(b.entry_count, 1)          # most entries share their row's term
            ref = orc.submit(b)
            db = engine.DeviceBatch32(gpu, b32)
            gpu.submit_device(db)
            gpu.sync()
            compare_outcomes(ref, db.outcome(), "%s batch %d (compact)" % (cfg.name, k))
            db.free()
        compare_states(orc.read_state(), gpu.read_state(), cfg.name + " final (compact)")
        gpu.close()
        orc.close()


def adverse_mix_case(groups, rounds, launches=3):
    """bench.py's adverse mix (conflicting AppendEntries, election churn, 1 % of the follower rows a retransmitted heartbeat whose prevLog lies below
    the four cached term runs) through rg_submit32c against the oracle: every row but the misses bit-identical; a miss row answers RG_NEED_HOST on
    the device and plain success at the oracle's lossless log — it changes nothing either way, and the stream parks the group (RG_EV_NONE) until
    the launch ends — so the final states agree as well."""
    import dataclasses
    from rafting_amd import workload
    cfg = dataclasses.replace(workload.config(3, groups), p_conflict=0.005, p_miss=0.01, p_timeout=0.03, p_vote_req=0.008, name="config3 adverse mix")
    gen = workload.ReplayGenerator(cfg)
    st0 = gen.initial_state()
    gpu = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    gpu.load_state(st0)
    orc.load_state(st0)
    ep = st0.role_epoch
    misses = 0
    G_adv = cfg.groups
    missed_any = np.zeros(G_adv, dtype=bool)
    for k in range(launches):
        b = gen.next_batch(rounds)
        ref = orc.submit(b)
        db = engine.DeviceBatch32(gpu, b, compact=True)
        gpu.submit_device(db)
        gpu.sync()
        got, ep = engine.unpack32(db.outcome32(), db.rounds, db.count, ep)
        db.free()
        miss = abi.flags_status(got.reply["flags"]) == abi.NEED_HOST
        misses += int(miss.sum())
        missed_any |= miss.reshape(b.rounds, b.count).any(axis=0)
        # the oracle applied those rows: a successful AppendEntries reply that re-arms the timer and changes nothing
        assert np.all(abi.flags_status(ref.reply["flags"][miss]) == abi.OK) and np.all(ref.reply["flags"][miss] & abi.F_SUCCESS)
        assert not np.any(ref.reply["flags"][miss] & (abi.F_COMMIT | abi.F_LOG_APPEND | abi.F_LOG_TRUNC | abi.F_PERSIST))
        for o in (ref, got):                          # ... and is compared everywhere else
            o.reply["flags"][miss] = 0; o.reply["resp_term"][miss] = 0
            o.logfx[miss] = (0, 0)
        compare_outcomes(ref, got, "adverse mix, launch %d" % k)
        # parked: every row of the group after its miss is RG_EV_NONE
        kinds = (b.head["hdr"] & 0xF).reshape(b.rounds, b.count)
        after = np.cumsum(miss.reshape(b.rounds, b.count), axis=0) - miss.reshape(b.rounds, b.count) > 0
        assert not kinds[after].any()
    assert misses == gen.miss_rows and misses > 0
    assert gpu.counters()[5] == misses
    # the one thing the retransmitted heartbeat does where it IS applied: Follower.currentLeader = leaderId (member/Follower.java:54) — visible only in a group
    # that had not heard from its leader since its last conversion; the next AppendEntries sets it on both sides
    ref_st, got_st = orc.read_state(), gpu.read_state()
    late = missed_any & (got_st.current_leader == abi.NO_NODE) & (ref_st.current_leader != abi.NO_NODE)
    assert late.sum() < 0.01 * G_adv
    ref_st.current_leader[late] = abi.NO_NODE
    compare_states(ref_st, got_st, "adverse mix final")
    gpu.close()
    orc.close()


def index_base_case(G=256, P=5, rounds=40, seed=77, offset=1 << 40):
    """Groups whose logs were compacted around 2^40: every live index is huge, every span small. With the table's index bases a little
    below the epochs the compact formats carry the traffic — rows packed relative to the bases (rg_batch32_pack_rel), compact outcome rows unpacked
    back onto them — and the 32-bit body decides it: NO workgroup takes the 64-bit body (rg_wide_body_workgroups), every outcome row and the final
    state bit-identical to the oracle's, which works on the absolute values throughout. A second table without bases takes the same state and the same
    rows in the wide format: same answers, by the wide-row kernel."""
    st0 = fuzz.random_initial_state(G, P, 1, seed, offset=offset)
    base = np.full(G, offset - 1000, dtype=np.int64)
    gpu, orc = engine.Table(G, P, 1, True), oracle_lib.OracleTable(G, P, 1, True)
    gpu.set_index_base(base)
    assert np.array_equal(gpu.index_base(), base)
    gpu.load_state(st0)
    orc.load_state(st0)
    fz = fuzz.Fuzzer(G, P, 1, seed, allow_miss=True)
    gpu.wide_body_workgroups(reset=True)
    compact = rows_total = hist_ok = misses = 0
    for r in range(rounds):
        cur = gpu.read_state()
        b = abi.Batch(1, G)
        fz.round(cur, b, 0)
        # rows the format cannot hold (the fuzzer's dirt: a leaderCommit of 3 for a log that starts at 2^40, ...) travel beside the batch as ONE sparse
        # wide submit — what the ingress does with them (SealedBatch::wide); the dense compact batch has RG_EV_NONE in their place
        kind = b.head["hdr"] & 0xF
        ixf = np.array([0, 0xA, 0x6, 0x2, 0x2, 0x2, 0, 0, 0, 0, 0x1, 0x2, 0, 0, 0, 0], dtype=np.uint32)[kind]
        cols = (b.ab["x"], b.ab["y"], b.cd["x"], b.cd["y"])
        beside = np.zeros(G, dtype=bool)
        for k, col in enumerate(cols):
            is_ix = ((ixf >> k) & 1) != 0
            beside |= is_ix & (col != 0) & ((col <= base) | (col - base >= (1 << 31)))
            beside |= ~is_ix & ((col < 0) | (col >= (1 << 31)))
        dense = abi.Batch(1, G)
        dense.head[:], dense.ab[:], dense.cd[:] = b.head, b.ab, b.cd
        dense.entry_terms, dense.entry_count = b.entry_terms, b.entry_count
        dense.head["hdr"][beside] = 0
        b32 = engine.pack32(dense, index_base=base)
        compact += int(np.count_nonzero(kind[~beside]))
        rows_total += int(np.count_nonzero(kind))
        raw = gpu.submit32c(b32, fill=0xAB)
        assert not np.any(raw.row["flags"] & abi.F_WIDE_VALUES)
        got, _ = engine.unpack32(raw, 1, G, cur.role_epoch, index_base=base)
        # the rows really are relative: commitIndex after the row, on the base, is the table's
        c = raw.row["commit_index"].astype(np.int64)
        assert np.array_equal(np.where(c == 0, 0, c + base), gpu.read_state().commit_index)
        if beside.any():
            rows = np.flatnonzero(beside)
            o2 = gpu.submit(_subset(b, rows, rows.astype(np.uint32)), fill=0xAB)
            got.reply[rows], got.logfx[rows], got.persist[rows] = o2.reply, o2.logfx, o2.persist
        misses += _resolve_need_host(gpu, orc, b, got, cur)      # (hints from the host's log as it stands BEFORE the round: the oracle's, here)
        ref = orc.submit(b, fill=0xAB)
        compare_outcomes(ref, got, "index base, round %d" % r)
        hist_ok += int(np.count_nonzero(ref.status == abi.OK))
    compare_states(orc.read_state(), gpu.read_state(), "index base final")
    assert compact >= 0.97 * rows_total, "only %d of %d rows travelled in the compact format" % (compact, rows_total)
    # (the fuzzer's acks claim arbitrary matchIndex values — 25 billion entries behind the leader, say: such a group has left ANY 2^30 window and takes
    # its workgroup to the 64-bit body, rightly. The clean stream of index_base_workload_case is where "no workgroup leaves" is asserted.)
    assert gpu.wide_body_workgroups() <= rounds * ((G + 63) // 64) // 2, "most workgroups of groups at 2^40 with small spans must stay on the 32-bit body"
    assert hist_ok > rounds * G // 2
    # without a base the same groups cannot use the compact row format at all, and a compact-format launch on their state takes the 64-bit body
    plain = engine.Table(G, P, 1, True)
    plain.load_state(st0)
    with pytest.raises(engine.EngineError):
        engine.pack32(b)
    quiet = abi.Batch(1, G)                               # RG_EV_NONE rows: expressible in any format
    plain.submit32(quiet)
    assert plain.wide_body_workgroups() == (G + 63) // 64
    # a base above a live index: the group leaves the 32-bit domain (its workgroup runs the 64-bit body), decisions unchanged
    gpu2, orc2 = engine.Table(G, P, 1, True), oracle_lib.OracleTable(G, P, 1, True)
    bad = base.copy()
    bad[7] = int(st0.epoch_index[7]) + 5                  # above group 7's epoch
    gpu2.set_index_base(bad)
    gpu2.load_state(st0)
    orc2.load_state(st0)
    raw = gpu2.submit32c(engine.pack32(quiet, index_base=bad))
    got, _ = engine.unpack32(raw, 1, G, st0.role_epoch, index_base=bad)
    compare_outcomes(orc2.submit(quiet), got, "a base above the epoch")
    assert gpu2.wide_body_workgroups() == 1
    compare_states(orc2.read_state(), gpu2.read_state(), "a base above the epoch")
    for t in (gpu, orc, plain, gpu2, orc2):
        t.close()


def index_base_workload_case(groups=2048, rounds=24, offset=1 << 40):
    """the BASELINE stream with every log compacted at 2^40 (workload index_base): compact rows relative to base = 2^40 - 1 through rg_submit32c in
    multi-round launches, against the oracle on the absolute stream; no workgroup on the 64-bit body"""
    import dataclasses
    from rafting_amd import workload
    cfg = dataclasses.replace(workload.config(3, groups), index_base=offset, name="config3 at index base 2^40")
    gen = workload.ReplayGenerator(cfg)
    st0 = gen.initial_state()
    base = np.full(groups, offset - 1, dtype=np.int64)
    gpu, orc = engine.Table(groups, cfg.cluster, cfg.self_slot, cfg.pre_vote), oracle_lib.OracleTable(groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    gpu.set_index_base(base)
    gpu.load_state(st0)
    orc.load_state(st0)
    ep = st0.role_epoch
    for k in range(2):
        b = gen.next_batch(rounds)
        ref = orc.submit(b)
        db = engine.DeviceBatch32(gpu, engine.pack32(b, index_base=base), compact=True)
        gpu.submit_device(db)
        gpu.sync()
        got, ep = engine.unpack32(db.outcome32(), db.rounds, db.count, ep, index_base=base)
        db.free()
        compare_outcomes(ref, got, "index base workload, launch %d" % k)
        assert int(np.count_nonzero(ref.status != abi.OK)) < 0.001 * len(ref.status)
    compare_states(orc.read_state(), gpu.read_state(), "index base workload final")
    assert gpu.wide_body_workgroups() == 0
    gpu.close()
    orc.close()


def tick_path_case(G=192, P=5, ticks=24, seed=123):
    """The once-per-tick path (rg_tick_*): ONE prepared submission (a HIP graph over the upload, the step kernel, the list packing and the download),
    refilled and replayed tick after tick from the same page-locked buffers, against the oracle row for row — dense single-round ticks from the state-aware
    fuzzer, then a sparse tick shape (a fixed set of groups)."""
    st0 = fuzz.random_initial_state(G, P, 0, seed)
    gpu, orc = engine.Table(G, P, 0, True), oracle_lib.OracleTable(G, P, 0, True)
    gpu.load_state(st0)
    orc.load_state(st0)
    fz = fuzz.Fuzzer(G, P, 0, seed, allow_miss=False)
    shape = abi.Batch(1, G)
    pb = engine.PackedBatch(gpu, shape, entry_cap=8 * G)
    tick = engine.Tick(gpu, pb)
    for r in range(ticks):
        b = abi.Batch(1, G)
        fz.round(gpu.read_state(), b, 0)
        tick.refill(b)
        tick.launch()
        tick.wait()
        compare_outcomes(orc.submit(b), pb.unpack(), "tick %d" % r)
    compare_states(orc.read_state(), gpu.read_state(), "after %d ticks" % ticks)
    tick.close()
    pb.free()
    # a sparse shape: every third group, one row each
    gids = np.arange(0, G, 3, dtype=np.uint32)
    sp = abi.Batch(1, len(gids), gid=gids)
    pbs = engine.PackedBatch(gpu, sp, entry_cap=8 * len(gids))
    tk = engine.Tick(gpu, pbs)
    for r in range(6):
        full = abi.Batch(1, G)
        fz.round(gpu.read_state(), full, 0)
        sub = _subset(full, gids.astype(np.int64), gids)
        sub.hint = None
        sub.head["hdr"] &= ~np.uint32(abi.HDR_HINT_BIT)
        tk.refill(sub)
        tk.launch()
        tk.wait()
        compare_outcomes(orc.submit(sub), pbs.unpack(), "sparse tick %d" % r)
    compare_states(orc.read_state(), gpu.read_state(), "after the sparse ticks")
    # misuse: pageable list memory is refused, as by rg_submit_async_packed
    with pytest.raises(engine.EngineError):
        bad = engine.PackedBatch(gpu, shape)
        bad.c_out.counts = np.zeros(2, dtype=np.uint32).ctypes.data
        engine.Tick(gpu, bad)
    tk.close()
    pbs.free()
    gpu.close()
    orc.close()


def tick2_case(G=4096, P=5, ticks=40, seed=321, device_resident=False, nodes=None):
    """The device-resident tick (rg_tick2_*, ABI 5): step32c -> timers_update32 -> health_update32 -> timers_expired -> replicate -> ready as ONE HIP graph,
    driven by its own timers: the tickets that fire become the TIMEOUT rows (with their role epochs) of the NEXT tick, tick after tick. Every tick is held
    against the oracle doing the same with separate calls: outcome rows, deadlines, expired lists + epochs, health statistics, send table, readiness."""
    self_slot = 2 % P
    st0 = fuzz.random_initial_state(G, P, self_slot, seed)
    gpu, orc = engine.Table(G, P, self_slot, True), oracle_lib.OracleTable(G, P, self_slot, True)
    fz = fuzz.Fuzzer(G, P, self_slot, seed, allow_miss=False)
    for t in (gpu, orc):
        t.load_state(st0)
        t.timers_configure(900, 300, 4321)
        t.timers_arm(10_000)
    assert np.array_equal(gpu.timers_read(), orc.timers_read())
    # (nodes: how the tick is RECORDED — 1 = one kernel (the default), 2 = step + fused tail, 4 = step, fold, replicate, ready; read at rg_tick2_create)
    saved = os.environ.get("RG_TICK_NODES")
    if nodes is not None:
        os.environ["RG_TICK_NODES"] = str(nodes)
    try:
        tick = engine.Tick2(gpu, 1, entry_cap=8 * G, expired_cap=G, critical_point=1, cool_down_ms=60, device_resident=device_resident)
    finally:
        if nodes is not None:
            if saved is None:
                del os.environ["RG_TICK_NODES"]
            else:
                os.environ["RG_TICK_NODES"] = saved
    fired_g, fired_e = np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    seen_fired = seen_send = repaired = 0
    rng = np.random.default_rng(seed)
    for k in range(ticks):
        now = 10_000 + 150 * k
        b = abi.Batch(1, G)
        cur = gpu.read_state()
        fz.round(cur, b, 0)
        for g, e in zip(fired_g, fired_e):                  # the tickets that fired at the end of the previous tick: their onTimeout, fenced
            b.head[int(g)] = (int(abi.hdr_make(abi.EV_TIMEOUT)), int(e))
        assert abi.batch_fits_32(b)
        hb = (rng.random(G) < 0.5).astype(np.uint8)
        fl = rng.integers(0, 24, (G, P - 1)).astype(np.uint16)
        tick.refill(b, [now], heartbeat=hb, in_flight=fl.T.reshape(-1))
        tick.launch()
        tick.wait()
        got, _ = engine.unpack32(tick.outcome32(), 1, G, cur.role_epoch)
        # (an ack whose quorum index lies below the cached term runs: the host half of the NEED_HOST protocol, then the repaired rows folded like the others;
        #  what the graph derived for THOSE groups in this tick — send rows, readiness — was derived before the repair and is not compared)
        bad = np.flatnonzero(got.status == abi.NEED_HOST)
        if len(bad):
            repaired += _resolve_need_host(gpu, orc, b, got, cur)
            sub = _subset(b, bad, bad.astype(np.uint32))
            gpu.timers_update(1, len(bad), got.reply[bad], [now], gid=bad.astype(np.uint32))
            gpu.health_update(sub, got.reply[bad], [now])
        oo = orc.submit(b, now=[now])
        compare_outcomes(oo, got, "tick %d" % k)
        orc.timers_update(1, G, oo.reply, [now])
        eo, epo, no = orc.timers_expired_epochs(now, capacity=G)
        eg, epg, ng = tick.expired()
        assert ng == no and np.array_equal(eg, eo) and np.array_equal(epg, epo), k
        assert np.array_equal(gpu.timers_read(), orc.timers_read()), k
        for a, c in zip(gpu.health_read(), orc.health_read()):
            assert np.array_equal(a, c), k
        ok = np.ones(G, dtype=bool)
        ok[bad] = False
        (hg, sg), (ho, so) = tick.sends(), orc.replicate(heartbeat=hb, in_flight=fl)
        if len(bad):
            gpu.replicate(gid=bad.astype(np.uint32), heartbeat=hb[bad], in_flight=fl[bad])     # (prepareReplication of a repaired leader, as the oracle just ran it)
        for f in ("term", "leader_commit", "epoch_index", "epoch_term", "role_epoch", "is_leader"):
            assert np.array_equal(hg[f][ok], ho[f][ok]), (k, f)
        for f in ("prev_index", "prev_term", "last_index", "count", "kind"):
            assert np.array_equal(sg[f][ok], so[f][ok]), (k, f)
        assert np.array_equal(tick.readiness()[ok], orc.ready(now, 1, 60)[ok]), k
        compare_states(orc.read_state(), gpu.read_state(), "tick %d" % k)
        fired_g, fired_e = eg, epg
        seen_fired += len(eg)
        seen_send += int(np.count_nonzero(so["kind"] == abi.SEND_APPEND))
    assert seen_fired > 0 and seen_send > 0
    # the separate calls on compact rows give what the wide ones give: one more batch, through rg_submit32c + rg_timers_update32 / rg_health_update32
    b = abi.Batch(1, G)
    cur = gpu.read_state()
    fz.round(cur, b, 0)
    raw = gpu.submit32c(b)
    got, _ = engine.unpack32(raw, 1, G, cur.role_epoch)
    if not np.any(got.status == abi.NEED_HOST):
        oo = orc.submit(b, now=[99_000])
        gpu.timers_update32(1, raw, [99_000])
        gpu.health_update32(b, raw, [99_000])
        orc.timers_update(1, G, oo.reply, [99_000])
        assert np.array_equal(gpu.timers_read(), orc.timers_read())
        for a, c in zip(gpu.health_read(), orc.health_read()):
            assert np.array_equal(a, c)
    # a recording is refused once the table's options have moved on ...
    gpu.set_option(abi.OPT_REQUIRE_FENCED_TIMEOUTS, 1)
    with pytest.raises(engine.EngineError):
        tick.launch()
    tick.close()
    gpu.close()
    orc.close()
    # ... and knows no table once its table is gone: launch / wait answer -1, destroy only frees the handle (the same holds for rg_tick_*)
    small = engine.Table(64, 3, 0, True)
    orphan = engine.Tick2(small, 1, expired_cap=64)
    shape = abi.Batch(1, 64)
    pb = engine.PackedBatch(small, shape)
    old_tick = engine.Tick(small, pb)
    small.close()
    L = engine.lib()
    assert L.rg_tick2_launch(orphan._h) == -1 and L.rg_tick2_wait(orphan._h) == -1 and L.rg_tick2_destroy(orphan._h) == 0
    assert L.rg_tick_launch(old_tick._h) == -1 and L.rg_tick_wait(old_tick._h) == -1 and L.rg_tick_destroy(old_tick._h) == 0
    orphan._h = old_tick._h = None                          # (their page-locked columns went with the table's context: nothing to free through it any more)


def test_the_device_resident_tick_matches_the_oracle(step_kernel_variant):
    if step_kernel_variant != "compact-out32":
        pytest.skip("one route: the tick always runs the compact-row kernel with compact outcome rows")
    tick2_case()                                            # (recorded as ONE kernel: tick_kernel)
    tick2_case(G=1024, ticks=12, seed=77, device_resident=True)
    tick2_case(G=131072 + 64, ticks=6, seed=5, device_resident=True)      # (the 128-VGPR variant; 2 049 workgroups take the expiry's ticket)
    tick2_case(G=2048, ticks=16, seed=78, nodes=2)          # (step + tick_tail_kernel)
    tick2_case(G=2048, ticks=16, seed=79, nodes=4)          # (the step-by-step recording)


def test_the_once_per_tick_graph_matches_the_oracle(step_kernel_variant):
    if step_kernel_variant != "compact":
        pytest.skip("one route: the tick path always runs the compact-row kernel")
    tick_path_case()


def test_compact_multi_round_launch_and_domain_exits():
    compact_multi_round_case(1024, 5, 48)


def test_groups_at_two_to_the_forty_stay_on_the_32_bit_body(step_kernel_variant):
    if step_kernel_variant != "compact-out32":
        pytest.skip("one route: compact rows in, compact outcome rows out, bases set")
    index_base_case()
    index_base_workload_case()


def test_adverse_mix_stream_matches_the_oracle_but_for_its_cache_misses(step_kernel_variant):
    if step_kernel_variant not in ("compact-out32", "compact-out32-forced-wide"):
        pytest.skip("one stream, the compact-outcome route: on the 32-bit body and on the 64-bit body")
    adverse_mix_case(4096, 24)


def test_compact_workload_replays():
    compact_workload_case(groups=5000, rounds=24)
