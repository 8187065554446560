"""N2 + the host half of a8: the ingress (rafting_amd/host/ingress.hpp, rw_ingress_* of include/raftwire.h) — wire frames of many
connections and the host's own rows, laid out as the multi-round compact batches the step kernel takes. CPU only: the oracle stands
where the table would be (the emulation and GPU tests run the same flow against the kernels)."""
import random
import threading

import numpy as np
import pytest

from rafting_amd import abi, wirelib
from tests import ingress_flow, oracle_lib

NODES = [("10.0.0.%d" % (i + 1), 7000 + i) for i in range(7)]


def _oracle_deciders(groups, cluster, self_slot, pre_vote, st0):
    orc = oracle_lib.OracleTable(groups, cluster, self_slot, pre_vote)
    orc.load_state(st0)
    return orc, (lambda b32: orc.submit(wirelib.unpack32(b32))), (lambda sparse: orc.submit(sparse))


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed,max_rounds", [(5, 2, True, 71, 64), (3, 0, True, 72, 5), (7, 3, False, 73, 3), (2, 1, True, 74, 1)])
def test_history_through_frames_and_rounds_is_decided_like_row_by_row(cluster, self_slot, pre_vote, seed, max_rounds):
    """40 fuzzed rounds (every row class, assertion triggers, values beyond 2^31 included), as frames + local rows through the ingress: the
    sealed batches (all 40 rounds in one when max_rounds allows, else several with rows held back in order) give every group the same
    rows, in the same order, with the same replies and response frames as deciding the history row by row; the final states agree."""
    G, rounds = 96, 40
    st0, batches, outs, final = ingress_flow.history(G, cluster, self_slot, pre_vote, rounds, seed)
    orc, d32, dsp = _oracle_deciders(G, cluster, self_slot, pre_vote, st0)
    sealed = ingress_flow.drive(d32, dsp, G, cluster, batches, outs, max_rounds, NODES[:cluster])
    assert sealed >= (rounds + max_rounds - 1) // max_rounds
    from tests.helpers import compare_states
    compare_states(final, orc.read_state(), "after the ingress")


def _tagged_request(nodes_b, ctx, conn_seq, tag, big=False):
    # a requestVote whose lastLogIndex carries the tag (any decodable request would do)
    return wirelib.request_frame(nodes_b, wirelib.M_REQUEST_VOTE, ctx, conn_seq, (1 << 33) if big else 3, 1, tag, 2)


def test_concurrent_feeders_keep_per_connection_order_and_lose_nothing():
    """8 connections fed from 8 threads in random pieces while the flush thread seals every few milliseconds, 4 rounds per batch, hot groups
    that overflow them, and rows outside the compact format: every row comes out exactly once; the rows of one (connection, group) come out
    in the order they were sent; a wide row closes its group for the rest of the batch."""
    G, C, R, per_conn = 64, 8, 4, 3000
    nodes_b = wirelib.nodes_arg(NODES[:5])
    ing = wirelib.Ingress(G, R, C, nodes=NODES[:5])
    ctx = [b"c%d" % g for g in range(G)]
    for g in range(G):
        assert ing.add_context(ctx[g], g)
    rng = random.Random(5)
    streams, sent = [], {}
    for c in range(C):
        s = bytearray()
        for k in range(per_conn):
            g = rng.randrange(8) if rng.random() < 0.5 else rng.randrange(G)       # hot groups: more rows than rounds
            tag = c * per_conn + k
            s += _tagged_request(nodes_b, ctx[g], k, tag, big=rng.random() < 0.01)
            sent.setdefault((c, g), []).append(tag)
        streams.append(bytes(s))
    done = threading.Event()

    def feeder(c):
        r, at, data = random.Random(100 + c), 0, streams[c]
        while at < len(data):
            n = r.choice((1, 7, 100, 1500, 9000))
            assert ing.feed(c, data[at:at + n]) >= 0
            at += n

    threads = [threading.Thread(target=feeder, args=(c,)) for c in range(C)]
    for t in threads:
        t.start()
    got = {}
    total = 0

    def collect(s):
        nonlocal total
        G_ = G
        for r in range(s.batch.rounds):
            hdr = s.batch.head["hdr"][r * G_:(r + 1) * G_]
            for g in np.flatnonzero(hdr & 0xF):
                cell = r * G_ + g
                conn, seq = ing.origin(s.bank, cell)
                tag = int(s.batch.abcd["b"][cell])
                assert tag == conn * per_conn + seq and int(s.batch.abcd["a"][cell]) == 3
                got.setdefault((conn, int(g)), []).append(tag)
                total += 1
        wide_groups = set()
        for g, hdr, aux, q, terms, origin in s.wide:
            assert q[0] == 1 << 33 and g not in wide_groups            # at most one wide row per group and batch
            wide_groups.add(g)
            depth = int(np.count_nonzero(s.batch.head["hdr"][g::G_][:s.batch.rounds] & 0xF))
            # the wide row comes after every compact row of its group in this batch: rows its connection sent earlier are among them
            got.setdefault((origin[0], g), []).append(q[1])
            assert depth <= R
            total += 1
        ing.recycle(s.bank)

    while any(t.is_alive() for t in threads):
        collect(ing.seal())
        done.wait(0.002)
    for t in threads:
        t.join()
    while True:
        s = ing.seal()
        if s.rows == 0 and not s.wide:
            break
        collect(s)
    assert ing.held() == 0 and ing.refused() == 0
    assert total == C * per_conn
    assert got == sent


def test_what_is_not_a_decision_row_is_refused_and_counted():
    ing = wirelib.Ingress(4, 2, 2, nodes=NODES[:3])
    nodes_b = wirelib.nodes_arg(NODES[:3])
    assert ing.add_context(b"a", 0) and not ing.add_context(b"a", 1) and not ing.add_context(b"b", 0) and not ing.add_context(b"x" * 129, 2)
    assert not ing.add_context(b"z", 4)                                               # beyond the table
    ing.set_peer(0, 1)
    assert ing.feed(0, wirelib.request_frame(nodes_b, 3, b"nobody", 1, 1, 1, 1, 1)) == 0           # unknown context
    assert ing.feed(0, wirelib.frame(wirelib.ENQ, 2, b"appendEntries:a", b"\x00garbage")) == 0     # undecodable body
    assert ing.feed(0, wirelib.frame(wirelib.ENQ, 3, b"sayHello:a", b"")) == 0                      # unknown method
    assert ing.feed(0, wirelib.response_frame(1, b"a", 9, 5, True)) == 0                            # a response nobody waits for
    ing.sent(0, 9, wirelib.M_APPEND_ENTRIES, 0, role_epoch=4, epoch_at_send=11, last_index_sent=12)
    assert ing.feed(0, wirelib.response_frame(3, b"a", 9, 5, True)) == 0                            # same sequence, another method
    assert ing.feed(0, wirelib.response_frame(1, b"a", 9, 5, True)) == 1
    assert ing.feed(0, wirelib.response_frame(1, b"a", 9, 5, True)) == 0                            # the invocation was removed (AsyncService.remove)
    assert ing.feed(1, wirelib.response_frame(1, b"a", 9, 5, True)) == 0                            # a connection without a peer slot
    assert ing.refused() == 7
    s = ing.seal()
    assert s.rows == 1 and s.batch.rounds == 1
    assert int(s.batch.head["hdr"][0]) == abi.hdr_make(abi.EV_AE_ACK, 1, 1, 0) and int(s.batch.head["aux"][0]) == 4
    assert tuple(int(v) for v in s.batch.abcd[0]) == (5, 11, 12, 0) and ing.origin(s.bank, 0) is None
    with pytest.raises(RuntimeError):                                                 # one sealed batch is with the flusher at a time
        ing.seal()
    ing.recycle(s.bank)
    assert ing.seal().rows == 0
    assert ing.feed(0, b"\x07") == -1                                                 # not SOH: the connection is dead
    assert ing.feed(0, wirelib.request_frame(nodes_b, 3, b"a", 1, 1, 1, 1, 1)) == -1
    # exitContext / destroyContext: the id stops resolving at once, its group id is free for another context after the next seal
    ing2 = wirelib.Ingress(4, 2, 1, nodes=NODES[:3])
    for g, c in enumerate((b"p", b"q", b"r")):
        assert ing2.add_context(c, g)
    assert ing2.feed(0, wirelib.request_frame(nodes_b, 3, b"q", 1, 1, 1, 1, 1)) == 1
    assert ing2.remove_context(b"q") and not ing2.remove_context(b"q") and not ing2.remove_context(b"nobody")
    assert ing2.feed(0, wirelib.request_frame(nodes_b, 3, b"q", 2, 1, 1, 1, 1)) == 0 and ing2.refused() == 1
    assert ing2.feed(0, wirelib.request_frame(nodes_b, 3, b"r", 3, 1, 1, 1, 1)) == 1       # probing walks over the tombstone
    assert not ing2.add_context(b"s", 1)                                                   # not before the seal
    s2 = ing2.seal()
    assert s2.rows == 2 and ing2.add_context(b"s", 1) and ing2.add_context(b"q", 3)
    ing2.recycle(s2.bank)
    assert ing2.feed(0, wirelib.request_frame(nodes_b, 3, b"s", 4, 1, 1, 1, 1) + wirelib.request_frame(nodes_b, 3, b"q", 5, 1, 1, 1, 1)) == 2
    s3 = ing2.seal()
    assert [int(h) & 0xF for h in s3.batch.head["hdr"][:4]] == [0, abi.EV_RV_REQ, 0, abi.EV_RV_REQ]
    # a new TCP connection in its place: frames are read again; what was filed for the old one is gone
    ing.sent(0, 20, wirelib.M_APPEND_ENTRIES, 0, role_epoch=4)
    ing.reset_conn(0)
    half = wirelib.request_frame(nodes_b, 3, b"a", 7, 1, 1, 1, 1)
    assert ing.feed(0, half[:10]) == 0 and ing.feed(0, half[10:]) == 1
    before = ing.refused()
    assert ing.feed(0, wirelib.response_frame(1, b"a", 20, 5, True)) == 0 and ing.refused() == before + 1


def test_entries_of_several_terms_use_the_term_array_and_same_term_rows_do_not():
    ing = wirelib.Ingress(2, 4, 1, nodes=NODES[:3], entry_cap=5)
    nodes_b = wirelib.nodes_arg(NODES[:3])
    for g, c in enumerate((b"a", b"b")):
        assert ing.add_context(c, g)
    f = lambda ctx, seq, terms: wirelib.request_frame(nodes_b, 1, ctx, seq, 9, 2, 100, 7, 99, terms)   # noqa: E731
    assert ing.feed(0, f(b"a", 1, [7, 7, 7]) + f(b"a", 2, [7, 8]) + f(b"b", 3, [8, 9, 9]) + f(b"b", 4, [9, 9, 9, 10]) + f(b"b", 5, [4])) == 5
    assert ing.held() == 2 and ing.held_on(0) == 2
    s = ing.seal()
    assert s.rows == 3 and s.batch.rounds == 2 and ing.held() == 0                      # the four-term row found the array full: it waits, and so does b's next row
    w = wirelib.unpack32(s.batch)
    rows = {(r, g): (int(w.head["hdr"][r * 2 + g]) >> 12, [int(t) for t in w.entry_terms[int(w.head["aux"][r * 2 + g]):][:int(w.head["hdr"][r * 2 + g]) >> 12]])
            for r in range(2) for g in range(2)}
    assert rows == {(0, 0): (3, [7, 7, 7]), (1, 0): (2, [7, 8]), (0, 1): (3, [8, 9, 9]), (1, 1): (0, [])}
    assert int(s.batch.head["hdr"][0]) & abi.HDR_SAME_TERM and not int(s.batch.head["hdr"][2]) & abi.HDR_SAME_TERM and s.batch.entry_count == 5
    ing.recycle(s.bank)
    s2 = ing.seal()                                                                     # b's held rows, in order
    assert s2.rows == 2 and s2.batch.rounds == 2 and ing.held() == 0
    w2 = wirelib.unpack32(s2.batch)
    assert [int(w2.head["hdr"][r * 2 + 1]) >> 12 for r in range(2)] == [4, 1] and [int(t) for t in w2.entry_terms[:4]] == [9, 9, 9, 10]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_ingress_threads_under_sanitizers(tmp_path, sanitizer):
    """tests/native/ingress_race.cpp: feeder threads, a thread creating contexts, a thread adding the host's own rows and the flush thread
    sealing / emitting / recycling without waiting for anybody — under ThreadSanitizer (no data race on the cells, the counters, the index,
    the pending rings) and under ASan + UBSan; every row comes out exactly once, rows of one (connection, group) in order."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "rafting_amd", "host")
    exe = str(tmp_path / "ingress_race")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-std=c++17", "-I" + host, "-I" + os.path.join(root, "include"),
                        os.path.join(root, "tests", "native", "ingress_race.cpp"), os.path.join(host, "ingress.cpp"), os.path.join(host, "wire.cpp"),
                        os.path.join(host, "kryo_body.cpp"), "-pthread", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no %s sanitizer runtime for g++ here: %s" % (sanitizer, r.stderr[-200:]))
    for shards in ("1", "3"):
        p = subprocess.run([exe, "128", "5", "1500", "4", shards], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ingress race ok=1" in p.stdout, p.stdout + p.stderr[-3000:]
        assert "Sanitizer" not in p.stderr, p.stderr[-3000:]


@pytest.mark.parametrize("sanitizer", ["", "thread"])
def test_context_index_survives_churn(tmp_path, sanitizer):
    """tombstones of erased contexts used to pile up until a miss lookup span for ever (ContextIndex(64) after 1 000 - 2 000
    insert / erase / reclaim cycles with distinct ids). tests/native/context_index_churn.cpp runs 20 000 such cycles on 64 groups and 200 000 on
    1 024 while two threads look ids up (hits and misses) behind the seal barrier the owner's contract names: every probe ends, live ids
    resolve, erased ones do not, tombstones stay bounded (the slot array is rebuilt beside the readers)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "rafting_amd", "host")
    exe = str(tmp_path / "context_index_churn")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17"] + (["-fsanitize=" + sanitizer, "-fno-sanitize-recover=all"] if sanitizer else []) +
                       ["-I" + host, "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "native", "context_index_churn.cpp"),
                        os.path.join(host, "ingress.cpp"), os.path.join(host, "wire.cpp"), os.path.join(host, "kryo_body.cpp"), "-pthread", "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0 and sanitizer:
        pytest.skip("no %s sanitizer runtime for g++ here: %s" % (sanitizer, r.stderr[-200:]))
    assert r.returncode == 0, r.stderr[-2000:]
    for args in (["64", "20000"], ["1024", "20000" if sanitizer else "200000"]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "context index churn ok=1" in p.stdout, p.stdout + p.stderr[-3000:]
        assert "Sanitizer" not in p.stderr, p.stderr[-3000:]
        assert int(p.stdout.split("rebuilds")[1]) > 0, p.stdout


def test_replication_loop_over_frames_equals_the_in_memory_loop():
    """N1 joined to N2: what rg_replicate plans leaves the leader as request frames with filed invocation records (Ingress.encode_sends),
    the followers decide them from their ingress batches and answer with response frames (emit), the leader's ingress matches every response
    to its invocation and turns it into the ack row — 12 ticks, client commands in between. All three nodes must end exactly where the same
    loop ends when rows are built directly from plans and reply rows, and the leader must have committed what it appended."""
    from tests.helpers import compare_states
    mk = lambda g, p, s, pv: oracle_lib.OracleTable(g, p, s, pv)      # noqa: E731
    mem = ingress_flow.replication_loop(mk, 48, 12, 7, over_the_wire=False)
    net = ingress_flow.replication_loop(mk, 48, 12, 7, over_the_wire=True)
    for node in range(3):
        compare_states(mem[node], net[node], "node %d" % node)
    assert int(np.min(mem[0].commit_index)) > 10 and np.array_equal(mem[0].last_index, mem[1].last_index)


@pytest.mark.parametrize("groups,shards,max_rounds", [(96, 3, 64), (100, 3, 4), (64, 8, 2), (50, 7, 64)])
def test_one_ingress_in_front_of_several_tables(groups, shards, max_rounds):
    """SURVEY 8(e) at the ingress: one set of connections, N tables (block partition gpu = gid / ceil(G / N), a last shard that may be smaller).
    Rows are routed by group id as they are placed; every shard's sealed batch is decided by ITS table (groups 0 .. count-1) — the same rows,
    order, replies, response frames and final state as the history decided row by row on one table holding all groups."""
    from tests.helpers import compare_states
    P, self_slot = 5, 1
    st0, batches, outs, final = ingress_flow.history(groups, P, self_slot, True, 30, 300 + shards)
    per = -(-groups // shards)
    n = -(-groups // per)                                        # shards that hold groups (50 groups over 7: per = 8, 7 shards of 8,8,8,8,8,8,2)
    tables = []
    for k in range(n):
        first, count = k * per, min(per, groups - k * per)
        t = oracle_lib.OracleTable(count, P, self_slot, True)
        t.load_state(ingress_flow.slice_state(st0, first, count))
        tables.append(t)
    d32 = [(lambda b32, t=t: t.submit(wirelib.unpack32(b32))) for t in tables]
    dsp = [(lambda sp, t=t: t.submit(sp)) for t in tables]
    ingress_flow.drive(d32, dsp, groups, P, batches, outs, max_rounds, NODES[:P], shards=n if n == shards else shards)
    for k, t in enumerate(tables):
        compare_states(ingress_flow.slice_state(final, k * per, t.groups), t.read_state(), "shard %d" % k)


def test_a_flooded_group_does_not_make_every_seal_walk_its_backlog(tmp_path):
    """A third of 7 x 60 000 rows go to four groups while a batch takes 3 rows per group: the backlog of those groups grows to tens of
    thousands of rows and the flusher seals tens of thousands of batches. Rows held over more than one batch wait in one queue per group, so
    a seal costs what it can place — the run takes a second or two (it took minutes when every seal sorted and retried the whole backlog);
    every row still comes out exactly once and in order."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "rafting_amd", "host")
    exe = str(tmp_path / "ingress_race")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + host, "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "native", "ingress_race.cpp"),
                    os.path.join(host, "ingress.cpp"), os.path.join(host, "wire.cpp"), os.path.join(host, "kryo_body.cpp"), "-pthread", "-o", exe], check=True)
    for args in (["512", "7", "60000", "3", "1"], ["300", "6", "40000", "2", "4"]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and "ingress race ok=1" in p.stdout, p.stdout + p.stderr[-2000:]


def test_a_send_whose_previous_entry_lies_below_the_cached_runs_is_completed_from_the_hosts_log():
    """rg_replicate answers RG_SEND_NEED_HOST when prevLogIndex lies below the term runs the table caches (the newest four of six here):
    Ingress.encode_sends reads prevLogTerm from the host's RaftLog (term_of) and ships the request all the same."""
    from tests.helpers import make_state
    P, runs = 3, [(1 + 10 * k, 2 + k) for k in range(6)]                    # indices 1..60, terms 2..7
    st = abi.GroupState(1, P, runs_total=len(runs))
    st.role[0], st.current_term[0], st.voted_for[0], st.repl_prepared[0], st.role_epoch[0] = abi.LEADER, 7, 0, 1, 4
    st.commit_index[0], st.first_index[0], st.last_index[0], st.run_count[0], st.run_offset[0] = 60, 1, 60, len(runs), 0
    for k, (s, t) in enumerate(runs):
        st.run_start[k], st.run_term[k] = s, t
    st.peer_next_index[:] = [16, 61]                                        # follower 0 is far behind: prev = 15 (term 3, a forgotten run)
    st.peer_match_index[:] = [0, 60]
    leader = oracle_lib.OracleTable(1, P, 0, True)
    leader.load_state(st)
    head, send = leader.replicate(heartbeat=0)
    assert int(send[0, 0]["kind"]) == abi.SEND_APPEND and (int(send[0, 0]["prev_index"]), int(send[0, 0]["prev_term"])) == (15, 3)   # the oracle's log is lossless
    missed = send[:, 0].copy()                                              # ... the table's answer for the same state: the term is the host's to supply
    missed["kind"], missed["prev_term"] = abi.SEND_NEED_HOST, 0
    ing = wirelib.Ingress(1, 2, 2, nodes=NODES[:P])
    assert ing.add_context(b"g", 0)
    term_of = lambda g, i: next(t for s, t in reversed(runs) if s <= i)      # noqa: E731
    data, frames, from_log = ing.encode_sends(0, 0, head, missed, term_of)
    assert (frames, from_log) == (1, 1)
    (ftype, seq, scope, body), = wirelib.split_frames(data)
    q = wirelib.decode_request(wirelib.nodes_arg(NODES[:P]), 1, body)
    n = int(send[0, 0]["count"])
    assert scope == b"appendEntries:g" and (q[0], q[1], q[2], q[3]) == (7, 0, 15, 3) and q[5] == [term_of(0, 16 + k) for k in range(n)] and n > 0
    whole, frames0, from_log0 = ing.encode_sends(0, 0, head, send[:, 0], term_of)      # the complete row gives the same request (under the next sequence number)
    assert (frames0, from_log0) == (1, 0) and wirelib.split_frames(whole)[0][3] == body and wirelib.split_frames(whole)[0][1] == seq + 1


def test_rows_of_an_erased_context_do_not_reach_the_next_owner_of_its_group(tmp_path):
    """held and backlogged rows are keyed by group id; after ContextIndex::erase + reclaim + insert of ANOTHER context under that id they
    would have been decided by the new context's group. Ingress::drop_rows_of (called between erase and reclaim) drops and counts them; the rows
    already in the batch being filled are decided with it (tests/native/ingress_drop_rows.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "rafting_amd", "host")
    exe = str(tmp_path / "ingress_drop_rows")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-I" + host, "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "native", "ingress_drop_rows.cpp"), os.path.join(host, "ingress.cpp"), os.path.join(host, "wire.cpp"),
                    os.path.join(host, "kryo_body.cpp"), "-pthread", "-o", exe], check=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "drop rows ok" in p.stdout, p.stdout + p.stderr
