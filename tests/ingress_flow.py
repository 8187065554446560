"""Shared by the CPU, emulation and GPU tests of the ingress (rafting_amd/host/ingress.hpp): a fuzzed multi-round history, decided row by
row by the oracle, is turned into what a deployment would see — the reference's wire frames on peer connections plus the host's own rows
(timeouts, client commands, log flushes) — pushed through the ingress, and the batches it seals are decided by the table under test.
Every group must see the same rows in the same order with the same answers, and every request must get the response frame the oracle's
reply calls for. TEST INFRASTRUCTURE."""
import numpy as np

from rafting_amd import abi, wirelib
from tests import fuzz, oracle_lib

REQ_METHOD = {abi.EV_AE_REQ: 1, abi.EV_PV_REQ: 2, abi.EV_RV_REQ: 3, abi.EV_IS_REQ: 4}
ACK_METHOD = {abi.EV_AE_ACK: 1, abi.EV_PV_REPLY: 2, abi.EV_RV_REPLY: 3, abi.EV_IS_ACK: 4}
LOCAL_CONN = 16                       # connections 0..15 = the peer in that slot; 16 = the host's own rows


def history(groups, cluster, self_slot, pre_vote, rounds, seed, view=None, allow_miss=False):
    """(initial state, [one-round batches], [the oracle's outcomes], final state). view: a table of the kind under test (engine.Table), decided
    in lockstep — the fuzzer then draws its rows from THAT table's state image, whose cached term runs are the device's own, and its outcomes
    are held to the oracle's. allow_miss: rows whose lookups leave the cached runs are drawn too (the view answers RG_NEED_HOST, resolved with
    hints from the oracle's lossless log as tests/test_gpu_parity.py does): the ingress flow then has to repair them (drive(shadow=...))."""
    from tests.helpers import compare_outcomes
    st0 = fuzz.random_initial_state(groups, cluster, self_slot, seed)
    for g in range(5, groups, 23):                    # a few groups whose terms lie beyond int32: their rows cannot be compact rows
        st0.current_term[g] += 1 << 33
    orc = oracle_lib.OracleTable(groups, cluster, self_slot, pre_vote)
    orc.load_state(st0)
    if view is not None:
        view.load_state(st0)
    fz = fuzz.Fuzzer(groups, cluster, self_slot, seed, allow_miss=allow_miss)
    batches, outs = [], []
    for r in range(rounds):
        b = abi.Batch(1, groups)
        fz.round((view or orc).read_state(), b, 0)
        hdr = b.head["hdr"]
        cannot_travel = ((hdr & 0xF) == abi.EV_AE_REQ) & ((((hdr >> 4) & 0xF) >= cluster) | ((hdr >> 12) > abi.MAX_AE_ENTRIES))    # no NodeID / refused frame
        cannot_travel |= (hdr & abi.HDR_HINT_BIT) != 0
        b.head[cannot_travel] = (0, 0)
        if view is not None:
            cur = view.read_state()
            og = view.submit(b)
            if allow_miss:
                from tests import test_gpu_parity as T
                T._resolve_need_host(view, orc, b, og, cur)          # (needs the oracle's log as it is BEFORE the round)
        outs.append(orc.submit(b))
        if view is not None:
            compare_outcomes(outs[-1], og, "history round %d" % r)
        batches.append(b)
    return st0, batches, outs, orc.read_state()


def drive(table_decide32, table_decide_sparse, groups, cluster, batches, outs, max_rounds, nodes, shadow=None, raw_submit=None, shards=1):
    """Feeds the history, seals until everything was decided. table_decide32(abi.Batch32) and table_decide_sparse(abi.Batch with gid) return
    objects with .reply (REPLY_DT rows) and .logfx. Returns the number of batches sealed.
    shadow + raw_submit: the table answers RG_NEED_HOST for rows that leave its cached term runs. shadow is an OracleTable loaded with the initial
    state — it plays the host's RaftLog (lossless), kept in step with what the table has applied; raw_submit(CBatch*, COutcome*) is rg_submit on
    the table. Missed rows and the rows skipped behind them are then decided through rw_ingress_repair before the next batch is sealed.
    shards > 1: the ingress stands in front of that many tables (block partition); table_decide32 / table_decide_sparse are then LISTS, one per
    shard, each deciding its own table's groups 0 .. count-1 (the flow maps them back to global group ids)."""
    ctx = [b"group-%05d" % g for g in range(groups)]
    nodes_b = wirelib.nodes_arg(nodes)
    ing = wirelib.Ingress(groups, max_rounds, LOCAL_CONN + 1, nodes=nodes, entry_cap=1 << 18, shards=shards)
    ing.retain_bodies()
    if shards == 1:
        table_decide32, table_decide_sparse = [table_decide32], [table_decide_sparse]
    assert shadow is None or shards == 1
    for g in range(groups):
        assert ing.add_context(ctx[g], g)
    for s in range(16):
        ing.set_peer(s, s)
    seq = [1000 * (c + 1) for c in range(LOCAL_CONN + 1)]
    expected = [[] for _ in range(groups)]            # per group: (kind, reply row, key of the response frame or None)
    answers = {}                                      # (conn, sequence) -> (term, success) the requester must receive
    queued = 0
    for b, o in zip(batches, outs):
        streams = [bytearray() for _ in range(16)]
        for g in range(groups):
            hdr, aux = int(b.head["hdr"][g]), int(b.head["aux"][g])
            kind, slot, flag, n = hdr & 0xF, (hdr >> 4) & 0xF, (hdr >> 8) & 1, hdr >> 12
            if kind == abi.EV_NONE:
                continue
            a, bb, c, d = int(b.ab["x"][g]), int(b.ab["y"][g]), int(b.cd["x"][g]), int(b.cd["y"][g])
            rep = o.reply[g]
            key = None
            on_wire = slot < cluster and not (hdr & abi.HDR_HINT_BIT)
            if kind in REQ_METHOD and on_wire and not (kind == abi.EV_IS_REQ and flag):
                conn = slot
                key = (conn, seq[conn])
                terms = b.entry_terms[aux:aux + n] if kind == abi.EV_AE_REQ else ()
                streams[conn] += wirelib.request_frame(nodes_b, REQ_METHOD[kind], ctx[g], seq[conn], a, slot, bb, c, d if kind == abi.EV_AE_REQ else 0, terms)
                seq[conn] += 1
            elif kind in ACK_METHOD and on_wire:
                conn = slot
                ing.sent(conn, seq[conn], ACK_METHOD[kind], g, aux, bb if kind in (abi.EV_AE_ACK, abi.EV_IS_ACK) else 0, c if kind == abi.EV_AE_ACK else 0)
                streams[conn] += wirelib.response_frame(ACK_METHOD[kind], ctx[g], seq[conn], a, flag)
                seq[conn] += 1
            else:                                      # the host's own rows, and rows no peer could have put on the wire
                assert kind != abi.EV_AE_REQ            # (history() took out the AppendEntries rows that cannot be frames)
                if kind in REQ_METHOD:
                    key = (LOCAL_CONN, seq[LOCAL_CONN])
                    seq[LOCAL_CONN] += 1
                ing.add_row(LOCAL_CONN, g, hdr, aux, a, bb, c, d, *(key if key else (wirelib.NO_CONN, 0)))
            queued += 1
            expected[g].append((kind, rep.copy(), key))
            if key is not None and int(rep["flags"]) & abi.F_REPLIED:
                answers[key] = (int(rep["resp_term"]), bool(int(rep["flags"]) & abi.F_SUCCESS))
        for conn, s in enumerate(streams):             # (peers first, the local rows of this round were queued above: any interleaving is legal,
            if s:                                      # what must hold is the order WITHIN a connection)
                got = ing.feed(conn, bytes(s))
                assert got >= 0
    # NOTE: a local row and a frame of the same round never address the same group (one row per group per round), so the per-group order of the
    # history is the order of arrival whatever the interleaving of connections inside a round.
    assert ing.refused() == 0
    seen = [0] * groups
    got_answers = {}
    sealed = 0
    repaired = []
    while True:
        s = ing.seal()
        if s.rows == 0 and not s.wide:
            assert ing.held() == 0
            ing.recycle(s.bank)
            break
        sealed += 1
        assert len(s.shards) == shards and sum(ev for _, ev, _ in s.shards) == s.rows
        for k, (b32, events, first) in enumerate(s.shards):
            if not b32.rounds:
                assert events == 0
                continue
            G = b32.count
            out = table_decide32[k](b32)
            reply = out.reply
            if shadow is not None:
                status = (reply["flags"] >> abi.F_STATUS_SHIFT) & 0xFF
                unapplied = (status == abi.NEED_HOST) | (status == abi.SKIPPED_AFTER_NEED_HOST)
                wide_form = wirelib.unpack32(b32)
                applied_part = wirelib.unpack32(b32)
                applied_part.head[unapplied] = (0, 0)
                shadow.submit(applied_part)                       # the host's log follows what the table applied
                if unapplied.any():
                    repaired.append(int(np.count_nonzero(unapplied & ((b32.head["hdr"] & 0xF) != 0))))

                    def on_applied(gid, cell, rep, lfx, per):
                        one = abi.Batch(1, 1, gid=np.array([gid], dtype=np.uint32))
                        hdr, aux = int(wide_form.head["hdr"][cell]), int(wide_form.head["aux"][cell])
                        n = hdr >> 12
                        ents = wide_form.entry_terms[aux:aux + n] if (hdr & 0xF) == abi.EV_AE_REQ and n else None
                        one.put(0, 0, hdr & 0xF, slot=(hdr >> 4) & 0xF, flag=(hdr >> 8) & 1, a=int(wide_form.ab["x"][cell]), b=int(wide_form.ab["y"][cell]),
                                c=int(wide_form.cd["x"][cell]), d=int(wide_form.cd["y"][cell]), aux=aux, entries=ents, n=n)
                        shadow.submit(one)

                    def term_at(gid, index):
                        t = shadow.log_term(gid, index)
                        return -1 if t is None else t

                    got = ing.repair(s.bank, reply, out.logfx, False, term_at, lambda g, first_, terms: shadow.log_conflict(g, first_, np.array(terms, dtype=np.int64)),
                                     lambda g: int(shadow.read_state(g, 1).epoch_index[0]), raw_submit, on_applied)
                    assert got == repaired[-1], (got, repaired[-1])
            n_events = 0
            for r in range(b32.rounds):
                kinds = b32.head["hdr"][r * G:(r + 1) * G] & 0xF
                for lg in np.flatnonzero(kinds):
                    g = first + int(lg)
                    kind, rep, key = expected[g][seen[g]]
                    assert kind == kinds[lg], (g, seen[g], kind, kinds[lg])
                    got = reply[r * G + lg]
                    assert (int(got["flags"]), int(got["role_epoch"])) == (int(rep["flags"]), int(rep["role_epoch"])), (g, seen[g], kind)
                    if int(rep["flags"]) & abi.F_REPLIED:
                        assert int(got["resp_term"]) == int(rep["resp_term"])
                    assert ing.origin(s.bank, r * G + lg, shard=k) == key
                    kept = ing.body(s.bank, r * G + lg, shard=k)             # the request body, for the host's RaftLog: exactly the row's request
                    hdr_c = int(b32.head["hdr"][r * G + lg])
                    if kind == abi.EV_AE_REQ and (hdr_c >> 12) > 0:
                        q = wirelib.decode_request(nodes_b, 1, kept)
                        cell_q = b32.abcd[r * G + lg]
                        assert q is not None and (q[0], q[2], q[3], q[4]) == tuple(int(v) for v in cell_q) and len(q[5]) == hdr_c >> 12
                    else:
                        assert kept is None
                    seen[g] += 1
                    n_events += 1
            assert n_events == events
            for conn in range(LOCAL_CONN + 1):
                for ftype, sq, head, body in wirelib.split_frames(ing.emit(s.bank, reply, conn, shard=k)):
                    assert ftype == wirelib.ACK and (conn, sq) not in got_answers
                    got_answers[(conn, sq)] = wirelib.decode_response(body)
                    g = int(head.split(b"-")[1])
                    assert first <= g < first + G
                    kind, rep, key = next(e for e in expected[g] if e[2] == (conn, sq))
                    assert head == wirelib.METHOD_NAME[REQ_METHOD[kind]] + b":" + ctx[g]
        per = -(-groups // shards)
        for k in range(shards):                        # rows the compact format cannot hold: one sparse round per table after the batch
            mine = [w for w in s.wide if w[0] // per == k]
            if not mine:
                continue
            first = k * per
            sp = abi.Batch(1, len(mine), gid=np.array([w[0] - first for w in mine], dtype=np.uint32))
            for i, (g, hdr, aux, q, terms, origin) in enumerate(mine):
                sp.put(0, i, hdr & 0xF, slot=(hdr >> 4) & 0xF, flag=(hdr >> 8) & 1, a=q[0], b=q[1], c=q[2], d=q[3], aux=aux, entries=terms or None, n=hdr >> 12)
            reply = table_decide_sparse[k](sp).reply
            for i, (g, hdr, aux, q, terms, origin) in enumerate(mine):
                kind, rep, key = expected[g][seen[g]]
                assert kind == hdr & 0xF and origin == key
                assert (int(reply[i]["flags"]), int(reply[i]["role_epoch"])) == (int(rep["flags"]), int(rep["role_epoch"])), (g, seen[g], kind, "wide")
                frame = ing.emit_wide(s.bank, s.wide.index(mine[i]), reply[i])
                if key is not None and int(rep["flags"]) & abi.F_REPLIED:
                    (ftype, sq, head, body), = wirelib.split_frames(frame[1])
                    assert frame[0] == key[0] and (ftype, sq) == (wirelib.ACK, key[1]) and head == wirelib.METHOD_NAME[REQ_METHOD[kind]] + b":" + ctx[g]
                    got_answers[key] = wirelib.decode_response(body)
                else:
                    assert frame is None
                seen[g] += 1
        ing.recycle(s.bank)
    assert seen == [len(e) for e in expected] and sum(seen) == queued
    assert got_answers == answers
    ing.close()
    return (sealed, sum(repaired)) if shadow is not None else sealed


def decide_sparse_with_hints(table, shadow, sp):
    """rg_submit of a sparse single-round batch (the wide rows beside an ingress batch) with the host half of the RG_NEED_HOST protocol:
    hints from the shadow log, resubmission of the missed rows; the shadow then follows. Returns the outcome with the final rows."""
    out = table.submit(sp)
    rows = np.flatnonzero(out.status == abi.NEED_HOST)
    for row in rows:
        g = int(sp.gid[row])
        hdr, aux = int(sp.head["hdr"][row]), int(sp.head["aux"][row])
        kind, n = hdr & 0xF, hdr >> 12
        ents = sp.entry_terms[aux:aux + n] if kind == abi.EV_AE_REQ and n else None
        one = abi.Batch(1, 1, gid=np.array([g], dtype=np.uint32), hints=True)
        r = one.put(0, 0, kind, slot=(hdr >> 4) & 0xF, flag=(hdr >> 8) & 1, a=int(sp.ab["x"][row]), b=int(sp.ab["y"][row]), c=int(sp.cd["x"][row]),
                    d=int(sp.cd["y"][row]), aux=aux, entries=ents, n=n)
        if kind == abi.EV_AE_REQ:
            prev, eidx = int(sp.ab["y"][row]), int(shadow.read_state(g, 1).epoch_index[0])
            terms, e0 = ([] if ents is None else list(ents)), prev + 1
            if n and e0 <= eidx:
                skip = min(n, eidx - e0 + 1)
                terms, e0 = terms[skip:], e0 + skip
            pt = shadow.log_term(g, prev)
            one.set_hint(r, -1 if pt is None else pt, shadow.log_conflict(g, e0, terms) if len(terms) else 0)
        else:
            idx = int(out.logfx["log_from"][row])
            t = shadow.log_term(g, idx)
            one.set_hint(r, idx, -1 if t is None else t)
        o2 = table.submit(one)
        assert o2.status[0] != abi.NEED_HOST
        out.reply[row], out.logfx[row], out.persist[row] = o2.reply[0], o2.logfx[0], o2.persist[0]
    shadow.submit(sp)
    return out


# ---- a closed replication loop over real frames: Leader.replicateLog -> wire -> Follower.appendEntries -> wire -> the leader's ack callback ------
def replication_loop(make_table, groups, ticks, seed, over_the_wire):
    """Three nodes (slot 0 leads every group, slots 1 and 2 follow), `ticks` rounds of: client commands at the leader -> rg_replicate -> the
    AppendEntries the plan calls for -> the followers decide -> their responses -> the leader's ack rows (match / commit advance).
    over_the_wire = False: rows are built directly from the send plans and the reply rows (the in-memory reference run).
    over_the_wire = True: the plans become request frames (Ingress.encode_sends, which files the invocation records), every node's inbound
    bytes go through its Ingress, replies go back as response frames (emit) and are matched to their invocations by (connection, sequence).
    Returns the three final states and the leader's commit indices. make_table(groups, cluster, self_slot, pre_vote) -> a table with
    load_state / read_state / submit / replicate (OracleTable or engine.Table)."""
    import random
    from tests.helpers import make_state, simple_log
    P, TERM, LAST0 = 3, 5, 10
    rng = random.Random(seed)
    nodes = [("10.2.0.%d" % i, 7100 + i) for i in range(P)]
    tabs = [make_table(groups, P, s, True) for s in range(P)]
    tabs[0].load_state(make_state(P, groups, role=abi.LEADER, term=TERM, voted_for=0, repl_prepared=1, role_epoch=4, commit=LAST0, log=simple_log(LAST0, TERM),
                                  peers=[(0, LAST0 + 1, LAST0, 0, 0)] * 2))
    for s in (1, 2):
        tabs[s].load_state(make_state(P, groups, role=abi.FOLLOWER, term=TERM, voted_for=0, leader=0, role_epoch=2, commit=LAST0, log=simple_log(LAST0, TERM)))
    ctx = [b"kv/%04d" % g for g in range(groups)]
    ing = None
    if over_the_wire:
        # leader: connection j -> follower slot j + 1, connection 2 = its own rows; follower: connection 0 -> the leader, connection 1 = its own rows
        ing = [wirelib.Ingress(groups, 4, 3, nodes=nodes), wirelib.Ingress(groups, 4, 2, nodes=nodes), wirelib.Ingress(groups, 4, 2, nodes=nodes)]
        for i in ing:
            for g in range(groups):
                assert i.add_context(ctx[g], g)
        ing[0].set_peer(0, 1)
        ing[0].set_peer(1, 2)
        ing[1].set_peer(0, 0)
        ing[2].set_peer(0, 0)

    def decide_all(node, then=None):
        """seal node's ingress until it is empty; every batch is decided, handed to then(sealed, outcome) and recycled before the next seal
        (no wide rows, no misses in this workload)"""
        while True:
            s = ing[node].seal()
            assert not s.wide
            if s.rows:
                out = tabs[node].submit(wirelib.unpack32(s.batch)) if not hasattr(tabs[node], "submit32") else tabs[node].submit32(s.batch)
                assert not np.any(((out.reply["flags"] >> abi.F_STATUS_SHIFT) & 0xFF) == abi.NEED_HOST)
                if then:
                    then(s, out)
            ing[node].recycle(s.bank)
            if not s.rows:
                assert ing[node].held() == 0
                return

    for tick in range(ticks):
        # 1. client commands
        cmds = abi.Batch(1, groups)
        for g in range(groups):
            if rng.random() < 0.6:
                cmds.put(0, g, abi.EV_CLIENT_APPEND, n=rng.randint(1, 3))
        if over_the_wire:
            for g in np.flatnonzero(cmds.head["hdr"] & 0xF):
                ing[0].add_row(2, int(g), int(cmds.head["hdr"][g]))
            decide_all(0)
        else:
            tabs[0].submit(cmds)
        # 2. the send side
        head, send = tabs[0].replicate(heartbeat=tick % 2)
        acks = abi.Batch(2, groups)                         # in-memory run: follower j's ack in round j
        for j in (0, 1):
            fol = j + 1
            if over_the_wire:
                data, frames, need = ing[0].encode_sends(j, 0, head, send[:, j], lambda g, i: TERM)
                assert need == 0 and frames == int(np.count_nonzero(send["kind"][:, j] == abi.SEND_APPEND))
                assert ing[fol].feed(0, data) == frames
                back = bytearray()
                decide_all(fol, lambda s, out: back.extend(ing[fol].emit(s.bank, out.reply, 0)))
                assert ing[0].feed(j, bytes(back)) == frames     # every response finds its invocation
            else:
                req = abi.Batch(1, groups)
                for g in range(groups):
                    sd = send[g, j]
                    assert int(sd["kind"]) in (abi.SEND_APPEND, abi.SEND_GATED, abi.SEND_NONE)
                    if int(sd["kind"]) == abi.SEND_APPEND:
                        req.put(0, g, abi.EV_AE_REQ, slot=0, a=int(head["term"][g]), b=int(sd["prev_index"]), c=int(sd["prev_term"]),
                                d=int(head["leader_commit"][g]), entries=[TERM] * int(sd["count"]))
                out = tabs[fol].submit(req)
                for g in range(groups):
                    if int(send[g, j]["kind"]) == abi.SEND_APPEND and int(out.reply["flags"][g]) & abi.F_REPLIED:
                        acks.put(j, g, abi.EV_AE_ACK, slot=fol, flag=int(bool(int(out.reply["flags"][g]) & abi.F_SUCCESS)), a=int(out.reply["resp_term"][g]),
                                 b=int(head["epoch_index"][g]), c=int(send[g, j]["last_index"]), aux=int(head["role_epoch"][g]))
        # 3. the acks at the leader
        if over_the_wire:
            decide_all(0)
        else:
            tabs[0].submit(acks)
    states = [t.read_state() for t in tabs]
    if over_the_wire:
        assert all(i.refused() == 0 for i in ing)
        for i in ing:
            i.close()
    return states


def slice_state(st, first, count):
    """the GroupState image of groups [first, first + count) of `st` (every term run kept): what a shard's table is loaded with"""
    F = st.followers
    total = int(np.sum(st.run_count[first:first + count]))
    packed = int(np.max(st.run_count[first:first + count], initial=0)) > abi.TERM_RUNS      # else the default layout: TERM_RUNS slots per group
    out = abi.GroupState(count, st.cluster, runs_total=max(total, 1)) if packed else abi.GroupState(count, st.cluster)
    for name, _, shape in abi._STATE_FIELDS:
        if shape == 1 and name != "run_offset":
            getattr(out, name)[:] = getattr(st, name)[first:first + count]
        elif shape == "peers":
            getattr(out, name)[:] = getattr(st, name)[first * F:(first + count) * F]
    at = 0
    for i in range(count):
        n, off = int(st.run_count[first + i]), int(st.run_offset[first + i])
        if not packed:
            at = i * abi.TERM_RUNS
        out.run_offset[i] = at
        out.run_start[at:at + n] = st.run_start[off:off + n]
        out.run_term[at:at + n] = st.run_term[off:off + n]
        at += n
    return out
