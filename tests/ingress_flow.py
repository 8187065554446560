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


def history(groups, cluster, self_slot, pre_vote, rounds, seed, view=None):
    """(initial state, [one-round batches], [the oracle's outcomes], final state). view: a table of the kind under test (engine.Table), decided
    in lockstep — the fuzzer then draws its rows from THAT table's state image, whose cached term runs are the device's own (no row of the
    history leaves them: RG_NEED_HOST round trips are the host's business and have their own tests), and its outcomes are held to the oracle's."""
    from tests.helpers import compare_outcomes
    st0 = fuzz.random_initial_state(groups, cluster, self_slot, seed)
    for g in range(5, groups, 23):                    # a few groups whose terms lie beyond int32: their rows cannot be compact rows
        st0.current_term[g] += 1 << 33
    orc = oracle_lib.OracleTable(groups, cluster, self_slot, pre_vote)
    orc.load_state(st0)
    if view is not None:
        view.load_state(st0)
    fz = fuzz.Fuzzer(groups, cluster, self_slot, seed, allow_miss=False)
    batches, outs = [], []
    for r in range(rounds):
        b = abi.Batch(1, groups)
        fz.round((view or orc).read_state(), b, 0)
        hdr = b.head["hdr"]
        cannot_travel = ((hdr & 0xF) == abi.EV_AE_REQ) & ((((hdr >> 4) & 0xF) >= cluster) | ((hdr >> 12) > abi.MAX_AE_ENTRIES))    # no NodeID / refused frame
        cannot_travel |= (hdr & abi.HDR_HINT_BIT) != 0
        b.head[cannot_travel] = (0, 0)
        outs.append(orc.submit(b))
        if view is not None:
            compare_outcomes(outs[-1], view.submit(b), "history round %d" % r)
        batches.append(b)
    return st0, batches, outs, orc.read_state()


def drive(table_decide32, table_decide_sparse, groups, cluster, batches, outs, max_rounds, nodes):
    """Feeds the history, seals until everything was decided. table_decide32(abi.Batch32) and table_decide_sparse(abi.Batch with gid) return
    objects with .reply (REPLY_DT rows). Returns the number of batches sealed."""
    ctx = [b"group-%05d" % g for g in range(groups)]
    nodes_b = wirelib.nodes_arg(nodes)
    ing = wirelib.Ingress(groups, max_rounds, LOCAL_CONN + 1, nodes=nodes, entry_cap=1 << 18)
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
    while True:
        s = ing.seal()
        if s.rows == 0 and not s.wide:
            assert ing.held() == 0
            ing.recycle(s.bank)
            break
        sealed += 1
        G = groups
        if s.batch.rounds:
            reply = table_decide32(s.batch).reply
            for r in range(s.batch.rounds):
                kinds = s.batch.head["hdr"][r * G:(r + 1) * G] & 0xF
                for g in np.flatnonzero(kinds):
                    kind, rep, key = expected[g][seen[g]]
                    assert kind == kinds[g], (g, seen[g], kind, kinds[g])
                    got = reply[r * G + g]
                    assert (int(got["flags"]), int(got["role_epoch"])) == (int(rep["flags"]), int(rep["role_epoch"])), (g, seen[g], kind)
                    if int(rep["flags"]) & abi.F_REPLIED:
                        assert int(got["resp_term"]) == int(rep["resp_term"])
                    assert ing.origin(s.bank, r * G + g) == key
                    seen[g] += 1
            for conn in range(LOCAL_CONN + 1):
                for ftype, sq, head, body in wirelib.split_frames(ing.emit(s.bank, reply, conn)):
                    assert ftype == wirelib.ACK and (conn, sq) not in got_answers
                    got_answers[(conn, sq)] = wirelib.decode_response(body)
                    kind, rep, key = next(e for e in expected[int(head.split(b"-")[1])] if e[2] == (conn, sq))
                    assert head == wirelib.METHOD_NAME[REQ_METHOD[kind]] + b":" + ctx[int(head.split(b"-")[1])]
        if s.wide:                                     # rows the compact format cannot hold: one sparse round after the batch
            sp = abi.Batch(1, len(s.wide), gid=np.array([w[0] for w in s.wide], dtype=np.uint32))
            for i, (g, hdr, aux, q, terms, origin) in enumerate(s.wide):
                sp.put(0, i, hdr & 0xF, slot=(hdr >> 4) & 0xF, flag=(hdr >> 8) & 1, a=q[0], b=q[1], c=q[2], d=q[3], aux=aux, entries=terms or None, n=hdr >> 12)
            reply = table_decide_sparse(sp).reply
            for i, (g, hdr, aux, q, terms, origin) in enumerate(s.wide):
                kind, rep, key = expected[g][seen[g]]
                assert kind == hdr & 0xF and origin == key
                assert (int(reply[i]["flags"]), int(reply[i]["role_epoch"])) == (int(rep["flags"]), int(rep["role_epoch"])), (g, seen[g], kind, "wide")
                if key is not None and int(rep["flags"]) & abi.F_REPLIED:
                    got_answers[key] = (int(reply[i]["resp_term"]), bool(int(reply[i]["flags"]) & abi.F_SUCCESS))
                seen[g] += 1
        ing.recycle(s.bank)
    assert seen == [len(e) for e in expected] and sum(seen) == queued
    assert got_answers == answers
    ing.close()
    return sealed
