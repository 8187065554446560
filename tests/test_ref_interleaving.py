"""The interleaving differential: the reference's own code under the ONE concurrency its callbacks have, against the oracle under
the two serial orders a host of the batched engine can produce.

In the reference a response callback runs on a transport thread: its off-loop half CASes the context's membership filter
(context/RaftRoutine.java:140-151) and queues the conversion at the HEAD of the context's event loop (context/RaftContext.java:205-215,
support/EventLoop.java:87-101). oracle/ref_shim/ref_driver.cpp normally drains the loop right after every row — the interleaving in which
nothing else of the group runs in between. `RefTable.submit_held(first, then)` is the other one: `first` (the callback) is delivered with
the loop left undrained, `then` runs its handler — the loop thread was already inside that task when the callback's thread came by — and only
then the loop is drained, the held urgent task first. For every callback family (member/Leader.java:174-188,218-237 replication responses,
member/Candidate.java:121-134 election replies, member/Follower.java:258-270 pre-election replies) and a grid of second events, the final
group state and the second event's reply are compared with the ORACLE's after `first; then` ("A;B") and after `then; first` ("B;A").

EXPECTED pins the classification: (orders that reproduce state + reply, orders that also reproduce the role epoch — the count of participant
objects, i.e. of (term, votedFor) fsyncs —, the status the reference's own run reports for the second event). What it says, in short
(INTEGRATION.md section 1 "ordering contract"):
  * a callback that only updates replication bookkeeping / commits commutes with everything (`leader_ack_commits`: both orders);
  * a callback that converts (higher term seen, election / pre-election won) is reproduced by A;B whenever the second event's own term
    arbitration loses to the callback's membership, by B;A whenever it wins or the handler never looks at the filter; where both hold the
    interleaving additionally SKIPS one participant object (role epoch of B;A);
  * status 21 (RG_A_NO_DOWNGRADE, context/RaftRoutine.java:170-172) is the reference logging an AssertionError from the held urgent task after the
    handler already applied the pending membership: state and reply are A;B's — the one way that "unreachable" assertion is reached;
  * NEITHER lists the cases no serial order reproduces: an un-fenced timeout (`aux` = 0 stands for "whoever is current", which a serial engine
    resolves AFTER the conversion and the reference's timer thread BEFORE it) and AppendEntries / PreVote of the SAME term landing between a
    Candidate's winning CAS and its conversion to Leader (the old Candidate object answers, then the conversion is refused or overtaken).
    The engine's contract for those: rows are decided in row order, one participant at a time — the host must hand a callback's row to a flush
    before any request it dequeues later, and fence timeouts with the role epoch (`aux` != 0), which makes the timeout cases A;B."""
import numpy as np
import pytest

from rafting_amd import abi
from tests import oracle_lib, ref_lib
from tests.helpers import C, F, L, canonical_state, make_state

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libref.so needs the reference checkout to be built")

LOG = (1, [(1, 4), (51, 5)], 100)           # last index 100, last term 5


def leader_state():
    return dict(role=L, term=5, voted_for=0, role_epoch=3, repl_prepared=1, log=LOG, peers=[(0, 101, 0, 0, 0)] * 4, commit=40)


def candidate_state():
    return dict(role=C, term=6, voted_for=0, role_epoch=7, votes=2, log=LOG, commit=40)


def prevote_state():
    return dict(role=F, term=5, voted_for=2, role_epoch=4, votes=2, timeout_detected=1, log=LOG, commit=40)


def ev(kind, **kw):
    return (kind, kw)


FAMILIES = {        # callback family -> (state, the callback A)
    "leader_ack_higher_term": (leader_state, ev(abi.EV_AE_ACK, slot=2, flag=0, a=8, b=0, c=100, aux=3)),        # member/Leader.java:224-226
    "leader_snap_ack_higher": (leader_state, ev(abi.EV_IS_ACK, slot=2, flag=1, a=8, b=0, aux=3)),               # member/Leader.java:178-181
    "leader_ack_commits": (leader_state, ev(abi.EV_AE_ACK, slot=2, flag=1, a=5, b=0, c=100, aux=3)),            # member/Leader.java:228-237, 247-280
    "candidate_wins": (candidate_state, ev(abi.EV_RV_REPLY, slot=1, flag=1, a=6, aux=7)),                       # member/Candidate.java:127-131
    "candidate_reply_higher": (candidate_state, ev(abi.EV_RV_REPLY, slot=1, flag=0, a=9, aux=7)),               # member/Candidate.java:124-126
    "prevote_wins": (prevote_state, ev(abi.EV_PV_REPLY, slot=1, flag=1, a=5, aux=4)),                           # member/Follower.java:263-267
    "prevote_reply_higher": (prevote_state, ev(abi.EV_PV_REPLY, slot=1, flag=0, a=9, aux=4)),                   # member/Follower.java:260-262
}


def seconds(family):
    st = FAMILIES[family][0]()
    t, ep = st["term"], st["role_epoch"]
    out = {
        "ae_same_term": ev(abi.EV_AE_REQ, slot=3, a=t, b=100, c=5, d=60, entries=[]),
        "ae_term_plus1": ev(abi.EV_AE_REQ, slot=3, a=t + 1, b=100, c=5, d=60, entries=[t + 1]),
        "ae_term_plus2": ev(abi.EV_AE_REQ, slot=3, a=t + 2, b=100, c=5, d=60, entries=[t + 2]),
        "ae_term_plus9": ev(abi.EV_AE_REQ, slot=3, a=t + 9, b=100, c=5, d=60, entries=[t + 9]),
        "rv_term_plus1": ev(abi.EV_RV_REQ, slot=4, a=t + 1, b=100, c=5),
        "rv_term_plus9": ev(abi.EV_RV_REQ, slot=4, a=t + 9, b=100, c=5),
        "pv_term_plus1": ev(abi.EV_PV_REQ, slot=4, a=t + 1, b=100, c=5),
        "timeout_any": ev(abi.EV_TIMEOUT, aux=0),
        "timeout_fenced": ev(abi.EV_TIMEOUT, aux=ep),
        "client_append": ev(abi.EV_CLIENT_APPEND, n=2),
    }
    reply = {L: abi.EV_AE_ACK, C: abi.EV_RV_REPLY, F: abi.EV_PV_REPLY}[st["role"]]
    if st["role"] == L:
        out["second_ack_ok"] = ev(reply, slot=1, flag=1, a=t, b=0, c=100, aux=ep)
        out["second_ack_higher"] = ev(reply, slot=1, flag=0, a=t + 1, b=0, c=100, aux=ep)
    else:
        out["second_reply_grant"] = ev(reply, slot=2, flag=1, a=t, aux=ep)
        out["second_reply_higher"] = ev(reply, slot=2, flag=0, a=t + 2, aux=ep)
    return out


EXPECTED = {
    ('leader_ack_higher_term', 'ae_same_term'): ('B;A', 'B;A', 8),
    ('leader_ack_higher_term', 'ae_term_plus1'): ('A;B', 'A;B', 21),
    ('leader_ack_higher_term', 'ae_term_plus2'): ('A;B', 'A;B', 21),
    ('leader_ack_higher_term', 'ae_term_plus9'): ('A;B', 'A;B', 0),
    ('leader_ack_higher_term', 'rv_term_plus1'): ('A;B', 'A;B', 21),
    ('leader_ack_higher_term', 'rv_term_plus9'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_higher_term', 'pv_term_plus1'): ('B;A', 'B;A', 0),
    ('leader_ack_higher_term', 'timeout_any'): ('B;A', 'B;A', 0),
    ('leader_ack_higher_term', 'timeout_fenced'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_higher_term', 'client_append'): ('A;B', 'A;B', 0),
    ('leader_ack_higher_term', 'second_ack_ok'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_higher_term', 'second_ack_higher'): ('A;B', 'A;B', 0),
    ('leader_snap_ack_higher', 'ae_same_term'): ('B;A', 'B;A', 8),
    ('leader_snap_ack_higher', 'ae_term_plus1'): ('A;B', 'A;B', 21),
    ('leader_snap_ack_higher', 'ae_term_plus2'): ('A;B', 'A;B', 21),
    ('leader_snap_ack_higher', 'ae_term_plus9'): ('A;B', 'A;B', 0),
    ('leader_snap_ack_higher', 'rv_term_plus1'): ('A;B', 'A;B', 21),
    ('leader_snap_ack_higher', 'rv_term_plus9'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_snap_ack_higher', 'pv_term_plus1'): ('B;A', 'B;A', 0),
    ('leader_snap_ack_higher', 'timeout_any'): ('B;A', 'B;A', 0),
    ('leader_snap_ack_higher', 'timeout_fenced'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_snap_ack_higher', 'client_append'): ('A;B', 'A;B', 0),
    ('leader_snap_ack_higher', 'second_ack_ok'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_snap_ack_higher', 'second_ack_higher'): ('A;B', 'A;B', 0),
    ('leader_ack_commits', 'ae_same_term'): ('A;B|B;A', 'A;B|B;A', 8),
    ('leader_ack_commits', 'ae_term_plus1'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'ae_term_plus2'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'ae_term_plus9'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'rv_term_plus1'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'rv_term_plus9'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'pv_term_plus1'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'timeout_any'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'timeout_fenced'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'client_append'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'second_ack_ok'): ('A;B|B;A', 'A;B|B;A', 0),
    ('leader_ack_commits', 'second_ack_higher'): ('A;B|B;A', 'A;B|B;A', 0),
    ('candidate_wins', 'ae_same_term'): ('neither', 'neither', 12),
    ('candidate_wins', 'ae_term_plus1'): ('A;B', 'neither', 0),
    ('candidate_wins', 'ae_term_plus2'): ('A;B', 'neither', 0),
    ('candidate_wins', 'ae_term_plus9'): ('A;B', 'neither', 0),
    ('candidate_wins', 'rv_term_plus1'): ('A;B', 'neither', 0),
    ('candidate_wins', 'rv_term_plus9'): ('A;B', 'neither', 0),
    ('candidate_wins', 'pv_term_plus1'): ('neither', 'neither', 0),
    ('candidate_wins', 'timeout_any'): ('neither', 'neither', 0),
    ('candidate_wins', 'timeout_fenced'): ('A;B', 'A;B', 0),
    ('candidate_wins', 'client_append'): ('B;A', 'B;A', 18),
    ('candidate_wins', 'second_reply_grant'): ('A;B|B;A', 'A;B|B;A', 0),
    ('candidate_wins', 'second_reply_higher'): ('A;B|B;A', 'B;A', 0),
    ('candidate_reply_higher', 'ae_same_term'): ('A;B', 'A;B', 21),
    ('candidate_reply_higher', 'ae_term_plus1'): ('A;B', 'A;B', 21),
    ('candidate_reply_higher', 'ae_term_plus2'): ('A;B', 'A;B', 21),
    ('candidate_reply_higher', 'ae_term_plus9'): ('B;A', 'B;A', 0),
    ('candidate_reply_higher', 'rv_term_plus1'): ('A;B', 'A;B', 21),
    ('candidate_reply_higher', 'rv_term_plus9'): ('A;B|B;A', 'B;A', 0),
    ('candidate_reply_higher', 'pv_term_plus1'): ('A;B', 'A;B', 21),
    ('candidate_reply_higher', 'timeout_any'): ('neither', 'neither', 0),
    ('candidate_reply_higher', 'timeout_fenced'): ('A;B', 'A;B', 0),
    ('candidate_reply_higher', 'client_append'): ('A;B|B;A', 'A;B|B;A', 18),
    ('candidate_reply_higher', 'second_reply_grant'): ('A;B|B;A', 'A;B', 17),
    ('candidate_reply_higher', 'second_reply_higher'): ('A;B', 'A;B', 17),
    ('prevote_wins', 'ae_same_term'): ('A;B', 'A;B', 0),
    ('prevote_wins', 'ae_term_plus1'): ('B;A', 'B;A', 0),
    ('prevote_wins', 'ae_term_plus2'): ('B;A', 'B;A', 0),
    ('prevote_wins', 'ae_term_plus9'): ('B;A', 'B;A', 0),
    ('prevote_wins', 'rv_term_plus1'): ('B;A', 'B;A', 0),
    ('prevote_wins', 'rv_term_plus9'): ('A;B|B;A', 'B;A', 0),
    ('prevote_wins', 'pv_term_plus1'): ('B;A', 'B;A', 0),
    ('prevote_wins', 'timeout_any'): ('neither', 'neither', 0),
    ('prevote_wins', 'timeout_fenced'): ('A;B', 'A;B', 0),
    ('prevote_wins', 'client_append'): ('A;B|B;A', 'A;B|B;A', 18),
    ('prevote_wins', 'second_reply_grant'): ('A;B|B;A', 'A;B|B;A', 0),
    ('prevote_wins', 'second_reply_higher'): ('B;A', 'B;A', 0),
    ('prevote_reply_higher', 'ae_same_term'): ('A;B', 'A;B', 21),
    ('prevote_reply_higher', 'ae_term_plus1'): ('A;B', 'A;B', 21),
    ('prevote_reply_higher', 'ae_term_plus2'): ('A;B', 'A;B', 21),
    ('prevote_reply_higher', 'ae_term_plus9'): ('B;A', 'B;A', 0),
    ('prevote_reply_higher', 'rv_term_plus1'): ('A;B', 'A;B', 21),
    ('prevote_reply_higher', 'rv_term_plus9'): ('A;B|B;A', 'B;A', 0),
    ('prevote_reply_higher', 'pv_term_plus1'): ('B;A', 'B;A', 0),
    ('prevote_reply_higher', 'timeout_any'): ('neither', 'neither', 0),
    ('prevote_reply_higher', 'timeout_fenced'): ('A;B', 'A;B', 0),
    ('prevote_reply_higher', 'client_append'): ('A;B|B;A', 'A;B|B;A', 18),
    ('prevote_reply_higher', 'second_reply_grant'): ('A;B', 'A;B', 17),
    ('prevote_reply_higher', 'second_reply_higher'): ('A;B', 'A;B', 17),
}

NEITHER = {k for k, v in EXPECTED.items() if v[0] == "neither"}


def _batch(e):
    b = abi.Batch(1, 1)
    b.put(0, 0, e[0], **e[1])
    return b


def _state_key(st, with_epoch):
    st = canonical_state(st)
    names = ["current_term", "voted_for", "role", "current_leader", "timeout_detected", "repl_prepared", "votes", "elected_epoch", "elected_term",
             "commit_index", "epoch_index", "epoch_term", "first_index", "last_index", "peer_last_epoch", "peer_next_index", "peer_match_index",
             "peer_rejection", "peer_pending", "run_count"] + (["role_epoch"] if with_epoch else [])
    rc = int(st.run_count[0])
    return tuple(tuple(np.asarray(getattr(st, n)).tolist()) for n in names) + (tuple(st.run_start[:rc].tolist()), tuple(st.run_term[:rc].tolist()))


def _reply_key(out):
    f = int(out.reply["flags"][0])
    rep = bool(f & abi.F_REPLIED)
    return (rep, bool(f & abi.F_SUCCESS) if rep else None, int(out.reply["resp_term"][0]) if rep else None)


def _status(out):
    return (int(out.reply["flags"][0]) >> abi.F_STATUS_SHIFT) & 0xFF


def _run(mk, state, first, then, held=False):
    t = mk(1, 5, 0, True)
    t.load_state(make_state(5, 1, **state))
    if held:
        o1, o2 = t.submit_held(_batch(first), _batch(then))
    else:
        o1, o2 = t.submit(_batch(first)), t.submit(_batch(then))
    st = t.read_state()
    t.close()
    return st, o1, o2


def _classify(family, second):
    mk_state, A = FAMILIES[family]
    B = seconds(family)[second]
    st_h, _, hb = _run(ref_lib.RefTable, mk_state(), A, B, held=True)
    st_ab, _, ab_b = _run(oracle_lib.OracleTable, mk_state(), A, B)
    st_ba, ba_b, _ = _run(oracle_lib.OracleTable, mk_state(), B, A)
    st_ref, _, ref_b = _run(ref_lib.RefTable, mk_state(), A, B)          # and the reference itself, drained after every row: the oracle's A;B
    assert _state_key(st_ref, True) == _state_key(st_ab, True) and _reply_key(ref_b) == _reply_key(ab_b) and _status(ref_b) == _status(ab_b)
    orders, with_epoch = [], []
    for tag, st, rb in (("A;B", st_ab, ab_b), ("B;A", st_ba, ba_b)):
        if _state_key(st_h, False) == _state_key(st, False) and _reply_key(hb) == _reply_key(rb):
            orders.append(tag)
            if int(st_h.role_epoch[0]) == int(st.role_epoch[0]):
                with_epoch.append(tag)
    return ("|".join(orders) or "neither", "|".join(with_epoch) or "neither", _status(hb))


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_a_callback_that_overtakes_a_running_handler_is_one_of_the_two_serial_orders(family):
    got = {(family, s): _classify(family, s) for s in seconds(family)}
    want = {k: v for k, v in EXPECTED.items() if k[0] == family}
    assert got == want, "\n".join("%s: got %s, pinned %s" % (k, got[k], want.get(k)) for k in sorted(got) if got[k] != want.get(k))


def test_what_no_serial_order_reproduces_is_what_the_ordering_contract_names():
    """un-fenced timeouts, and same-term requests between a Candidate's winning CAS and its conversion"""
    assert NEITHER == {("candidate_wins", "ae_same_term"), ("candidate_wins", "pv_term_plus1"), ("candidate_wins", "timeout_any"),
                       ("candidate_reply_higher", "timeout_any"), ("prevote_wins", "timeout_any"), ("prevote_reply_higher", "timeout_any")}
    assert all(v[2] in (abi.OK, abi.DROPPED_STALE_ROLE, abi.NOT_LEADER, abi.A_NO_DOWNGRADE, abi.A_SAME_TERM_LEADER, abi.A_LEADER_UNCHANGED) for v in EXPECTED.values())
    # with the fence every timeout case is A;B (state, reply and role epoch)
    assert all(EXPECTED[(f, "timeout_fenced")][1] in ("A;B", "A;B|B;A") for f in FAMILIES)
