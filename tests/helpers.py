"""Scenario helpers shared by the oracle KATs (CPU) and the HIP parity tests (GPU).

A `Sim` drives ONE group of a table (oracle or HIP engine — both expose load_state / read_state /
submit) one event at a time, the way the reference EventLoop would feed a RaftContext.
"""
from types import SimpleNamespace

import numpy as np

from rafting_amd import abi

F, C, L = abi.FOLLOWER, abi.CANDIDATE, abi.LEADER


def make_state(cluster, count=1, **kw):
    """GroupState with every group set from keyword scalars:
    role, term, voted_for, leader, timeout_detected, repl_prepared, role_epoch, votes, elected_epoch,
    elected_term, commit, epoch=(index, term), log=(first, [(start, term)...], last),
    peers=[(last_epoch, next_index, match_index, rejection, pending), ...]"""
    st = abi.GroupState(count, cluster)
    for g in range(count):
        set_group(st, g, **kw)
    return st


def set_group(st, g, role=F, term=0, voted_for=abi.NO_NODE, leader=abi.NO_NODE, timeout_detected=0,
              repl_prepared=0, role_epoch=1, votes=1, elected_epoch=0, elected_term=0, commit=0,
              epoch=(0, 0), log=None, peers=None):
    st.role[g], st.current_term[g], st.voted_for[g], st.current_leader[g] = role, term, voted_for, leader
    st.timeout_detected[g], st.repl_prepared[g] = timeout_detected, repl_prepared
    st.role_epoch[g], st.votes[g] = role_epoch, votes
    st.elected_epoch[g], st.elected_term[g] = elected_epoch, elected_term
    st.commit_index[g] = commit
    st.epoch_index[g], st.epoch_term[g] = epoch
    if log is None:
        st.set_log(g, 0, [], 0)
    else:
        st.set_log(g, log[0], log[1], log[2])
    if peers is not None:
        assert len(peers) == st.followers
        for j, p in enumerate(peers):
            i = g * st.followers + j
            st.peer_last_epoch[i], st.peer_next_index[i], st.peer_match_index[i] = p[0], p[1], p[2]
            st.peer_rejection[i], st.peer_pending[i] = p[3], p[4]


def simple_log(last, term=1, first=1):
    """log with keys [first..last], all of one term"""
    return (first, [(first, term)], last)


class Sim:
    def __init__(self, table, group=0):
        self.t, self.g = table, group
        self.now = None                      # set: events carry this wall clock (Leadership.State statistics are kept)

    def load(self, **kw):
        st = make_state(self.t.cluster, self.t.groups, **kw)
        self.t.load_state(st)
        return self

    def state(self):
        st = self.t.read_state()
        g = self.g
        runs = [(int(st.run_start[g * abi.TERM_RUNS + k]), int(st.run_term[g * abi.TERM_RUNS + k]))
                for k in range(int(st.run_count[g]))]
        Fn = st.followers
        return SimpleNamespace(
            role=int(st.role[g]), term=int(st.current_term[g]), voted_for=int(st.voted_for[g]),
            leader=int(st.current_leader[g]), timeout_detected=int(st.timeout_detected[g]),
            repl_prepared=int(st.repl_prepared[g]), role_epoch=int(st.role_epoch[g]), votes=int(st.votes[g]),
            elected_epoch=int(st.elected_epoch[g]), elected_term=int(st.elected_term[g]),
            commit=int(st.commit_index[g]), epoch=(int(st.epoch_index[g]), int(st.epoch_term[g])),
            first=int(st.first_index[g]), last=int(st.last_index[g]), runs=runs,
            last_term=runs[-1][1] if runs else None,
            peers=[(int(st.peer_last_epoch[g * Fn + j]), int(st.peer_next_index[g * Fn + j]),
                    int(st.peer_match_index[g * Fn + j]), int(st.peer_rejection[g * Fn + j]),
                    int(st.peer_pending[g * Fn + j])) for j in range(Fn)])

    def event(self, kind, hint=None, **kw):
        b = abi.Batch(1, self.t.groups, hints=hint is not None)
        row = b.put(0, self.g, kind, **kw)
        if hint is not None:
            b.set_hint(row, *hint)
        out = self.t.submit(b, fill=0xAB) if self.now is None else self.t.submit_timed(b, [self.now], fill=0xAB)
        f = int(out.reply["flags"][row])
        return SimpleNamespace(
            flags=f, status=(f >> abi.F_STATUS_SHIFT) & 0xFF,
            replied=bool(f & abi.F_REPLIED), success=bool(f & abi.F_SUCCESS),
            resp_term=int(out.reply["resp_term"][row]), role_epoch=int(out.reply["role_epoch"][row]),
            role=(f & abi.F_ROLE_MASK) >> abi.F_ROLE_SHIFT, emit=(f & abi.F_EMIT_MASK) >> abi.F_EMIT_SHIFT,
            persist=bool(f & abi.F_PERSIST), role_changed=bool(f & abi.F_ROLE_CHANGED),
            reset_timer=bool(f & abi.F_RESET_TIMER), muted=bool(f & abi.F_TIMER_MUTED), commit_adv=bool(f & abi.F_COMMIT),
            truncated=bool(f & abi.F_LOG_TRUNC), appended=bool(f & abi.F_LOG_APPEND),
            commit=int(out.logfx["commit_index"][row]), log_from=int(out.logfx["log_from"][row]),
            p_term=int(out.persist["term"][row]), p_vote=int(out.persist["voted_for"][row]),
            p_role=int(out.persist["role"][row]))

    # the RaftParticipant surface (RaftParticipant.java:34-44) ---------------------------------
    def append_entries(self, term, leader, prev_index, prev_term, entries, leader_commit, hint=None):
        return self.event(abi.EV_AE_REQ, slot=leader, a=term, b=prev_index, c=prev_term, d=leader_commit,
                          entries=list(entries), hint=hint)

    def request_vote(self, term, cand, last_index, last_term):
        return self.event(abi.EV_RV_REQ, slot=cand, a=term, b=last_index, c=last_term)

    def pre_vote(self, term, cand, last_index, last_term):
        return self.event(abi.EV_PV_REQ, slot=cand, a=term, b=last_index, c=last_term)

    def on_timeout(self, ticket_epoch=0):
        return self.event(abi.EV_TIMEOUT, aux=ticket_epoch)

    def install_snapshot(self, term, leader, last_index, last_term, host_ok=True):
        return self.event(abi.EV_IS_REQ, slot=leader, flag=int(host_ok), a=term, b=last_index, c=last_term)

    # response callbacks -------------------------------------------------------------------------
    def ae_ack(self, peer, resp_term, success, epoch_at_send, last_sent, sent_epoch, hint=None):
        return self.event(abi.EV_AE_ACK, slot=peer, flag=int(success), a=resp_term, b=epoch_at_send,
                          c=last_sent, aux=sent_epoch, hint=hint)

    def is_ack(self, peer, resp_term, success, epoch_at_send, sent_epoch):
        return self.event(abi.EV_IS_ACK, slot=peer, flag=int(success), a=resp_term, b=epoch_at_send, aux=sent_epoch)

    def rv_reply(self, peer, resp_term, granted, sent_epoch):
        return self.event(abi.EV_RV_REPLY, slot=peer, flag=int(granted), a=resp_term, aux=sent_epoch)

    def pv_reply(self, peer, resp_term, granted, sent_epoch):
        return self.event(abi.EV_PV_REPLY, slot=peer, flag=int(granted), a=resp_term, aux=sent_epoch)

    def client_append(self, n):
        return self.event(abi.EV_CLIENT_APPEND, n=n)

    def log_flush(self, index, term):
        return self.event(abi.EV_LOG_FLUSH, a=index, b=term)


def compare_outcomes(ref, got, where=""):
    """Bit-exact comparison of two Outcome objects, honouring the 'valid iff flag' contract of the
    conditional fields. Raises AssertionError naming the first differing row."""
    rf, gf = ref.reply["flags"], got.reply["flags"]
    _same(rf, gf, "reply.flags", where)
    _same(ref.reply["role_epoch"], got.reply["role_epoch"], "reply.role_epoch", where)
    rep = (rf & abi.F_REPLIED) != 0
    _same(ref.reply["resp_term"][rep], got.reply["resp_term"][rep], "reply.resp_term", where)
    lf = (rf & (abi.F_COMMIT | abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) != 0
    _same(ref.logfx["commit_index"][lf], got.logfx["commit_index"][lf], "logfx.commit_index", where)
    lg = (rf & (abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) != 0
    _same(ref.logfx["log_from"][lg], got.logfx["log_from"][lg], "logfx.log_from", where)
    pf = (rf & abi.F_PERSIST) != 0
    for k in ("term", "voted_for", "role"):
        _same(ref.persist[k][pf], got.persist[k][pf], "persist." + k, where)


def check_out32_rows(raw, unpacked, before, after, rounds, count):
    """What include/raftgpu.h promises about rg_out32_t / rg_persist32_t rows beyond what rg_outcome32_unpack carries over: commit_index is
    RaftLog.lastCommitted() after EVERY row (it moves exactly where RG_F_COMMIT says, and ends at the table's value), resp_term is 0 unless
    RG_F_REPLIED, a persist row names the role its reply names, and the role epochs end at the table's. Rows flagged RG_F_WIDE_VALUES
    (the 64-bit body met a value beyond int32) are exempt from the numeric checks: their truth is in the overflow columns."""
    fl = raw.row["flags"].reshape(rounds, count)
    wide = (fl & abi.F_WIDE_VALUES) != 0
    commit = raw.row["commit_index"].astype(np.int64).reshape(rounds, count)
    prev = np.vstack([before.commit_index[None, :], commit[:-1]])
    moved = (fl & abi.F_COMMIT) != 0
    ok = wide | np.vstack([np.zeros((1, count), bool), wide[:-1]]) | np.where(moved, commit > prev, commit == prev)
    if not ok.all():
        r, g = np.argwhere(~ok)[0]
        raise AssertionError("rg_out32_t.commit_index of round %d group %d is %d after %d (RG_F_COMMIT %s)" % (r, g, commit[r, g], prev[r, g], bool(moved[r, g])))
    small = (after.commit_index < (1 << 31)) & ~wide[-1]
    assert np.array_equal(commit[-1][small], after.commit_index[small]), "the last rows' commit_index is not the table's"
    silent = ((fl & abi.F_REPLIED) == 0) & ~wide
    assert not raw.row["resp_term"].reshape(rounds, count)[silent].any(), "resp_term of a row without RG_F_REPLIED"
    per = (fl & abi.F_PERSIST) != 0
    assert np.array_equal(raw.persist["role"].reshape(rounds, count)[per].astype(np.uint32), abi.flags_role(fl[per])), "persist.role != the reply's role"
    ep = before.role_epoch.copy()
    pe = raw.persist["role_epoch"].reshape(rounds, count)
    wr = None if raw.wide is None else raw.wide.reply["role_epoch"].reshape(rounds, count)
    for r in range(rounds):
        ep = np.where(per[r], pe[r], ep)
        if wr is not None:
            ep = np.where(wide[r], wr[r], ep)
    assert np.array_equal(ep, after.role_epoch), "role epochs carried by the persist rows do not end at the table's"
    _same(unpacked.reply["role_epoch"].reshape(rounds, count)[-1], after.role_epoch, "unpacked role_epoch (last round)", "rg_outcome32_unpack")


def _same(a, b, what, where):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b):
        bad = np.flatnonzero(a != b)
        i = int(bad[0]) if len(bad) else -1
        raise AssertionError("%s %s differs at %d of %d rows; first row %d: ref=%r got=%r" % (
            where, what, len(bad), len(a), i, a[i] if i >= 0 else None, b[i] if i >= 0 else None))


def canonical_state(st):
    """Zero the fields of a GroupState image that name an object which does not exist in the reference at that moment:
    the vote counter outside a running (pre-)election (`AtomicInteger votes` is a local of startElection / prepareElection),
    the winner's term without a live winner head, and the Leadership.State columns of a group that has not run
    prepareReplication.  The C-ABI keeps stale values there; the reference has nothing to compare them with."""
    live = (st.role == C) | ((st.role == F) & (st.timeout_detected != 0))
    st.votes[~live] = 1
    st.elected_term[st.elected_epoch == 0] = 0
    dead = np.repeat(st.repl_prepared == 0, st.followers)
    for nm in ("peer_last_epoch", "peer_next_index", "peer_match_index", "peer_rejection", "peer_pending"):
        getattr(st, nm)[dead] = 0
    return st


def compare_states(ref, got, where=""):
    """Bit-exact comparison of two GroupState images. Peer columns only matter once
    repl_prepared; the device run cache must be a suffix of the oracle's run list."""
    for name in ("current_term", "voted_for", "role", "current_leader", "timeout_detected", "repl_prepared",
                 "role_epoch", "votes", "elected_epoch", "elected_term", "commit_index", "epoch_index",
                 "epoch_term", "first_index", "last_index",
                 "peer_last_epoch", "peer_next_index", "peer_match_index", "peer_rejection", "peer_pending"):
        _same(getattr(ref, name), getattr(got, name), "state." + name, where)
    K = abi.TERM_RUNS
    rc_r, rc_g = ref.run_count.astype(np.int64), got.run_count.astype(np.int64)
    if np.any(rc_g > rc_r) or np.any((rc_r > 0) != (rc_g > 0)):
        raise AssertionError(where + " state.run_count: device cache is not a suffix of the oracle runs")
    rs_r, rt_r = ref.run_start.reshape(-1, K), ref.run_term.reshape(-1, K)
    rs_g, rt_g = got.run_start.reshape(-1, K), got.run_term.reshape(-1, K)
    for k in range(1, K + 1):      # k-th newest run
        have = rc_g >= k
        idx = np.flatnonzero(have)
        tr, tg = rt_r[idx, rc_r[idx] - k], rt_g[idx, rc_g[idx] - k]
        _same(tr, tg, "state.run_term[-%d]" % k, where)
        sr, sg = rs_r[idx, rc_r[idx] - k], rs_g[idx, rc_g[idx] - k]
        oldest = rc_g[idx] == k    # the device's oldest cached run may start later than the true run (hint-rebuilt cache)
        if np.any(sg[~oldest] != sr[~oldest]) or np.any(sg[oldest] < sr[oldest]):
            raise AssertionError(where + " state.run_start[-%d] differs" % k)
