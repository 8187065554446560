"""Known-answer scenarios for the hot path, hand-derived from the reference SOURCE (the reference has
no test that pins this path — SURVEY.md §4/§8c).  Each scenario names the reference lines the
expected values were read from (paths relative to .../io/lubricant/consensus/raft/).

Scenarios are backend-agnostic: `mk(groups, cluster, self_slot, pre_vote)` returns a table exposing
load_state/read_state/submit — the CPU oracle in tests/test_oracle_kat.py, the HIP engine through
the C-ABI in tests/test_gpu_parity.py.
"""
from rafting_amd import abi
from tests.helpers import C, F, L, Sim, simple_log

NO = abi.NO_NODE


def _sim(mk, cluster=3, self_slot=0, pre_vote=True, **state):
    return Sim(mk(1, cluster, self_slot, pre_vote)).load(**state)


# --------------------------------------------------------------------------------------------------
# a1: Follower.appendEntries  member/Follower.java:35-88 (+logContains :177-191, purgeEntries :209-221)

def ae_stale_term(mk):
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5))
    r = s.append_entries(4, 1, 10, 5, [], 0)                  # :39-41
    assert (r.status, r.replied, r.success, r.resp_term) == (abi.OK, True, False, 5)
    assert not r.reset_timer and not r.role_changed
    assert s.state().leader == NO


def ae_heartbeat_commit(mk):
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5), commit=3)
    r = s.append_entries(5, 1, 10, 5, [], 8)                  # :57 match, :76-82 commit min(8,10)
    assert (r.status, r.replied, r.success, r.resp_term) == (abi.OK, True, True, 5)
    assert r.commit_adv and r.commit == 8 and r.reset_timer and not r.role_changed and not r.appended
    st = s.state()
    assert (st.commit, st.leader, st.role_epoch) == (8, 1, 1)
    r = s.append_entries(5, 1, 10, 5, [], 99)                 # commit truncated to last.index (:80)
    assert r.commit_adv and r.commit == 10
    r = s.append_entries(5, 1, 10, 5, [], 10)                 # markCommitted(==) -> false, no flag
    assert r.success and not r.commit_adv


def ae_prev_mismatch(mk):
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5))
    r = s.append_entries(5, 1, 10, 4, [], 0)                  # entry.term()!=term :190
    assert (r.status, r.replied, r.success, r.resp_term) == (abi.OK, True, False, 5)
    assert s.state().leader == 1                              # :54 ran before :57
    r = s.append_entries(5, 1, 11, 5, [], 0)                  # get(11)==null :190
    assert (r.replied, r.success) == (True, False)


def ae_term_bump_keeps_vote(mk):                              # Q1, Q3
    s = _sim(mk, role=F, term=5, voted_for=2, leader=2)
    r = s.append_entries(7, 1, 0, 0, [], 0)                   # :45-47 switchTo(Follower, term, lastCandidate)
    assert (r.status, r.replied, r.success, r.resp_term) == (abi.OK, True, True, 7)
    assert r.persist and r.role_changed and (r.p_term, r.p_vote, r.p_role) == (7, 2, F)
    st = s.state()
    assert (st.term, st.voted_for, st.leader, st.role_epoch) == (7, 2, 1, 2)
    assert r.role_epoch == 2 and not r.commit_adv


def ae_after_own_timeout(mk):                                 # Q2: same-term Follower->Follower refresh
    s = _sim(mk, role=F, term=5, voted_for=1, leader=2, timeout_detected=1, votes=2)
    r = s.append_entries(5, 1, 0, 0, [], 0)                   # :45 timeoutDetected branch: no two-leader check
    assert (r.status, r.success, r.resp_term, r.role_changed, r.persist) == (abi.OK, True, 5, True, True)
    st = s.state()
    assert (st.timeout_detected, st.leader, st.role_epoch, st.votes, st.voted_for) == (0, 1, 2, 1, 1)


def ae_two_leaders(mk):                                       # Q12
    s = _sim(mk, role=F, term=5, leader=2)
    r = s.append_entries(5, 1, 0, 0, [], 0)                   # :48-50
    assert r.status == abi.A_TWO_LEADERS and not r.replied and r.reset_timer
    assert s.state().leader == 2


def ae_at_candidate(mk):
    s = _sim(mk, role=C, term=5, voted_for=0, role_epoch=4)
    r = s.append_entries(4, 1, 0, 0, [], 0)                   # Candidate.java:32-34
    assert (r.replied, r.success, r.resp_term, r.role) == (True, False, 5, C)
    r = s.append_entries(5, 1, 0, 0, [], 0)                   # :39 same term: Follower beats Candidate (Membership.java:92)
    assert (r.status, r.success, r.resp_term, r.role, r.role_epoch) == (abi.OK, True, 5, F, 5)
    st = s.state()
    assert (st.role, st.term, st.voted_for, st.leader) == (F, 5, 0, 1)


def ae_at_leader(mk):                                         # Q4
    s = _sim(mk, role=L, term=5, voted_for=0, role_epoch=4)
    assert s.append_entries(9, 0, 0, 0, [], 0).status == abi.A_LEADER_SELF_AE       # Leader.java:71-73
    r = s.append_entries(4, 1, 0, 0, [], 0)                                        # :75-77
    assert (r.replied, r.success, r.resp_term) == (True, False, 5)
    assert s.append_entries(5, 1, 0, 0, [], 0).status == abi.A_SAME_TERM_LEADER    # :79-81
    assert s.state().role == L
    r = s.append_entries(8, 1, 0, 0, [], 0)                   # :84-85 then Follower.java:45-47: two conversions
    assert (r.status, r.success, r.resp_term, r.role, r.role_epoch) == (abi.OK, True, 8, F, 6)
    assert (r.p_term, r.p_vote, r.p_role) == (8, 0, F)
    st = s.state()
    assert (st.role, st.term, st.voted_for, st.leader, st.role_epoch) == (F, 8, 0, 1, 6)


def ae_log_contains_quirks(mk):                               # Q9
    s = _sim(mk, role=F, term=5, epoch=(5, 3), log=(6, [(6, 3)], 8))
    assert s.append_entries(5, 1, 0, 0, [], 0).success                              # :178
    r = s.append_entries(5, 1, 0, 5, [], 0)
    assert r.status == abi.A_PREV_ZERO_MISMATCH and not r.replied                   # :179-181
    assert s.append_entries(5, 1, 5, 0, [], 0).status == abi.A_PREV_ZERO_MISMATCH
    assert s.append_entries(5, 1, 3, 99, [], 0).success                             # index<epoch: term not checked :183-187
    assert s.append_entries(5, 1, 5, 3, [], 0).success
    assert s.append_entries(5, 1, 5, 4, [], 0).status == abi.A_EPOCH_TERM_MISMATCH  # :184-186
    assert s.state().leader == 1


def ae_append(mk):
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5), commit=3)
    r = s.append_entries(5, 1, 10, 5, [5, 5], 11)             # :68-74 append; :80 min(11, 12)
    assert (r.status, r.success, r.appended, r.truncated, r.log_from) == (abi.OK, True, True, False, 11)
    assert r.commit_adv and r.commit == 11
    st = s.state()
    assert (st.last, st.runs, st.commit) == (12, [(1, 5)], 11)
    r = s.append_entries(6, 1, 12, 5, [6], 11)                # new term run
    assert r.success and r.log_from == 13
    st = s.state()
    assert (st.last, st.runs, st.term) == (13, [(1, 5), (13, 6)], 6)


def ae_append_on_empty_log(mk):
    s = _sim(mk, role=F, term=1)
    r = s.append_entries(1, 2, 0, 0, [1, 1, 1], 2)
    assert (r.success, r.appended, r.log_from, r.commit_adv, r.commit) == (True, True, 1, True, 2)
    st = s.state()
    assert (st.first, st.last, st.runs) == (1, 3, [(1, 1)])


def ae_conflict_truncates(mk):
    s = _sim(mk, role=F, term=5, log=(1, [(1, 3), (6, 4)], 10))
    r = s.append_entries(5, 1, 7, 4, [4, 5, 5], 0)            # idx 8 ok, idx 9: 4 != 5 -> RocksLog.conflict :210-212
    assert (r.status, r.success, r.truncated, r.appended, r.log_from) == (abi.OK, True, True, True, 9)
    st = s.state()
    assert (st.last, st.runs) == (10, [(1, 3), (6, 4), (9, 5)])
    r = s.append_entries(5, 1, 5, 3, [6], 0)                  # conflict at 6: whole tail goes, shorter log
    assert (r.truncated, r.log_from) == (True, 6)
    st = s.state()
    assert (st.last, st.runs) == (6, [(1, 3), (6, 6)])


def ae_stale_duplicate_never_shortens(mk):                    # Q10
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5))
    r = s.append_entries(5, 1, 5, 5, [5, 5], 0)               # RocksLog.append :183 skips nothing new
    assert r.success and not r.appended and not r.truncated
    assert s.state().last == 10
    r = s.append_entries(5, 1, 9, 5, [5, 5, 5], 0)            # overlap 10, new 11..12
    assert r.appended and r.log_from == 11 and s.state().last == 12


def ae_purge_entries(mk):
    s = _sim(mk, role=F, term=5, epoch=(5, 3), log=(6, [(6, 3)], 8))
    r = s.append_entries(5, 1, 3, 2, [2, 3, 3, 3, 4], 0)      # idx 4,5 purged (:209-221); idx 8: 3 != 4 conflict
    assert (r.status, r.success, r.truncated, r.log_from) == (abi.OK, True, True, 8)
    assert s.state().runs == [(6, 3), (8, 4)]
    r = s.append_entries(5, 1, 3, 2, [2, 3], 0)               # everything purged -> entries = null
    assert r.success and not r.appended and not r.truncated


def ae_commit_rollback(mk):                                   # Q8
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5), commit=9)
    r = s.append_entries(5, 1, 10, 5, [], 7)                  # RocksLog.markCommitted :101-103
    assert r.status == abi.A_COMMIT_ROLLBACK and not r.replied and not r.commit_adv
    st = s.state()
    assert (st.commit, st.leader) == (9, 1)
    r = s.append_entries(5, 1, 10, 5, [5], 7)                 # entries are appended BEFORE the throw
    assert r.status == abi.A_COMMIT_ROLLBACK and r.appended and r.log_from == 11 and not r.replied
    assert s.state().last == 11


def ae_commit_guards(mk):
    s = _sim(mk, role=F, term=5, epoch=(5, 3))               # empty log: last()==null :77-78
    r = s.append_entries(5, 1, 5, 3, [], 9)
    assert r.success and not r.commit_adv and s.state().commit == 0
    s = _sim(mk, role=F, term=5, epoch=(5, 3), log=(6, [(6, 3)], 8))
    r = s.append_entries(5, 1, 8, 3, [], 5)                   # leaderCommit > epoch.index is false :76
    assert r.success and not r.commit_adv


# --------------------------------------------------------------------------------------------------
# a5: voter side  Follower.java:91-127,193-207  Candidate.java:44-72  Leader.java:89-111

def rv_follower(mk):                                          # KAT-6
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5))
    r = s.request_vote(4, 1, 99, 9)
    assert (r.replied, r.success, r.resp_term) == (True, False, 5)                  # :112-113
    assert not s.request_vote(5, 1, 99, 9).success                                  # :115 votedFor()==null
    s.load(role=F, term=5, voted_for=1, log=simple_log(10, 5))
    r = s.request_vote(5, 1, 0, 0)
    assert (r.success, r.resp_term, r.role_changed) == (True, 5, False)             # :115 same candidate, no log check
    assert not s.request_vote(5, 2, 99, 9).success
    r = s.request_vote(6, 2, 10, 5)                                                 # :122-126 fresh log
    assert (r.status, r.success, r.resp_term, r.persist, r.reset_timer) == (abi.OK, True, 6, True, True)
    assert (r.p_term, r.p_vote) == (6, 2)
    st = s.state()
    assert (st.term, st.voted_for, st.role_epoch) == (6, 2, 2)
    r = s.request_vote(7, 1, 9, 5)                                                  # stale log: term still bumps, vote null
    assert (r.success, r.resp_term, r.p_term, r.p_vote) == (False, 7, 7, NO)
    st = s.state()
    assert (st.term, st.voted_for) == (7, NO)
    assert s.request_vote(8, 1, 3, 6).success                                       # higher last term wins :196


def rv_follower_without_last(mk):                             # KAT-3
    def fresh():
        return _sim(mk, role=F, term=5, epoch=(5, 3))
    assert fresh().request_vote(6, 1, 5, 3).success                                 # :205
    assert fresh().request_vote(6, 1, 9, 3).success
    r = fresh().request_vote(6, 1, 4, 3)
    assert (r.status, r.success, r.resp_term) == (abi.OK, False, 6)
    r = fresh().request_vote(6, 1, 6, 2)                                            # :199
    assert r.status == abi.A_IMPOSSIBLE_LOG and not r.replied and r.reset_timer and not r.role_changed
    s = fresh()
    assert s.request_vote(6, 1, 5, 4).status == abi.A_IMPOSSIBLE_LOG               # :200
    assert s.state().term == 5


def pv_follower(mk):                                          # KAT-10
    s = _sim(mk, role=F, term=5, log=simple_log(10, 5))
    r = s.pre_vote(6, 1, 10, 5)                                                     # :94 !timeoutDetected
    assert (r.replied, r.success, r.resp_term, r.reset_timer) == (True, False, 5, False)
    s.load(role=F, term=5, log=simple_log(10, 5), timeout_detected=1, role_epoch=3)
    r = s.pre_vote(6, 1, 10, 5)                                                     # :100-101 replies currentTerm
    assert (r.status, r.success, r.resp_term, r.reset_timer, r.role_changed) == (abi.OK, True, 5, True, False)
    st = s.state()
    assert (st.term, st.role_epoch, st.voted_for, st.timeout_detected) == (5, 3, NO, 1)
    assert not s.pre_vote(5, 1, 10, 5).success                                      # term <= currentTerm
    assert not s.pre_vote(6, 1, 9, 5).success                                       # stale
    assert s.pre_vote(6, 1, 9, 5).resp_term == 5


def vote_at_candidate(mk):                                    # KAT-7 / Q5
    def cand(**kw):
        return _sim(mk, role=C, term=5, voted_for=0, log=simple_log(10, 5), role_epoch=2, **kw)
    s = cand()
    assert s.request_vote(6, 0, 10, 5).status == abi.A_CAND_SELF_RV                 # Candidate.java:53-55
    r = s.request_vote(4, 1, 10, 5)
    assert (r.success, r.resp_term) == (False, 5)
    r = s.request_vote(5, 1, 10, 5)                                                 # :62-63
    assert (r.success, r.resp_term, r.role) == (False, 5, C)
    r = s.request_vote(6, 1, 0, 0)                                                  # :70-71 stale log still granted
    assert (r.status, r.success, r.resp_term, r.role, r.role_epoch) == (abi.OK, True, 6, F, 3)
    st = s.state()
    assert (st.role, st.term, st.voted_for) == (F, 6, 1)
    s = cand()
    r = s.pre_vote(6, 2, 0, 0)                                                      # :44-46 PreVote == RequestVote here
    assert (r.success, r.resp_term, r.role) == (True, 6, F)
    assert (s.state().term, s.state().voted_for) == (6, 2)
    s = _sim(mk, role=C, term=5, voted_for=1)
    assert s.request_vote(5, 1, 0, 0).status == abi.A_CAND_NOT_SELF_VOTE            # :64-66


def vote_at_leader(mk):
    def lead(**kw):
        return _sim(mk, role=L, term=5, voted_for=0, log=simple_log(10, 5), role_epoch=2, **kw)
    s = lead()
    r = s.pre_vote(9, 1, 99, 9)                                                     # Leader.java:89-91
    assert (r.replied, r.success, r.resp_term, r.role) == (True, False, 5, L)
    assert s.request_vote(4, 1, 99, 9).resp_term == 5
    r = s.request_vote(5, 1, 99, 9)                                                 # :99-100
    assert (r.status, r.success, r.resp_term) == (abi.OK, False, 5)
    r = s.request_vote(6, 1, 10, 5)                                                 # :108 then Follower.java:122-126
    assert (r.status, r.success, r.resp_term, r.role, r.role_epoch) == (abi.OK, True, 6, F, 4)
    assert (s.state().term, s.state().voted_for) == (6, 1)
    s = lead()
    r = s.request_vote(6, 1, 9, 5)
    assert (r.success, r.resp_term, r.p_vote) == (False, 6, NO)
    s = _sim(mk, role=L, term=5, voted_for=1)
    assert s.request_vote(5, 2, 0, 0).status == abi.A_LEADER_NOT_SELF_VOTE          # :101-105
    s = _sim(mk, role=L, term=5, voted_for=0, epoch=(5, 3), role_epoch=2)
    r = s.request_vote(9, 1, 6, 2)                                                  # first conversion sticks, then Follower.java:199 throws
    assert r.status == abi.A_IMPOSSIBLE_LOG and not r.replied and r.persist and (r.p_term, r.p_vote) == (5, 1)
    st = s.state()
    assert (st.role, st.term, st.voted_for, st.role_epoch) == (F, 5, 1, 3)


# --------------------------------------------------------------------------------------------------
# a6/a7: tallies and timeouts  Candidate.java:82-143  Follower.java:156-168,223-279

def election_tally(mk):                                       # Q7, Q13
    s = _sim(mk, cluster=5, role=C, term=6, voted_for=0, role_epoch=7)
    r = s.rv_reply(1, 6, True, 7)
    assert (r.status, r.role, s.state().votes) == (abi.OK, C, 2)
    assert s.rv_reply(3, 6, False, 7).status == abi.OK and s.state().votes == 2
    assert s.rv_reply(1, 6, True, 6).status == abi.DROPPED_STALE_ROLE               # fenced head
    r = s.rv_reply(2, 6, True, 7)                                                   # votes 3 >= majority 3
    assert (r.status, r.role, r.role_epoch, r.persist, r.emit) == (abi.OK, L, 8, True, abi.EMIT_NONE)
    assert (r.p_term, r.p_vote, r.p_role) == (6, 0, L)
    st = s.state()
    assert (st.role, st.term, st.elected_epoch, st.elected_term, st.repl_prepared) == (L, 6, 7, 6, 0)
    r = s.rv_reply(3, 6, True, 7)                                                   # Q13 late grant: Leader stays (Membership.java:96)
    assert (r.status, r.role, r.role_changed) == (abi.OK, L, False)
    r = s.rv_reply(4, 9, False, 7)                                                  # Q13 late higher term: step down, ballot = responder
    assert (r.status, r.role, r.p_term, r.p_vote) == (abi.OK, F, 9, 4)
    st = s.state()
    assert (st.role, st.term, st.voted_for, st.elected_epoch) == (F, 9, 4, 0)
    assert s.rv_reply(3, 6, True, 7).status == abi.DROPPED_STALE_ROLE               # head aborted now


def election_higher_term_reply(mk):
    s = _sim(mk, cluster=5, role=C, term=6, voted_for=0, role_epoch=7)
    r = s.rv_reply(2, 8, False, 7)                                                  # Candidate.java:124-126
    assert (r.role, r.p_term, r.p_vote, r.role_epoch) == (F, 8, 2, 8)
    assert s.state().elected_epoch == 0
    assert s.rv_reply(1, 6, True, 7).status == abi.DROPPED_STALE_ROLE


def prevote_round(mk):
    s = _sim(mk, cluster=5, pre_vote=True, role=F, term=5, voted_for=3, leader=3, log=simple_log(10, 5))
    r = s.on_timeout()                                                              # Follower.java:158-164
    assert (r.status, r.role, r.role_epoch, r.emit, r.persist) == (abi.OK, F, 2, abi.EMIT_PREVOTE, True)
    st = s.state()
    assert (st.term, st.voted_for, st.leader, st.timeout_detected, st.votes) == (5, 3, NO, 1, 1)
    assert s.pv_reply(1, 6, False, 2).status == abi.OK                              # result.term == nextTerm: not '>'
    assert s.state().role_epoch == 2
    assert s.pv_reply(1, 5, True, 2).role == F
    r = s.pv_reply(2, 5, True, 2)                                                   # :265-266 -> Candidate(nextTerm)
    assert (r.status, r.role, r.role_epoch, r.emit, r.p_term, r.p_vote) == (abi.OK, C, 3, abi.EMIT_REQVOTE, 6, 0)
    st = s.state()
    assert (st.role, st.term, st.voted_for, st.votes, st.timeout_detected) == (C, 6, 0, 1, 0)
    assert s.pv_reply(4, 5, True, 2).status == abi.DROPPED_STALE_ROLE
    s = _sim(mk, cluster=5, pre_vote=True, role=F, term=5, timeout_detected=1, role_epoch=2)
    r = s.pv_reply(4, 7, False, 2)                                                  # :261-263 result.term > nextTerm
    assert (r.role, r.p_term, r.p_vote, r.role_epoch) == (F, 7, 4, 3)
    assert s.state().timeout_detected == 0


def timeouts(mk):
    s = _sim(mk, pre_vote=False, role=F, term=5, voted_for=1)
    r = s.on_timeout()                                                              # Follower.java:166
    assert (r.role, r.emit, r.p_term, r.p_vote, r.role_epoch) == (C, abi.EMIT_REQVOTE, 6, 0, 2)
    r = s.on_timeout()                                                              # Candidate.java:83
    assert (r.role, r.emit, r.p_term, r.role_epoch) == (C, abi.EMIT_REQVOTE, 7, 3)
    assert s.state().votes == 1
    s = _sim(mk, role=L, term=5, voted_for=0, epoch=(2, 1), log=(3, [(3, 4)], 10), role_epoch=4)
    r = s.on_timeout()                                                              # Leader.java:120-126 -> prepareReplication :30-50
    assert (r.status, r.role, r.emit, r.role_changed, r.reset_timer) == (abi.OK, L, abi.EMIT_HEARTBEAT, False, True)
    st = s.state()
    assert st.repl_prepared == 1 and st.peers == [(2, 11, 0, 0, 0)] * 2
    s.load(role=L, term=5, voted_for=0, epoch=(2, 1), role_epoch=4, repl_prepared=1,
           peers=[(2, 7, 5, 1, 0), (2, 3, 0, 0, 1)])
    s.on_timeout()                                                                  # already prepared: untouched
    assert s.state().peers == [(2, 7, 5, 1, 0), (2, 3, 0, 0, 1)]
    s = _sim(mk, role=L, term=5, voted_for=0, epoch=(7, 2))                        # empty log -> epoch.index+1
    s.on_timeout()
    assert s.state().peers == [(7, 8, 0, 0, 0)] * 2


# --------------------------------------------------------------------------------------------------
# a3/a4: Leader ack path  Leader.java:174-188,218-280  Leadership.java:53-63,75-130

def _leader(mk, cluster=3, log=(1, [(1, 4), (51, 5)], 100), peers=None, commit=0, epoch=(0, 0)):
    peers = peers or [(0, 101, 0, 0, 0)] * (cluster - 1)
    return _sim(mk, cluster=cluster, role=L, term=5, voted_for=0, role_epoch=3, repl_prepared=1,
                log=log, peers=peers, commit=commit, epoch=epoch)


def ack_commit_current_term(mk):
    s = _leader(mk)
    r = s.ae_ack(1, 5, True, 0, 100, 3)                       # F=2: sorted[1] = 100, term(100)==5 -> commit major
    assert (r.status, r.commit_adv, r.commit) == (abi.OK, True, 100)
    st = s.state()
    assert st.peers[0] == (0, 101, 100, 0, 0) and st.commit == 100
    s = _leader(mk, cluster=5)                                # F=4: major = sorted[4/2]
    assert not s.ae_ack(1, 5, True, 0, 90, 3).commit_adv      # sorted [0,0,0,90] -> major 0
    r = s.ae_ack(2, 5, True, 0, 80, 3)                        # sorted [0,0,80,90] -> major 80
    assert (r.commit_adv, r.commit) == (True, 80)             # Leadership.java:121-126 table, N=5
    r = s.ae_ack(3, 5, True, 0, 85, 3)                        # sorted [0,80,85,90] -> 85
    assert (r.commit_adv, r.commit) == (True, 85)


def ack_commit_old_term_uses_full(mk):                        # KAT-8 / Q6
    s = _leader(mk, log=simple_log(100, 4))
    r = s.ae_ack(1, 5, True, 0, 100, 3)                       # major=100, term 4 != 5 -> commit = full = 0 -> nothing
    assert (r.status, r.commit_adv) == (abi.OK, False)
    r = s.ae_ack(2, 5, True, 0, 60, 3)                        # full = 60: old-term entry commits once on ALL followers
    assert (r.commit_adv, r.commit) == (True, 60)


def ack_reject_backoff(mk):                                   # KAT-4
    s = _leader(mk, peers=[(0, 100, 0, 0, 0), (0, 100, 0, 0, 0)])
    for want_next, want_rej in ((99, 1), (97, 2), (95, 3)):   # step = round(ln(e+r)) = 1, 2, 2
        r = s.ae_ack(1, 5, False, 0, 99, 3)
        assert r.status == abi.OK
        p = s.state().peers[0]
        assert (p[1], p[3]) == (want_next, want_rej)
    s.ae_ack(1, 5, True, 0, 94, 3)                            # success: rejection counter resets (Leadership.java:60-62)
    assert s.state().peers[0] == (0, 95, 94, 0, 0)
    s.ae_ack(1, 5, False, 0, 94, 3)                           # matchIndex != 0: only the counter moves (:103)
    assert s.state().peers[0] == (0, 95, 94, 1, 0)


def ack_backoff_floor_and_snapshot(mk):
    s = _leader(mk, epoch=(90, 4), log=(91, [(91, 5)], 100), peers=[(90, 92, 0, 0, 0), (90, 101, 0, 0, 0)])
    s.ae_ack(1, 5, False, 90, 91, 3)                          # next = max(92-1, 91) = 91
    assert s.state().peers[0] == (90, 91, 0, 1, 0)
    s.ae_ack(1, 5, False, 90, 90, 3)                          # next = max(91-2, 91) = 91 ; min(90, 91) = 90 <= epoch -> pending
    assert s.state().peers[0] == (90, 90, 0, 2, 1)
    s.ae_ack(1, 5, True, 90, 95, 3)                           # pendingInstallation != snapshot: ignored (:90), counter resets
    assert s.state().peers[0] == (90, 90, 0, 0, 1)
    r = s.is_ack(1, 5, False, 90, 3)                          # snapshot refused: stays pending
    assert r.status == abi.OK and s.state().peers[0] == (90, 90, 0, 1, 1)
    s.is_ack(1, 5, True, 90, 3)                               # :92-96 nextIndex = max(nextIndex, epoch+1)
    assert s.state().peers[0] == (90, 91, 0, 0, 0)
    r = s.is_ack(1, 5, True, 90, 3)                           # not pending any more: ignored
    assert s.state().peers[0] == (90, 91, 0, 0, 0) and not r.commit_adv


def ack_epoch_rules(mk):                                      # Q11
    s = _leader(mk, epoch=(0, 0), log=(1, [(1, 5)], 200), peers=[(50, 101, 10, 0, 0), (0, 101, 0, 0, 0)])
    s.ae_ack(1, 5, True, 40, 150, 3)                          # epoch < lastEpoch: ignored (:83)
    assert s.state().peers[0] == (50, 101, 10, 0, 0)
    s.ae_ack(1, 5, True, 120, 150, 3)                         # epoch > lastEpoch (:84-88) then success
    assert s.state().peers[0] == (120, 151, 150, 0, 0)
    r = s.ae_ack(1, 5, True, 120, 140, 3)                     # index < matchIndex: AbstractMethodError (:76-81)
    assert r.status == abi.A_MATCH_ROLLBACK
    assert s.state().peers[0] == (120, 151, 150, 0, 0)
    r = s.ae_ack(1, 5, False, 120, 140, 3)                    # counter moved BEFORE the throw (Leader.java:229 precedes :230)
    assert r.status == abi.A_MATCH_ROLLBACK and s.state().peers[0][3] == 1


def ack_higher_term_steps_down(mk):                           # Q7
    s = _leader(mk)
    r = s.ae_ack(2, 8, False, 0, 100, 3)                      # Leader.java:224-226
    assert (r.status, r.role, r.p_term, r.p_vote, r.role_epoch) == (abi.OK, F, 8, 2, 4)
    assert s.ae_ack(1, 5, True, 0, 100, 3).status == abi.DROPPED_STALE_ROLE
    s = _leader(mk)
    assert s.is_ack(2, 8, True, 0, 3).role == F               # :178-180


def ack_major_null_and_rollback(mk):
    s = _leader(mk)
    r = s.ae_ack(1, 5, True, 0, 200, 3)                       # log.get(200)==null -> NPE, caught (:256-257,277-279)
    assert r.status == abi.NPE_MAJOR_NULL and not r.commit_adv
    assert s.state().peers[0][2] == 200
    s = _leader(mk, log=simple_log(100, 4), peers=[(0, 71, 70, 0, 0), (0, 101, 0, 0, 0)], commit=80)
    r = s.ae_ack(2, 5, True, 0, 60, 3)                        # full=60 < lastCommitted 80 (Q6 + Q8)
    assert r.status == abi.A_COMMIT_ROLLBACK and s.state().commit == 80
    s = _leader(mk, peers=[(0, 101, 100, 0, 0), (0, 101, 100, 0, 0)], commit=100)
    r = s.ae_ack(1, 5, True, 0, 100, 3)                       # commitIndex == lastCommitted: nothing (:262)
    assert r.status == abi.OK and not r.commit_adv


def ack_unprepared_or_wrong_role(mk):
    s = _sim(mk, role=L, term=5, voted_for=0, role_epoch=3)   # nothing was ever sent under this epoch
    assert s.ae_ack(1, 5, True, 0, 1, 3).status == abi.BAD_EVENT
    s = _sim(mk, role=F, term=5, role_epoch=3)
    assert s.ae_ack(1, 5, True, 0, 1, 3).status == abi.BAD_EVENT
    assert s.ae_ack(0, 5, True, 0, 1, 3).status == abi.BAD_EVENT                    # responder == self


# --------------------------------------------------------------------------------------------------
# host-driven log changes the device must mirror

def client_append(mk):
    s = _sim(mk, role=L, term=5, voted_for=0, log=simple_log(10, 4))
    r = s.client_append(2)                                    # RocksLog.newEntry :82-89; prepareReplication after the FIRST entry
    assert (r.status, r.appended, r.log_from, r.emit) == (abi.OK, True, 11, abi.EMIT_HEARTBEAT)
    st = s.state()
    assert (st.last, st.runs, st.repl_prepared) == (12, [(1, 4), (11, 5)], 1)
    assert st.peers == [(0, 12, 0, 0, 0)] * 2
    s = _sim(mk, role=L, term=5, voted_for=0)
    assert s.client_append(1).log_from == 1 and s.state().runs == [(1, 5)]
    assert _sim(mk, role=F, term=5).client_append(1).status == abi.NOT_LEADER      # RaftStub.java:83-89
    assert _sim(mk, role=L, term=5, voted_for=0, epoch=(7, 2)).client_append(1).status == abi.UNSUPPORTED_LOG_STATE


def log_flush(mk):
    s = _sim(mk, role=F, term=5, log=(1, [(1, 3), (6, 4)], 10), commit=8)
    assert s.log_flush(5, 3).status == abi.OK                 # RocksLog.flush :228-242: key == index survives
    st = s.state()
    assert (st.epoch, st.first, st.last, st.runs) == ((5, 3), 5, 10, [(5, 3), (6, 4)])
    assert s.log_flush(4, 3).status == abi.FLUSH_OUT_OF_BOUNDS
    assert s.append_entries(5, 1, 5, 9, [], 0).status == abi.A_EPOCH_TERM_MISMATCH
    s.log_flush(7, 4)
    st = s.state()
    assert (st.epoch, st.first, st.runs) == ((7, 4), 7, [(7, 4)])
    s.log_flush(20, 6)                                        # beyond last: log emptied
    st = s.state()
    assert (st.epoch, st.runs, st.last) == ((20, 6), [], 0)
    r = s.append_entries(6, 1, 20, 6, [6], 0)                 # first key after an install: epoch.index+1
    assert r.success and r.log_from == 21 and s.state().first == 21


# --------------------------------------------------------------------------------------------------
# N1: the leader's send side  Leader.replicateLog member/Leader.java:142-245, RaftLog.batch storage/RocksLog.java:131-166

def _send(s, heartbeat, in_flight=None):
    head, send = s.t.replicate(heartbeat=heartbeat, in_flight=in_flight)
    rows = [(int(x["prev_index"]), int(x["prev_term"]), int(x["last_index"]), int(x["count"]), int(x["kind"])) for x in send[0]]
    h = head[0]
    return (int(h["term"]), int(h["leader_commit"]), int(h["epoch_index"]), int(h["epoch_term"]), int(h["role_epoch"]),
            int(h["is_leader"])), rows


def replicate_ranges(mk):
    s = _sim(mk, role=L, term=5, voted_for=0, role_epoch=4, epoch=(2, 1), log=(3, [(3, 4), (8, 5)], 10), commit=6,
             repl_prepared=1, peers=[(2, 11, 0, 0, 0), (2, 5, 4, 0, 0)])
    head, rows = _send(s, heartbeat=1)
    assert head == (5, 6, 2, 1, 4, 1)
    # follower 0: nextIndex 11 -> prev = entry 10 (term 5), nothing to ship, lastIndex = prevIndex (:204-206)
    assert rows[0] == (10, 5, 10, 0, abi.SEND_APPEND)
    # follower 1: nextIndex 5 -> batch(4, 26): entry 4 is prevLog (term 4), entries 5..10
    assert rows[1] == (4, 4, 10, 6, abi.SEND_APPEND)
    s.load(role=L, term=5, voted_for=0, epoch=(2, 1), log=(3, [(3, 4)], 100), repl_prepared=1,
           peers=[(2, 3, 0, 0, 0), (2, 40, 0, 0, 0)])
    _, rows = _send(s, heartbeat=1)
    # nextIndex-1 == epoch.index: batch skips the epoch key, prevLog stays the epoch, fetchLimit 25 entries (:193-203)
    assert rows[0] == (2, 1, 27, 25, abi.SEND_APPEND)
    assert rows[1] == (39, 4, 64, 25, abi.SEND_APPEND)
    _, rows = _send(s, heartbeat=0)                            # acceptCommand path: REPLICATE_LIMIT 50
    assert rows[0] == (2, 1, 52, 50, abi.SEND_APPEND) and rows[1] == (39, 4, 89, 50, abi.SEND_APPEND)


def replicate_gates_and_snapshot(mk):
    s = _sim(mk, role=L, term=5, voted_for=0, epoch=(20, 3), log=(21, [(21, 5)], 30), repl_prepared=1,
             peers=[(20, 31, 30, 0, 0), (20, 20, 0, 3, 1)])
    _, rows = _send(s, heartbeat=1, in_flight=[[3, 0]])        # 3 > IN_FLIGHT_LIMIT/10 (:162-166)
    assert rows[0][4] == abi.SEND_GATED
    assert rows[1] == (20, 3, 20, 0, abi.SEND_SNAPSHOT)        # pendingInstallation -> installSnapshot(epoch) (:168-190)
    _, rows = _send(s, heartbeat=0, in_flight=[[3, 21]])       # limit 20 on the acceptCommand path
    assert rows[0] == (30, 5, 30, 0, abi.SEND_APPEND) and rows[1][4] == abi.SEND_GATED
    s.load(role=L, term=5, voted_for=0, epoch=(20, 3), repl_prepared=1, peers=[(20, 21, 0, 0, 0)] * 2)
    _, rows = _send(s, heartbeat=1)                            # empty log: heartbeat on the epoch, lastIndex = epoch.index (:209-211)
    assert rows[0] == (20, 3, 20, 0, abi.SEND_APPEND)


def replicate_prepares_and_skips_non_leaders(mk):
    s = _sim(mk, role=L, term=5, voted_for=0, epoch=(0, 0), log=simple_log(7, 5), role_epoch=9)
    head, rows = _send(s, heartbeat=1)                         # first send of a new leader: prepareReplication (:146, :30-50)
    assert head[4:] == (9, 1) and rows == [(7, 5, 7, 0, abi.SEND_APPEND)] * 2
    st = s.state()
    assert st.repl_prepared == 1 and st.peers == [(0, 8, 0, 0, 0)] * 2
    s = _sim(mk, role=F, term=5)
    head, rows = _send(s, heartbeat=1)
    assert head[5] == 0 and all(r[4] == abi.SEND_NONE for r in rows)
    assert s.state().repl_prepared == 0


# --------------------------------------------------------------------------------------------------
# N4: timers  RaftRoutine.resetTimer / electionTimeout / keepAlive  context/RaftRoutine.java:53-130

def timers_follow_reset_timer(mk):
    E, H = 900, 300                                            # raft1.xml:10-13 through RaftConfig.java:187-198
    s = _sim(mk, role=F, term=5, voted_for=1, leader=1, log=simple_log(10, 5))
    t = s.t
    t.timers_configure(E, H, 11)
    assert t.timers_read()[0] == 0                             # no ticket yet
    t.timers_arm(1000)
    d0 = int(t.timers_read()[0])
    assert 1000 + E <= d0 <= 1000 + 2 * E                      # election timeout in [E, 2E]
    assert t.timers_expired(d0 - 1)[1] == 0

    def step(now, **ev):
        b = abi.Batch(1, 1)
        b.put(0, 0, **ev)
        return t.submit_and_update_timers(b, [now])

    step(1500, kind=abi.EV_AE_REQ, slot=1, a=5, b=10, c=5, d=0)   # heartbeat: Follower.java:43,84 re-arm the timer
    d1 = int(t.timers_read()[0])
    assert 1500 + E <= d1 <= 1500 + 2 * E
    step(1600, kind=abi.EV_AE_REQ, slot=1, a=4, b=10, c=5, d=0)   # stale term: no resetTimer (Follower.java:39-41)
    assert int(t.timers_read()[0]) == d1
    gids, n = t.timers_expired(d1)                              # electionTimeout: CAS deadline -> TIMEOUT (:68)
    assert (gids.tolist(), n) == ([0], 1) and int(t.timers_read()[0]) == -1
    assert t.timers_expired(d1 + 10 ** 6)[1] == 0              # fires once
    step(d1 + 1, kind=abi.EV_AE_REQ, slot=1, a=5, b=10, c=5, d=0)  # same participant, ticket already fired: moment < 0 (:96-98)
    assert int(t.timers_read()[0]) == -1
    out = step(d1 + 2, kind=abi.EV_TIMEOUT)                     # onTimeout -> new participant -> fresh ticket
    assert out.reply["flags"][0] & abi.F_ROLE_CHANGED
    d2 = int(t.timers_read()[0])
    assert d1 + 2 + E <= d2 <= d1 + 2 + 2 * E
    s = _sim(mk, cluster=3, role=C, term=6, voted_for=0, role_epoch=3)
    t = s.t
    t.timers_configure(E, H, 11)
    b = abi.Batch(1, 1)
    b.put(0, 0, abi.EV_RV_REPLY, slot=1, flag=1, a=6, aux=3)    # majority of 3 -> Leader
    t.submit_and_update_timers(b, [7000])
    assert int(t.timers_read()[0]) == 7000                      # first keepAlive at once (:117-118 exist == null ? 0)
    assert t.timers_expired(7000)[0].tolist() == [0]
    b = abi.Batch(1, 1)
    b.put(0, 0, abi.EV_TIMEOUT)                                 # the heartbeat tick itself
    t.submit_and_update_timers(b, [7001])
    assert int(t.timers_read()[0]) == 7001 + H


def extreme_values(mk):
    """Java longs near their limits: nothing on the path may truncate to 32 bits or misorder signed values."""
    big = (1 << 62) + 12345
    top = (1 << 63) - 1000
    s = _sim(mk, role=F, term=big, voted_for=1, epoch=(top - 50, big - 3), log=(top - 49, [(top - 49, big - 2), (top - 20, big - 1)], top))
    r = s.append_entries(big, 1, top, big - 1, [big, big], top + 1)          # append at the very end of the index space
    assert (r.status, r.success, r.resp_term, r.log_from, r.commit) == (abi.OK, True, big, top + 1, top + 1)
    st = s.state()
    assert (st.last, st.last_term, st.commit) == (top + 2, big, top + 1)
    r = s.append_entries(big + (1 << 40), 2, top - 30, big - 2, [big - 2], 0)  # higher term far away, prevLog deep in the first run
    assert (r.success, r.resp_term, r.p_term) == (True, big + (1 << 40), big + (1 << 40))
    assert not s.append_entries(big, 1, top, big - 1, [], 0).success          # now stale: (cur, false)
    assert s.request_vote(big + (1 << 41), 2, top + 2, big).success
    s = _sim(mk, cluster=5, role=L, term=big, voted_for=0, role_epoch=0xFFFFFFF0, repl_prepared=1,
             log=(1, [(1, big)], top), peers=[(0, top - 5, top - 6, 0, 0)] * 4)
    s.ae_ack(1, big, True, 0, top - 1, 0xFFFFFFF0)
    r = s.ae_ack(2, big, True, 0, top - 2, 0xFFFFFFF0)
    assert (r.status, r.commit_adv, r.commit) == (abi.OK, True, top - 2)
    _, rows = _send(s, heartbeat=0)
    assert rows[0] == (top - 1, big, top, 1, abi.SEND_APPEND) and rows[2] == (top - 6, big, top, 6, abi.SEND_APPEND)
    r = s.ae_ack(3, big + 1, False, 0, 0, 0xFFFFFFF0)                          # role epoch wraps around 2^32 on the step-down
    assert (r.role, r.role_epoch) == (F, 0xFFFFFFF1)


def none_rows_do_nothing(mk):
    s = _sim(mk, role=C, term=5, voted_for=0, role_epoch=9)
    r = s.event(abi.EV_NONE)
    assert (r.status, r.replied, r.role, r.role_epoch, r.flags & 0xFF) == (abi.OK, False, C, 9, 0)


def install_snapshot_request(mk):
    """RaftParticipant.installSnapshot: member/Follower.java:129-152; Candidate / Leader inherit member/RaftMember.java:61-66"""
    s = _sim(mk, role=F, term=5, voted_for=1, leader=1, role_epoch=3, log=simple_log(10, 5))
    r = s.install_snapshot(4, 1, 50, 4)                        # :136-137 stale term -> failure(currentTerm); :134 muted the timer first
    assert (r.status, r.replied, r.success, r.resp_term, r.reset_timer, r.muted, r.role_changed) == (abi.OK, True, False, 5, True, True, False)
    r = s.install_snapshot(6, 1, 50, 4)                        # :138-139 a leader must appendEntries first
    assert (r.status, r.replied, r.reset_timer, r.muted) == (abi.A_INSTALL_BEFORE_AE, False, True, True)
    r = s.install_snapshot(5, 1, 50, 4, host_ok=True)          # :147-152 the verdict of ctx.installSnapshot, timer un-muted by the finally
    assert (r.status, r.replied, r.success, r.resp_term, r.muted, r.persist) == (abi.OK, True, True, 5, False, False)
    r = s.install_snapshot(5, 1, 50, 4, host_ok=False)
    assert (r.replied, r.success, r.resp_term) == (True, False, 5)
    assert s.state().epoch == (0, 0) and s.state().last == 10  # the epoch moves with the host's RG_EV_LOG_FLUSH, not here
    s.on_timeout()                                             # pre-vote pending: timeoutDetected
    assert s.state().timeout_detected == 1 and s.state().role_epoch == 4
    r = s.install_snapshot(5, 1, 50, 4)                        # :140-142 refresh to a fresh Follower of the same term, then as above
    assert (r.status, r.success, r.persist, r.role_changed, r.role_epoch, r.p_term, r.p_vote, r.muted) == (abi.OK, True, True, True, 5, 5, 1, False)
    assert s.state().timeout_detected == 0 and s.state().leader == abi.NO_NODE
    for role in (C, L):                                        # RaftMember.installSnapshot: term >= currentTerm asserts, lower fails
        s = _sim(mk, role=role, term=5, voted_for=0, role_epoch=2)
        r = s.install_snapshot(4, 1, 9, 3)
        assert (r.status, r.replied, r.success, r.resp_term, r.reset_timer) == (abi.OK, True, False, 5, False)
        for term in (5, 6):
            r = s.install_snapshot(term, 1, 9, 3)
            assert (r.status, r.replied, r.reset_timer) == (abi.A_INSTALL_BEFORE_AE, False, False)
        assert s.state().role == role and s.state().role_epoch == 2


def timeout_of_a_replaced_participant_is_dropped(mk):
    """context/RaftRoutine.java:70: the queued timeout task runs onTimeout only if ticket.participant() == context.participant()"""
    s = _sim(mk, role=F, term=5, voted_for=1, leader=1, role_epoch=7, log=simple_log(10, 5))
    r = s.append_entries(6, 2, 10, 5, [], 0)                   # a higher-term leader replaces the participant (epoch 8) ...
    assert r.role_epoch == 8
    r = s.on_timeout(ticket_epoch=7)                           # ... before the timeout of the old one is drained: ignored
    assert (r.status, r.role_changed, r.persist, r.emit, r.role_epoch) == (abi.DROPPED_STALE_ROLE, False, False, 0, 8)
    assert s.state().timeout_detected == 0 and s.state().term == 6
    r = s.on_timeout(ticket_epoch=8)                           # the live participant's own ticket
    assert (r.status, r.role_changed, r.emit, r.role_epoch) == (abi.OK, True, 1, 9)
    r = s.on_timeout()                                         # aux 0: whoever is current (host-owned timers)
    assert (r.status, r.role_changed, r.role_epoch) == (abi.OK, True, 10)
    s = _sim(mk, role=L, term=5, voted_for=0, role_epoch=4, log=simple_log(3, 5))
    assert s.on_timeout(ticket_epoch=3).status == abi.DROPPED_STALE_ROLE and s.state().repl_prepared == 0
    r = s.on_timeout(ticket_epoch=4)                           # keepAlive of the live Leader (:57)
    assert (r.status, r.emit, r.reset_timer) == (abi.OK, 3, True) and s.state().repl_prepared == 1


def handlers_that_leave_the_timer_muted(mk):
    """resetTimer(this, true) parks the deadline at Long.MAX_VALUE (context/RaftRoutine.java:101-107); three paths return or
    throw before the un-muting call, and the election timer then never fires until a later handler re-arms it"""
    E, H = 900, 300
    s = _sim(mk, role=F, term=5, voted_for=1, leader=1, log=simple_log(10, 5))
    t = s.t
    t.timers_configure(E, H, 3)
    t.timers_arm(1000)

    def step(now, **ev):
        b = abi.Batch(1, 1)
        b.put(0, 0, **ev)
        return t.submit_and_update_timers(b, [now])

    out = step(1100, kind=abi.EV_AE_REQ, slot=2, a=5, b=10, c=5, d=0)        # member/Follower.java:43 mutes, :48-50 throws
    f = int(out.reply["flags"][0])
    assert (f >> abi.F_STATUS_SHIFT) & 0xFF == abi.A_TWO_LEADERS and f & abi.F_TIMER_MUTED and f & abi.F_RESET_TIMER
    assert int(t.timers_read()[0]) == 2 ** 63 - 1 and t.timers_expired(10 ** 12)[1] == 0
    out = step(5000, kind=abi.EV_AE_REQ, slot=1, a=5, b=10, c=5, d=0)        # the next good heartbeat re-arms it (:84)
    assert not int(out.reply["flags"][0]) & abi.F_TIMER_MUTED
    assert 5000 + E <= int(t.timers_read()[0]) <= 5000 + 2 * E
    out = step(5100, kind=abi.EV_AE_REQ, slot=1, a=5, b=0, c=3, d=0)         # logContains throws INSIDE the try: the finally un-mutes
    f = int(out.reply["flags"][0])
    assert (f >> abi.F_STATUS_SHIFT) & 0xFF == abi.A_PREV_ZERO_MISMATCH and not f & abi.F_TIMER_MUTED
    assert 5100 + E <= int(t.timers_read()[0]) <= 5100 + 2 * E
    out = step(5200, kind=abi.EV_IS_REQ, slot=1, flag=1, a=4, b=50, c=4)     # member/Follower.java:134 mutes, :136-137 returns
    assert int(out.reply["flags"][0]) & abi.F_TIMER_MUTED and int(t.timers_read()[0]) == 2 ** 63 - 1
    s = _sim(mk, role=F, term=5, voted_for=1, epoch=(20, 4))                 # empty log: logUpToDate can throw (:199-204)
    t = s.t
    t.timers_configure(E, H, 3)
    t.timers_arm(1000)
    out = step(1200, kind=abi.EV_RV_REQ, slot=2, a=6, b=20, c=3)             # :118 mutes, logUpToDate throws before :125
    f = int(out.reply["flags"][0])
    assert (f >> abi.F_STATUS_SHIFT) & 0xFF == abi.A_IMPOSSIBLE_LOG and f & abi.F_TIMER_MUTED
    assert int(t.timers_read()[0]) == 2 ** 63 - 1
    out = step(1300, kind=abi.EV_PV_REQ, slot=2, a=6, b=20, c=3)             # preVote needs timeoutDetected: not even muted (:94-96)
    assert not int(out.reply["flags"][0]) & (abi.F_TIMER_MUTED | abi.F_RESET_TIMER)


def oversized_append_entries_is_rejected(mk):
    """a row may carry at most RG_MAX_AE_ENTRIES = 4 x REPLICATE_LIMIT entries (member/Leadership.java:10, member/Leader.java:194)"""
    s = _sim(mk, role=F, term=5, voted_for=1, leader=1, log=simple_log(10, 5))
    r = s.append_entries(5, 1, 10, 5, [5] * abi.MAX_AE_ENTRIES, 0)
    assert (r.status, r.success) == (abi.OK, True) and s.state().last == 10 + abi.MAX_AE_ENTRIES
    r = s.append_entries(5, 1, 210, 5, [5] * (abi.MAX_AE_ENTRIES + 1), 0)
    assert (r.status, r.replied) == (abi.BAD_EVENT, False) and s.state().last == 210


# --------------------------------------------------------------------------------------------------
# N4b: follower health + Leader.isReady  member/Leadership.java:43-73, member/Leader.java:52-64

def health_ready_gate(mk):
    CP, CD = 2, 500                                            # availableCriticalPoint, recoveryCoolDownMills
    s = _sim(mk, cluster=5, role=C, term=5, voted_for=0, role_epoch=2, log=simple_log(100, 5))
    t = s.t
    s.now = 1000
    assert s.rv_reply(1, 5, True, 2).status == abi.OK
    assert s.rv_reply(2, 5, True, 2).role == L                 # majority of 5
    assert t.ready(1000, CP, CD)[0] == 0                       # followerStatus == null (Leader.java:53)
    s.on_timeout()                                             # keepAlive -> prepareReplication
    assert t.ready(1001, CP, CD)[0] == 0                       # requestSuccess == 0 everywhere (Leadership.java:49)
    s.now = 2000
    assert s.ae_ack(1, 5, True, 0, 100, 3).status == abi.OK
    ok, fl, rc = t.health_read()
    assert ok[0].tolist() == [2000, 0, 0, 0] and not fl.any() and not rc.any()
    assert t.ready(2001, CP, CD)[0] == 0                       # ready = 1+1 = 2, half = 4/2 = 2: 2 > 2 is false
    s.now = 2100
    s.ae_ack(3, 5, False, 0, 100, 3)                           # a rejection is still a statSuccess (Leader.java:229)
    assert t.ready(2101, CP, CD)[0] == 1                       # 3 > 2
    s.now = 1500
    s.ae_ack(3, 5, False, 0, 99, 3)                            # increaseMono: the clock never moves back (Leadership.java:39-41)
    assert t.health_read()[0][0].tolist() == [2000, 0, 2100, 0]
    # statFailure: unreachable x3 > criticalPoint 2 -> unhealthy
    for k in range(3):
        t.health_failure([0], [3], [1], 3000 + k)
        assert t.ready(10_000, CP, CD)[0] == (1 if k < 2 else 0), k          # Integer.compareUnsigned(recent, 2) > 0 at 3
    ok, fl, rc = t.health_read()
    assert fl[0].tolist() == [0, 0, 3002, 0] and rc[0].tolist() == [0, 0, 3, 0]
    assert t.ready(10_000, 0, CD)[0] == 1                      # criticalPoint 0 switches the check off
    s.now = 10_000
    s.ae_ack(3, 5, True, 0, 100, 3)                            # one success clears recentFailure (Leadership.java:55-57)
    assert t.health_read()[2][0].tolist() == [0, 0, 0, 0]
    assert t.ready(10_001, CP, CD)[0] == 1
    # cool-down: a failure less than CD ms ago keeps the peer out
    t.health_failure([0], [1], [0], 10_100)                    # error == null, result == null: neither counter moves
    assert t.health_read()[2][0].tolist() == [0, 0, 0, 0]
    assert t.ready(10_599, CP, CD)[0] == 0                     # now - requestFailure = 499 < 500
    assert t.ready(10_600, CP, CD)[0] == 1
    assert t.ready(10_599, CP, 0)[0] == 1                      # coolDown 0 switches the check off
    rej0 = s.state().peers[0][3]
    t.health_failure([0], [1], [2], 10_050)                    # reject: recentRejection++ ; requestFailure keeps its max
    assert s.state().peers[0][3] == rej0 + 1 and t.health_read()[1][0, 0] == 10_100
    # fenced / stale callbacks do not count
    s.now = 20_000
    assert s.ae_ack(2, 5, True, 0, 100, 2).status == abi.DROPPED_STALE_ROLE
    assert t.health_read()[0][0].tolist() == [2000, 0, 10_000, 0]
    # a higher term in the response: no statistics, the Leader steps down, isReady is false from then on
    r = s.ae_ack(2, 6, False, 0, 100, 3)
    assert r.role_changed and r.role == F
    assert t.health_read()[0][0, 1] == 0 and t.ready(20_001, CP, CD)[0] == 0
    t.health_failure([0], [1], [1], 20_002)                    # no State object any more: ignored
    assert t.health_read()[2][0].tolist() == [0, 0, 0, 0]


def health_small_clusters_and_pending(mk):
    s = _leader(mk, cluster=3)                                 # half = 2/2 = 1: one live follower is enough
    s.now = 500
    s.ae_ack(2, 5, True, 0, 100, 3)
    assert s.t.ready(501, 3, 100)[0] == 1
    s = _leader(mk, cluster=2)                                 # half = 1/2 = 0, but ++ready only runs for a ready State
    assert s.t.ready(501, 3, 100)[0] == 0
    s.now = 500
    s.ae_ack(1, 5, True, 0, 100, 3)
    assert s.t.ready(501, 3, 100)[0] == 1
    s = _leader(mk, cluster=3, peers=[(0, 101, 0, 0, 1), (0, 101, 0, 0, 0)])
    s.now = 700
    assert s.is_ack(1, 5, False, 0, 3).status == abi.OK        # IS-Echo: statSuccess too (Leader.java:182), still pending
    assert s.t.health_read()[0][0].tolist() == [700, 0]
    assert s.t.ready(701, 3, 100)[0] == 0                      # pendingInstallation (Leadership.java:49)
    s.is_ack(1, 5, True, 0, 3)
    assert s.t.ready(701, 3, 100)[0] == 1
    # match index rollback throws AFTER statSuccess (Leader.java:229-230)
    s = _leader(mk, cluster=3, peers=[(0, 101, 90, 0, 0), (0, 101, 0, 0, 0)])
    s.now = 900
    assert s.ae_ack(1, 5, True, 0, 80, 3).status == abi.A_MATCH_ROLLBACK
    assert s.t.health_read()[0][0].tolist() == [900, 0]
    # a reloaded table starts from zeroed statistics
    s.load(role=L, term=5, voted_for=0, role_epoch=3, repl_prepared=1, log=simple_log(10, 5), peers=[(0, 11, 0, 0, 0)] * 2)
    assert not s.t.health_read()[0].any()


SCENARIOS = [v for k, v in sorted(globals().items())
             if callable(v) and getattr(v, "__module__", None) == __name__ and not k.startswith("_")]
