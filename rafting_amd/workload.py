"""Synthetic RPC replay streams for BASELINE.json's configs.

A closed, deterministic workload model: for every raft group it tracks just enough of the protocol
state (term, log tail, role epoch, follower matchIndex ...) to emit the NEXT event a real cluster
could deliver to that group — AppendEntries requests to followers, AppendEntries acks and client
appends to leaders, and the PreVote / RequestVote fan-in of an election — one event per group per
round.  It never consults the decision engine or the oracle; tests replay its streams through both
and require (a) bit-identical results and (b) that virtually every row is a well-formed, non-stale,
assertion-free decision, which is what proves the model tracks the protocol correctly.

Randomness is counter based: every draw is splitmix64(seed, group id, round, draw#), so a group's
stream does not depend on which GPU / shard generates it (1/2/4/8-GPU runs are bit-comparable).
"""
from dataclasses import dataclass, replace

import numpy as np

from . import abi

FOLLOW, LEAD, PV, RV = 0, 1, 2, 3

# algorithmic bytes per decision, SURVEY.md §8(d) (state fields at natural width, each touched once)
def algorithmic_bytes(kind, n, followers):
    kind = np.asarray(kind)
    n = np.asarray(n).astype(np.int64)
    out = np.zeros(kind.shape, dtype=np.int64)
    ae = kind == abi.EV_AE_REQ
    out[ae] = np.where(n[ae] == 0, 128, 144 + 16 * n[ae])
    out[(kind == abi.EV_AE_ACK) | (kind == abi.EV_IS_ACK)] = 128 + 8 * followers
    out[(kind == abi.EV_RV_REQ) | (kind == abi.EV_PV_REQ)] = 108
    out[(kind == abi.EV_RV_REPLY) | (kind == abi.EV_PV_REPLY)] = 48
    out[kind == abi.EV_TIMEOUT] = 72
    return out


DECISION_KINDS = (abi.EV_AE_REQ, abi.EV_AE_ACK, abi.EV_IS_ACK, abi.EV_RV_REQ, abi.EV_PV_REQ, abi.EV_RV_REPLY,
                  abi.EV_PV_REPLY, abi.EV_TIMEOUT)


def is_decision(kind):
    kind = np.asarray(kind)
    return (kind >= abi.EV_AE_REQ) & (kind <= abi.EV_TIMEOUT)


@dataclass(frozen=True)
class ReplayConfig:
    name: str
    groups: int
    cluster: int
    seed: int
    leader_frac: float = 0.2      # share of groups that start in leader view
    p_higher_term: float = 0.01   # events that carry a higher term (forces a role arbitration)
    p_timeout: float = 0.0025     # follower election timeouts per round
    p_vote_req: float = 0.002     # RequestVote / PreVote requests reaching a follower from some other candidate
    p_q5: float = 0.0             # a running candidate receives a higher-term RequestVote (Candidate.java:69-71)
    stale_cand_frac: float = 0.2  # RequestVote senders whose log is behind
    role_sorted: bool = False     # experiment: the leader-view groups are the FIRST leader_frac of the slots instead of a hash of the group
                                  # id — what a host gets that keeps the groups it leads in slots of their own (wavefronts become single-role)
    grant_prob: float = 0.9
    p_client: float = 0.2         # leader rounds that are client appends rather than acks
    p_reject: float = 0.02
    p_conflict: float = 0.0       # higher-term AppendEntries (a subset of p_higher_term) whose entries OVERWRITE the last 1-3 uncommitted
                                  # entries of the follower: prevLogIndex below the tail, conflict -> truncate -> append (general handlers)
    p_miss: float = 0.0           # adverse mix (bench.py's value_adverse_mix): follower rows that are a RETRANSMITTED heartbeat of the current leader whose
                                  # prevLogIndex lies below the device's cached term runs (the logs then start with five runs) -> RG_NEED_HOST. The row
                                  # changes nothing whether it is applied or not (but Follower.currentLeader where that was still null); the host PARKS the group (RG_EV_NONE rows) for the rest of the launch —
                                  # what RG_SKIPPED_AFTER_NEED_HOST would do to its rows anyway — and drops the duplicate instead of resubmitting it
    index_base: int = 0           # long-lived groups: every log was compacted at index_base (epoch = (index_base, 1)), all live indices lie above it — the
                                  # stream is the same otherwise. With the table's index bases at index_base - 1 the compact formats carry it (rg_index_base_set)
    ae_entries: tuple = (0, 1, 2, 4)   # entries per AppendEntries request, equiprobable
    self_slot: int = 0
    pre_vote: bool = True


CONFIGS = {
    # BASELINE.json configs[1]: 4 096 RaftContexts x 3 peers, batched AppendEntries quorum (all leader view)
    2: ReplayConfig("config2: 4096 groups x 3 peers, leader-view AppendEntries quorum", 4096, 3, 0xC0FFEE01,
                    leader_frac=1.0, p_higher_term=0.0, p_timeout=0.0, p_vote_req=0.0),
    # the mirrored follower view of configs[1]: every group a follower of a steady leader, AppendEntries requests with
    # n in {0, 1, 2, 4} entries (25 % each), prevLogIndex = lastIndex, leaderCommit = lastIndex - U{0..3}
    "2f": ReplayConfig("config2 (follower view): 4096 groups x 3 peers, AppendEntries requests", 4096, 3, 0xC0FFEE01,
                       leader_frac=0.0, p_higher_term=0.0, p_timeout=0.0, p_vote_req=0.0),
    # configs[2]: 65 536 x 5, mixed leader/follower roles — the configuration the metric is quoted on
    3: ReplayConfig("config3: 65536 groups x 5 peers, mixed leader/follower roles", 65536, 5, 0xC0FFEE02),
    # configs[3]: 1 M x 5 sharded over 8 GPUs, same mix
    4: ReplayConfig("config4: 1048576 groups x 5 peers, mixed roles, block-sharded", 1 << 20, 5, 0xC0FFEE03),
    # configs[4]: 1 M x 5 with leader churn: batched PreVote + RequestVote tally
    5: ReplayConfig("config5: 1048576 groups x 5 peers, leader churn (PreVote + RequestVote tally)", 1 << 20, 5,
                    0xC0FFEE04, p_timeout=0.05, p_vote_req=0.01, p_q5=0.01, grant_prob=0.7),
}


def config(number, groups=None):
    c = CONFIGS[number]
    return c if groups is None else replace(c, groups=groups, name=c.name + " [groups=%d]" % groups)


def _mix(x):
    """splitmix64 finaliser on uint64 arrays"""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class ReplayGenerator:
    """Stream of dense batches for groups [first_gid, first_gid+count) of a config."""

    def __init__(self, cfg, first_gid=0, count=None):
        self.cfg = cfg
        self.first = first_gid
        self.n = cfg.groups - first_gid if count is None else count
        assert 0 < self.n and first_gid + self.n <= cfg.groups
        self.P, self.F = cfg.cluster, cfg.cluster - 1
        self.majority = cfg.cluster // 2 + 1
        self.round_no = 0
        self.gkey = _mix(np.uint64(cfg.seed) ^ (np.arange(first_gid, first_gid + self.n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)))
        self.others = np.array([s for s in range(self.P) if s != cfg.self_slot], dtype=np.int64)
        self._init_model()

    # -- randomness ------------------------------------------------------------------------------
    def _u(self, draw):
        """uniform [0,1) per group for (current round, draw#)"""
        with np.errstate(over="ignore"):
            k = self.gkey ^ (np.uint64(self.round_no + 1) * np.uint64(0xA24BAED4963EE407)) ^ (np.uint64(draw + 1) * np.uint64(0x9FB21C651E98DF25))
        return (_mix(k) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def _ri(self, draw, n):
        """uniform integer in [0, n) per group"""
        return np.minimum((self._u(draw) * n).astype(np.int64), n - 1)

    # -- model -----------------------------------------------------------------------------------
    def _init_model(self):
        n, F, cfg = self.n, self.F, self.cfg
        self.round_no = -1                       # draws for the initial state live in "round -1"
        lead = (np.arange(self.first, self.first + self.n) < int(cfg.leader_frac * cfg.groups)) if cfg.role_sorted else (self._u(0) < cfg.leader_frac)
        self.mode = np.where(lead, LEAD, FOLLOW).astype(np.int64)
        self.term = 1 + self._ri(1, 8)
        self.last = cfg.index_base + 8 + self._ri(2, 1 << 20)
        self.last_term = self.term.copy()
        self.epoch = np.ones(n, dtype=np.int64)                  # role epoch
        self.leader = self.others[self._ri(3, F)]                # leader slot a follower hears from
        self.commit = self.last - self._ri(4, 4)
        self.match = np.zeros((n, F), dtype=np.int64)
        for j in range(F):
            self.match[:, j] = self.last - self._ri(5 + j, 4)
        lead = self.mode == LEAD
        self.commit[lead] = np.sort(self.match[lead], axis=1)[:, F // 2]
        self.cur_term_start = np.full(n, cfg.index_base + 1, dtype=np.int64)      # first log index written in the current term
        self.tail_run_start = np.full(n, cfg.index_base + 1, dtype=np.int64)      # first index of the run of entries that carries last_term
        self.k = np.zeros(n, dtype=np.int64)                     # replies of the running (pre-)election delivered so far
        self.grants = np.zeros(n, dtype=np.int64)
        self.late_k = np.full(n, F, dtype=np.int64)              # next late RequestVote reply of a won election
        self.late_epoch = np.zeros(n, dtype=np.int64)
        self.needs_append = np.zeros(n, dtype=bool)
        self.parked = np.zeros(n, dtype=bool)                    # p_miss: groups that met a cache miss in the launch being built
        self.miss_rows = 0
        if cfg.p_miss > 0.0:
            # five term runs per log: [1, last - 44] of term t - 4, three one-entry runs, then [last - 40, last] of term t. The device caches the newest
            # four: an index below last0 - 43 is a miss for the rest of the stream (later runs only push the cached window further up)
            self.term = self.term + 4
            self.last = np.maximum(self.last, 64)
            self.last_term = self.term.copy()
            self.commit = np.maximum(self.commit, self.last - 3)
            self.match = np.maximum(self.match, (self.last - 3)[:, None])
            self.tail_run_start = self.last - 40
            self.cur_term_start = self.last - 40
            self.miss_below = self.last - 43
            self.run_starts = np.stack([np.ones(n, dtype=np.int64), self.last - 43, self.last - 42, self.last - 41, self.last - 40], axis=1)
            self.run_terms = self.term[:, None] + np.arange(-4, 1, dtype=np.int64)[None, :]
        self.round_no = 0

    def initial_state(self):
        n, F, cfg = self.n, self.F, self.cfg
        st = abi.GroupState(n, self.P)
        lead = self.mode == LEAD
        st.role[:] = np.where(lead, abi.LEADER, abi.FOLLOWER)
        st.current_term[:] = self.term
        st.voted_for[:] = np.where(lead, cfg.self_slot, self.leader)
        st.current_leader[:] = np.where(lead, abi.NO_NODE, self.leader)
        st.repl_prepared[:] = lead
        st.commit_index[:] = self.commit
        st.first_index[:] = cfg.index_base + 1
        st.last_index[:] = self.last
        st.run_count[:] = 1
        st.run_start[0::abi.TERM_RUNS] = cfg.index_base + 1
        st.run_term[0::abi.TERM_RUNS] = self.term
        st.peer_match_index[:] = np.where(lead[:, None], self.match, 0).reshape(-1)
        st.peer_next_index[:] = np.where(lead[:, None], self.match + 1, 0).reshape(-1)
        if cfg.index_base:
            assert cfg.p_miss == 0.0
            st.epoch_index[:] = cfg.index_base
            st.epoch_term[:] = 1
            st.peer_last_epoch[:] = np.repeat(np.where(lead, cfg.index_base, 0), F)       # Leader.prepareReplication: lastEpoch = epoch.index
        if cfg.p_miss > 0.0:                                      # five runs per group: the table keeps the newest RG_TERM_RUNS
            st = abi.GroupState(n, self.P, runs_total=5 * n)
            st.role[:] = np.where(lead, abi.LEADER, abi.FOLLOWER)
            st.current_term[:] = self.term
            st.voted_for[:] = np.where(lead, cfg.self_slot, self.leader)
            st.current_leader[:] = np.where(lead, abi.NO_NODE, self.leader)
            st.repl_prepared[:] = lead
            st.commit_index[:] = self.commit
            st.first_index[:] = 1
            st.last_index[:] = self.last
            st.run_count[:] = 5
            st.run_offset[:] = np.arange(n, dtype=np.uint32) * 5
            st.run_start[:] = self.run_starts.reshape(-1)
            st.run_term[:] = self.run_terms.reshape(-1)
            st.peer_match_index[:] = np.where(lead[:, None], self.match, 0).reshape(-1)
            st.peer_next_index[:] = np.where(lead[:, None], self.match + 1, 0).reshape(-1)
        return st

    # -- one round -------------------------------------------------------------------------------
    def _round(self, b, r):
        cfg, n, F = self.cfg, self.n, self.F
        sl = slice(r * n, (r + 1) * n)
        kind = np.zeros(n, dtype=np.int64)
        slot = np.zeros(n, dtype=np.int64)
        flag = np.zeros(n, dtype=np.int64)
        nent = np.zeros(n, dtype=np.int64)
        aux = np.zeros(n, dtype=np.int64)
        a = np.zeros(n, dtype=np.int64); bb = np.zeros(n, dtype=np.int64)
        c = np.zeros(n, dtype=np.int64); d = np.zeros(n, dtype=np.int64)
        ent_term = np.zeros(n, dtype=np.int64)

        u0, u1, u2, u3 = self._u(0), self._u(1), self._u(2), self._u(3)
        mode = self.mode.copy()                                   # decisions below use the mode at round start
        other = self.others[self._ri(4, F)]
        if cfg.p_miss > 0.0:
            mode[self.parked] = -1                                # parked until the launch ends: RG_EV_NONE rows, no model change
            miss = (mode == FOLLOW) & (self._u(10) < cfg.p_miss)
            mode[miss] = -1
            kind[miss] = abi.EV_AE_REQ                            # the leader's heartbeat once more, with a prevLog far behind: (term, leader, prev, prevTerm, [], 0)
            slot[miss] = self.leader[miss]
            a[miss] = self.term[miss]
            bb[miss] = (self.miss_below - 1 - self._ri(11, 4))[miss]
            c[miss] = self.run_terms[miss, 0]
            self.parked |= miss
            self.miss_rows += int(np.count_nonzero(miss))

        # ---- follower view ----------------------------------------------------------------------
        fol = mode == FOLLOW
        t_out = fol & (u0 < cfg.p_timeout)
        v_req = fol & ~t_out & (u0 < cfg.p_timeout + cfg.p_vote_req)
        ae = fol & ~t_out & ~v_req
        # RaftParticipant.onTimeout at a follower
        kind[t_out] = abi.EV_TIMEOUT
        self.epoch[t_out] += 1
        if cfg.pre_vote:
            self.mode[t_out] = PV
        else:
            self.mode[t_out] = RV
            self.term[t_out] += 1
        self.k[t_out] = 0; self.grants[t_out] = 0
        # a (pre-)vote request from some other candidate
        pre = v_req & (u1 < 0.5) & cfg.pre_vote
        rv = v_req & ~pre
        stale = u2 < cfg.stale_cand_frac
        kind[pre] = abi.EV_PV_REQ; kind[rv] = abi.EV_RV_REQ
        slot[v_req] = other[v_req]
        a[v_req] = self.term[v_req] + 1
        bb[v_req] = np.where(stale[v_req], self.last[v_req] - 1, self.last[v_req])
        c[v_req] = self.last_term[v_req]
        self.term[rv] += 1                                         # Follower.requestVote: term moves even when refused
        self.epoch[rv] += 1
        self.leader[rv] = other[rv]                                # whoever wins leads this term from now on
        # AppendEntries from the group's leader
        hi = ae & (u1 < cfg.p_higher_term)
        self.term[hi] += 1; self.epoch[hi] += 1; self.leader[hi] = other[hi]
        kind[ae] = abi.EV_AE_REQ
        slot[ae] = self.leader[ae]
        nn = np.array(cfg.ae_entries, dtype=np.int64)[self._ri(5, len(cfg.ae_entries))]
        back = np.zeros(n, dtype=np.int64)
        if cfg.p_conflict > 0.0:
            assert cfg.p_conflict <= cfg.p_higher_term, "conflicting AppendEntries come from a NEW leader: p_conflict <= p_higher_term"
            # the new leader's log diverges `back` entries before this follower's tail: it sends prev = last - back and at least one entry
            # of its own term; the entries being replaced are uncommitted and belong to the follower's last run (so prevLogTerm is known)
            want = 1 + np.minimum((u3 * 3).astype(np.int64), 2)
            room = np.minimum(self.last - self.commit, self.last - self.tail_run_start)
            back = np.where(hi & (u1 < cfg.p_conflict), np.minimum(want, np.maximum(room, 0)), 0)
            nn = np.where(back > 0, np.maximum(nn, 1), nn)
        nent[ae] = nn[ae]
        prev = self.last - back
        a[ae] = self.term[ae]; bb[ae] = prev[ae]; c[ae] = self.last_term[ae]
        ent_term[ae] = self.term[ae]
        new_last = prev + np.where(ae, nn, 0)
        lc = np.maximum(self.commit, new_last - self._ri(6, 4))
        d[ae] = lc[ae]
        grew = ae & (nn > 0)
        newrun = grew & (self.last_term != self.term)
        self.tail_run_start[newrun] = prev[newrun] + 1
        self.last_term[grew] = self.term[grew]
        self.last[ae] = new_last[ae]
        self.commit[ae] = np.maximum(self.commit[ae], np.minimum(lc[ae], new_last[ae]))

        # ---- leader view ------------------------------------------------------------------------
        lead = mode == LEAD
        late = lead & (self.late_k < F)                            # stragglers of the election this leader won
        kind[late] = abi.EV_RV_REPLY
        slot[late] = self.others[np.minimum(self.late_k[late], F - 1)]
        flag[late] = u1[late] < cfg.grant_prob
        a[late] = self.term[late]
        aux[late] = self.late_epoch[late]
        self.late_k[late] += 1
        cl = lead & ~late & (self.needs_append | (u0 < cfg.p_client))
        ncmd = np.where(self.needs_append, 1, 1 + self._ri(7, 4))
        kind[cl] = abi.EV_CLIENT_APPEND
        nent[cl] = ncmd[cl]
        fresh = cl & self.needs_append
        self.cur_term_start[fresh] = self.last[fresh] + 1
        newrun_cl = cl & (self.last_term != self.term)
        self.tail_run_start[newrun_cl] = self.last[newrun_cl] + 1
        self.last[cl] += ncmd[cl]
        self.last_term[cl] = self.term[cl]
        self.needs_append[cl] = False
        ack = lead & ~late & ~cl
        j = self._ri(8, F)
        mj = self.match[np.arange(n), j]
        sent = np.where(mj == 0, self.last, np.minimum(mj + 1 + self._ri(9, 4), self.last))
        ok = u2 >= cfg.p_reject
        down = ack & (u1 < cfg.p_higher_term)
        kind[ack] = abi.EV_AE_ACK
        pj = self.others[j]
        slot[ack] = pj[ack]
        flag[ack] = ok[ack]
        a[ack] = np.where(down[ack], self.term[ack] + 1, self.term[ack])
        bb[ack] = cfg.index_base                                  # epoch.index at send: no compaction in this model
        c[ack] = sent[ack]
        aux[ack] = self.epoch[ack]
        adv = ack & ~down & ok
        rows = np.flatnonzero(adv)
        self.match[rows, j[rows]] = np.maximum(mj[rows], sent[rows])
        srt = np.sort(self.match[rows], axis=1)
        major = srt[:, F // 2]
        good = major >= self.cur_term_start[rows]                 # Leader.tryCommit: only current-term entries here
        self.commit[rows[good]] = np.maximum(self.commit[rows[good]], major[good])
        # a higher-term ack ends the leadership: Follower(result.term, responder)
        self.term[down] += 1; self.epoch[down] += 1
        self.mode[down] = FOLLOW
        self.leader[down] = other[down]
        self.late_k[down] = F

        # ---- pre-vote fan-in (Follower.prepareElection) and election fan-in (Candidate.startElection) --
        for m_, k_reply in ((PV, abi.EV_PV_REPLY), (RV, abi.EV_RV_REPLY)):
            el = mode == m_
            q5 = el & (m_ == RV) & (u3 < cfg.p_q5)                 # a rival's higher-term RequestVote is granted blindly
            kind[q5] = abi.EV_RV_REQ
            slot[q5] = other[q5]
            a[q5] = self.term[q5] + 1; bb[q5] = 0; c[q5] = 0
            self.term[q5] += 1; self.epoch[q5] += 1
            self.mode[q5] = FOLLOW; self.leader[q5] = other[q5]
            el = el & ~q5
            again = el & (self.k >= F)                            # every reply is in and it was not enough: time out again
            kind[again] = abi.EV_TIMEOUT
            self.epoch[again] += 1
            if m_ == RV:
                self.term[again] += 1
            self.k[again] = 0; self.grants[again] = 0
            rep = el & ~again
            kind[rep] = k_reply
            slot[rep] = self.others[np.minimum(self.k[rep], F - 1)]
            g_ = u1 < cfg.grant_prob
            flag[rep] = g_[rep]
            a[rep] = self.term[rep]
            aux[rep] = self.epoch[rep]
            self.k[rep] += 1
            self.grants[rep & g_] += 1
            won = rep & (self.grants >= self.majority - 1)
            if m_ == PV:                                          # -> Candidate(currentTerm + 1)
                self.term[won] += 1; self.epoch[won] += 1
                self.mode[won] = RV
            else:                                                 # -> Leader; the rest of the replies arrive late (Q13)
                self.late_epoch[won] = self.epoch[won]
                self.late_k[won] = self.k[won]
                self.epoch[won] += 1
                self.mode[won] = LEAD
                self.needs_append[won] = True
                self.match[won] = 0
            self.k[won & (m_ == PV)] = 0
            self.grants[won] = 0

        # ---- pack -------------------------------------------------------------------------------
        is_ae = (kind == abi.EV_AE_REQ) & (nent > 0)
        off = self._ent_count + np.concatenate(([0], np.cumsum(np.where(is_ae, nent, 0))[:-1]))
        aux = np.where(kind == abi.EV_AE_REQ, np.where(is_ae, off, 0), aux)
        tot = int(np.where(is_ae, nent, 0).sum())
        if tot:
            self._ents.append(np.repeat(ent_term[is_ae], nent[is_ae]))
            self._ent_count += tot
        b.head["hdr"][sl] = abi.hdr_make(kind, slot, flag, nent)
        b.head["aux"][sl] = aux.astype(np.uint32)
        b.ab["x"][sl] = a; b.ab["y"][sl] = bb
        b.cd["x"][sl] = c; b.cd["y"][sl] = d
        self.round_no += 1

    def next_batch(self, rounds):
        """The next `rounds` rounds as one dense multi-round batch (entry offsets local to the batch)."""
        b = abi.Batch(rounds, self.n)
        self._ents, self._ent_count = [], 0
        self.parked[:] = False
        for r in range(rounds):
            self._round(b, r)
        b.entry_terms = np.concatenate(self._ents) if self._ents else np.zeros(0, dtype=np.int64)
        b.entry_count = self._ent_count
        return b


def batch_stats(batch, followers):
    """(#decisions, algorithmic bytes, per-kind counts) of a batch, per SURVEY.md §8(d)."""
    hdr = batch.head["hdr"]
    kind = hdr & 0xF
    n = hdr >> 12
    dec = is_decision(kind)
    nbytes = int(algorithmic_bytes(kind, n, followers)[dec].sum())
    return int(np.count_nonzero(dec)), nbytes, np.bincount(kind, minlength=11)
