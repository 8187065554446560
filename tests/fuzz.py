"""State-aware random event generator for differential replay (test infrastructure).

Each round it looks at the current state of every group (as reported by read_state of the table under
test — so the term-run floor it respects is the device's own) and draws ONE plausible event per group:
mostly protocol-conformant traffic, salted with stale terms, wrong prevLogTerms, conflicting entries,
fenced responses and the occasional assertion trigger, so that every branch of the decision path and
every status code is visited.
"""
import random

import numpy as np

from rafting_amd import abi

F, C, L = abi.FOLLOWER, abi.CANDIDATE, abi.LEADER


class View:
    """Per-group python view of a GroupState image."""

    def __init__(self, st, g):
        K, Fn = abi.TERM_RUNS, st.followers
        self.role, self.term, self.vote = int(st.role[g]), int(st.current_term[g]), int(st.voted_for[g])
        self.leader, self.td = int(st.current_leader[g]), int(st.timeout_detected[g])
        self.prepared, self.epoch = int(st.repl_prepared[g]), int(st.role_epoch[g])
        self.elected_epoch = int(st.elected_epoch[g])
        self.commit = int(st.commit_index[g])
        self.eidx, self.eterm = int(st.epoch_index[g]), int(st.epoch_term[g])
        rc = int(st.run_count[g])
        self.runs = [(int(st.run_start[g * K + k]), int(st.run_term[g * K + k])) for k in range(rc)]
        self.first, self.last = int(st.first_index[g]), int(st.last_index[g])
        self.has_log = rc > 0
        self.floor = self.runs[0][0] if rc else None
        self.peers = [(int(st.peer_last_epoch[g * Fn + j]), int(st.peer_next_index[g * Fn + j]),
                       int(st.peer_match_index[g * Fn + j]), int(st.peer_pending[g * Fn + j])) for j in range(Fn)]

    def term_at(self, i):
        """term of a cached index (None when unknown / absent)"""
        if not self.has_log or i < self.floor or i > self.last:
            return None
        t = None
        for s, tt in self.runs:
            if s <= i:
                t = tt
        return t

    @property
    def last_term(self):
        return self.runs[-1][1] if self.runs else None


class Fuzzer:
    def __init__(self, groups, cluster, self_slot, seed, assert_rate=0.01, allow_miss=True):
        self.G, self.P, self.self_slot = groups, cluster, self_slot
        self.rng = random.Random(seed)
        self.others = [s for s in range(cluster) if s != self_slot]
        self.assert_rate = assert_rate
        self.allow_miss = allow_miss     # False: never draw an event whose lookups leave the cached term runs

    # -- building blocks -------------------------------------------------------------------------
    def _ae(self, b, r, g, v):
        rng = self.rng
        term = rng.choices([v.term - 1, v.term, v.term + 1, v.term + 3], [4, 80, 12, 4])[0]
        term = max(term, 0)
        leader = v.leader if (v.leader != abi.NO_NODE and rng.random() < 0.9) else rng.choice(self.others)
        if rng.random() < self.assert_rate:
            leader = rng.randrange(self.P)
        # prevLog
        choice = rng.random()
        if not v.has_log:
            prev = rng.choice([v.eidx, v.eidx, 0, max(v.eidx - 1, 0), v.eidx + 1])
        elif choice < 0.70:
            prev = v.last
        elif choice < 0.85:
            prev = rng.randint(max(v.floor, v.last - 6), v.last)
        elif choice < 0.92:
            prev = v.last + rng.randint(1, 3)
        else:
            prev = rng.choice([v.eidx, max(v.eidx - 1, 0), 0, v.eidx + 1])
            if prev > v.eidx and prev < v.floor:
                prev = v.floor
        prev = max(prev, 0)
        incomplete = v.has_log and v.floor > v.first
        if incomplete and v.eidx < prev < v.floor and not self.allow_miss:
            prev = v.floor
        true_t = v.term_at(prev)
        if prev == 0:
            pterm = 0 if rng.random() > self.assert_rate else 3
        elif prev <= v.eidx:
            pterm = v.eterm if (prev == v.eidx and rng.random() > self.assert_rate) else rng.randint(1, 9)
            if prev == v.eidx and pterm == 0:
                pterm = 1
            if prev < v.eidx and pterm == 0:
                pterm = 1
        elif true_t is not None:
            pterm = true_t if rng.random() < 0.93 else true_t + rng.choice([-1, 1])
            pterm = max(pterm, 1)
        else:
            pterm = rng.randint(1, 9)
        # entries
        n = rng.choices([0, 1, 2, 3, 5], [35, 30, 20, 10, 5])[0]
        if incomplete and prev + 1 < v.floor and not self.allow_miss:
            n = 0                                            # the conflict scan would leave the cached runs
        entries = []
        base_t = max(term, 1)
        for k in range(n):
            idx = prev + 1 + k
            t_here = v.term_at(idx)
            if t_here is not None and rng.random() < 0.9:
                entries.append(t_here)                      # duplicate of what is stored
            elif idx <= v.eidx:
                entries.append(max(v.eterm, 1))
            else:
                lo = entries[-1] if entries else (true_t if true_t is not None else max(pterm, 1))
                lo = max(lo, 1)
                entries.append(rng.choice([lo, lo, lo, min(lo + 1, max(base_t, lo)), base_t if base_t >= lo else lo]))
        top = prev + n
        lc = rng.choices([v.commit, v.commit + rng.randint(0, 3), top + 2, v.eidx, max(v.commit - 1, 0)],
                         [30, 45, 15, 8, 2 if rng.random() < 5 * self.assert_rate else 0])[0]
        b.put(r, g, abi.EV_AE_REQ, slot=leader, a=term, b=prev, c=pterm, d=lc, entries=entries)

    def _vote_req(self, b, r, g, v, pre):
        rng = self.rng
        term = rng.choices([v.term - 1, v.term, v.term + 1, v.term + 2], [5, 20, 60, 15])[0]
        term = max(term, 0)
        cand = rng.choice(self.others) if rng.random() > self.assert_rate else rng.randrange(self.P)
        if v.has_log:
            li = v.last + rng.choice([-2, -1, 0, 0, 0, 1, 4])
            lt = v.last_term + rng.choice([-1, 0, 0, 0, 1])
        else:
            li = v.eidx + rng.choice([-1, 0, 0, 1, 3])
            lt = v.eterm + rng.choice([0, 0, 0, 1]) - (1 if rng.random() < self.assert_rate else 0)
        b.put(r, g, abi.EV_PV_REQ if pre else abi.EV_RV_REQ, slot=cand, a=term, b=max(li, 0), c=max(lt, 0))

    def _ack(self, b, r, g, v):
        rng = self.rng
        peer = rng.choice(self.others)
        j = peer if peer < self.self_slot else peer - 1
        le, nx, ma, pend = v.peers[j]
        resp = v.term if rng.random() < 0.97 else v.term + rng.randint(1, 2)
        epoch = v.epoch if rng.random() < 0.95 else max(v.epoch - 1, 0)
        at_send = rng.choices([v.eidx, le, max(le - 1, 0), v.eidx + 1], [70, 20, 5, 5])[0]
        if pend and rng.random() < 0.7:
            b.put(r, g, abi.EV_IS_ACK, slot=peer, flag=int(rng.random() < 0.8), a=resp, b=at_send, aux=epoch)
            return
        top = v.last if v.has_log else v.eidx
        floor = v.floor if v.has_log else 0
        lo = max(ma, floor, 1)
        hi = max(top, lo)
        sent = rng.randint(lo, hi) if rng.random() < 0.93 else hi + rng.randint(1, 3)
        if rng.random() < self.assert_rate and ma > floor + 1:
            sent = ma - 1                                    # A_MATCH_ROLLBACK
        b.put(r, g, abi.EV_AE_ACK, slot=peer, flag=int(rng.random() < 0.85), a=resp, b=at_send, c=sent, aux=epoch)

    def _vote_reply(self, b, r, g, v, pre, epoch=None):
        rng = self.rng
        peer = rng.choice(self.others)
        t = v.term + (1 if pre else 0)
        resp = t if rng.random() < 0.9 else t + rng.randint(-1, 2)
        epoch = v.epoch if epoch is None else epoch
        if rng.random() < 0.05:
            epoch = max(epoch - 1, 0)
        b.put(r, g, abi.EV_PV_REPLY if pre else abi.EV_RV_REPLY, slot=peer, flag=int(rng.random() < 0.7),
              a=max(resp, 0), aux=epoch)

    def _is_req(self, b, r, g, v):
        rng = self.rng
        term = rng.choices([v.term - 1, v.term, v.term + 1], [15, 75, 10])[0]
        b.put(r, g, abi.EV_IS_REQ, slot=rng.choice(self.others), flag=int(rng.random() < 0.8), a=max(term, 0),
              b=v.eidx + rng.randint(0, 40), c=max(v.eterm, 1))

    def _timeout(self, b, r, g, v):
        """the ticket that fired: usually the live participant's (or 0 = host-owned timers), sometimes a replaced one's"""
        x = self.rng.random()
        b.put(r, g, abi.EV_TIMEOUT, aux=0 if x < 0.4 else (v.epoch if x < 0.9 else max(v.epoch - 1, 1)))

    def _flush(self, b, r, g, v):
        rng = self.rng
        if v.has_log and v.commit >= v.floor and v.commit <= v.last and rng.random() < 0.8:
            idx = rng.randint(max(v.floor, v.eidx), max(v.commit, max(v.floor, v.eidx)))
            idx = min(idx, v.last)
            t = v.term_at(idx)
            if t is not None and idx >= v.eidx:
                b.put(r, g, abi.EV_LOG_FLUSH, a=idx, b=t)
                return
        if rng.random() < 0.3:
            top = (v.last if v.has_log else v.eidx) + rng.randint(1, 4)     # snapshot install past the end
            b.put(r, g, abi.EV_LOG_FLUSH, a=top, b=max(v.term, 1))
        elif rng.random() < 0.2:
            b.put(r, g, abi.EV_LOG_FLUSH, a=max(v.eidx - 1, 0), b=1)         # out of bounds unless eidx == 0

    # -- one round -------------------------------------------------------------------------------
    def round(self, st, b, r):
        """Fill round r of batch b from state image st."""
        rng = self.rng
        for g in range(self.G):
            v = View(st, g)
            x = rng.random()
            if x < 0.04:
                continue                                      # RG_EV_NONE
            if x < 0.07:
                self._flush(b, r, g, v)
                continue
            if x < 0.085:
                self._is_req(b, r, g, v)
                continue
            if v.role == F:
                if x < 0.62:
                    self._ae(b, r, g, v)
                elif x < 0.74:
                    self._vote_req(b, r, g, v, pre=False)
                elif x < 0.82:
                    self._vote_req(b, r, g, v, pre=True)
                elif x < 0.90:
                    self._timeout(b, r, g, v)
                elif x < 0.97 and v.td:
                    self._vote_reply(b, r, g, v, pre=True)
                elif x < 0.985:
                    b.put(r, g, abi.EV_CLIENT_APPEND, n=1)   # NOT_LEADER
                elif v.elected_epoch:
                    self._vote_reply(b, r, g, v, pre=False, epoch=v.elected_epoch)
                else:
                    self._ack(b, r, g, v)                     # stale / BAD_EVENT
            elif v.role == C:
                if x < 0.55:
                    self._vote_reply(b, r, g, v, pre=False)
                elif x < 0.70:
                    self._ae(b, r, g, v)
                elif x < 0.85:
                    self._vote_req(b, r, g, v, pre=rng.random() < 0.4)
                elif x < 0.95:
                    self._timeout(b, r, g, v)
                else:
                    self._vote_reply(b, r, g, v, pre=True)    # fenced pre-vote reply
            else:
                if not v.prepared:
                    if x < 0.5:
                        self._timeout(b, r, g, v)
                    elif x < 0.8:
                        b.put(r, g, abi.EV_CLIENT_APPEND, n=rng.randint(1, 3))
                    elif v.elected_epoch and x < 0.9:
                        self._vote_reply(b, r, g, v, pre=False, epoch=v.elected_epoch)
                    else:
                        self._ack(b, r, g, v)                 # BAD_EVENT: nothing was sent yet
                elif x < 0.60:
                    self._ack(b, r, g, v)
                elif x < 0.75:
                    b.put(r, g, abi.EV_CLIENT_APPEND, n=rng.randint(1, 3))
                elif x < 0.80:
                    self._timeout(b, r, g, v)
                elif x < 0.88:
                    self._ae(b, r, g, v)
                elif x < 0.95:
                    self._vote_req(b, r, g, v, pre=rng.random() < 0.3)
                elif v.elected_epoch:
                    self._vote_reply(b, r, g, v, pre=False, epoch=v.elected_epoch)
                else:
                    self._vote_reply(b, r, g, v, pre=False)


def random_initial_state(groups, cluster, self_slot, seed, offset=0):
    """Mixed-role start: followers/candidates/leaders with logs of 1-3 term runs, epochs, peers. offset > 0: long-lived groups — every log was
    compacted at or above `offset` (epoch.index >= offset), so all their live indices are huge while their spans stay small."""
    rng = random.Random(seed ^ 0x5EED)
    st = abi.GroupState(groups, cluster)
    Fn = cluster - 1
    for g in range(groups):
        role = rng.choices([F, C, L], [60, 10, 30])[0]
        term = rng.randint(1, 8)
        eidx = offset + rng.choice([0, 0, rng.randint(1, 50)])
        eterm = 0 if eidx == 0 else rng.randint(1, term)
        st.role[g], st.current_term[g] = role, term
        st.epoch_index[g], st.epoch_term[g] = eidx, eterm
        st.role_epoch[g] = rng.randint(1, 5)
        if role == F:
            st.voted_for[g] = rng.choice([abi.NO_NODE] + list(range(cluster)))
            st.current_leader[g] = rng.choice([abi.NO_NODE] + [s for s in range(cluster) if s != self_slot])
        else:
            st.voted_for[g] = self_slot
        if rng.random() < 0.9:
            first = eidx + (0 if (eidx > 0 and rng.random() < 0.3) else 1)
            t = max(eterm, 1) if first == eidx else rng.randint(max(eterm, 1), term)
            runs, start = [], first
            for _ in range(rng.randint(1, 3)):
                runs.append((start, t))
                start += rng.randint(1, 30)
                if t >= term:
                    break
                t += rng.randint(1, min(2, term - t))
            last = start - 1
            st.set_log(g, first, runs, last)
            st.commit_index[g] = rng.randint(eidx, last)
        if role == L and rng.random() < 0.8:
            st.repl_prepared[g] = 1
            top = int(st.last_index[g]) if st.run_count[g] else eidx
            for j in range(Fn):
                m = rng.choice([0, rng.randint(max(eidx - 30, 0) if offset else 0, top)])
                st.peer_last_epoch[g * Fn + j] = eidx
                st.peer_match_index[g * Fn + j] = m
                st.peer_next_index[g * Fn + j] = (m + 1) if m else top + 1
    return st


def concat_batches(batches):
    """Stack single-round dense batches into one multi-round batch (entry offsets rebased)."""
    count = batches[0].count
    out = abi.Batch(len(batches), count)
    ents, off = [], 0
    for r, b in enumerate(batches):
        sl = slice(r * count, (r + 1) * count)
        out.head[sl] = b.head
        out.ab[sl] = b.ab
        out.cd[sl] = b.cd
        is_ae = (b.head["hdr"] & 0xF) == abi.EV_AE_REQ
        aux = out.head["aux"][sl]
        aux[is_ae] = aux[is_ae] + np.uint32(off)
        out.head["aux"][sl] = aux
        ents.append(b.entry_terms[: b.entry_count])
        off += b.entry_count
    out.entry_terms = np.concatenate(ents) if ents else np.zeros(0, dtype=np.int64)
    out.entry_count = off
    return out


def concat_outcomes(outs):
    o = abi.Outcome(0)
    o.reply = np.concatenate([x.reply for x in outs])
    o.logfx = np.concatenate([x.logfx for x in outs])
    o.persist = np.concatenate([x.persist for x in outs])
    return o
