/*
 * raft_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See raft_oracle.h (incl. how it is pinned to the
 * reference's own compiled sources, oracle/_ref).
 *
 * Every function cites the reference lines it restates.  Paths are relative to
 *   /root/reference/src/main/java/io/lubricant/consensus/raft/
 * The reference is read, not copied: the Java keeps one object graph per RaftContext and runs
 * handlers on an EventLoop thread; here one `group_t` holds the same fields and `step()` plays
 * one EventLoop task.  Quirks Q1-Q12 of SURVEY.md §8a are reproduced on purpose, plus
 *   Q13  a Candidate that WON keeps its election AsyncHead un-aborted (Candidate.onFencing skips
 *        the abort when `elected`, member/Candidate.java:75-79), so late RequestVote replies are
 *        still processed by the new Leader (and whatever it becomes later);
 *   Q14  RaftLog.newEntry writes index 1 on an empty log whatever the epoch
 *        (storage/RocksLog.java:83-84) — reported as RG_UNSUPPORTED_LOG_STATE when epoch.index>0.
 */
#include "raft_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------- */
/* lossless log: contiguous key window [first,last] as maximal equal-term runs                  */

typedef struct { int64_t start, term; } run_t;
typedef struct { int64_t first, last; uint32_t n, cap; run_t *r; } olog_t;

typedef struct {                 /* Leadership.State, member/Leadership.java:26-38 */
    int64_t last_epoch, next_index, match_index;
    int32_t rejection;
    uint8_t pending;
    int64_t request_success, request_failure;   /* N4b: health statistics, member/Leadership.java:30-34 */
    int32_t recent_failure;
} peer_t;

typedef struct {
    int64_t  current_term;       /* RaftMember.currentTerm  member/RaftMember.java:17 */
    int32_t  voted_for;          /* RaftMember.lastCandidate :18 */
    int32_t  role;
    int32_t  current_leader;     /* Follower.currentLeader  member/Follower.java:21 */
    uint8_t  timeout_detected;   /* Follower.timeoutDetected :23 */
    uint8_t  repl_prepared;      /* Leader.followerStatus != null  member/Leader.java:31 */
    uint32_t role_epoch;
    int32_t  votes;
    uint32_t elected_epoch;
    int64_t  elected_term;
    int64_t  commit_index;       /* RocksLog.commitIndex  storage/RocksLog.java:50 */
    int64_t  epoch_index, epoch_term;
    olog_t   log;
    peer_t   peers[RG_MAX_CLUSTER - 1];
} group_t;

struct orc_table {
    int64_t *deadline;           /* N4: per-group timer ticket: 0 none, -1 fired (TimerTicket.TIMEOUT), >0 armed */
    int64_t election_ms, heartbeat_ms;
    uint64_t timer_seed;
    uint32_t groups, cluster, self, followers;
    int      pre_vote;
    int      require_fence;      /* RG_OPT_REQUIRE_FENCED_TIMEOUTS (a contract of the C-ABI, not reference behaviour): TIMEOUT rows with aux == 0 are RG_BAD_EVENT */
    int      majority;           /* RaftContext.majority()  context/RaftContext.java:170 */
    const int64_t *clock;        /* N4b: System.currentTimeMillis() per round of the next orc_submit (orc_health_clock) */
    group_t *g;
};

typedef struct {                 /* what one event task produced */
    uint32_t flags;
    uint32_t status;
    int64_t  resp_term;
    int64_t  log_from;
    int64_t  now;                /* N4b: wall clock of this round; valid when timed */
    int      timed;
} fx_t;

static inline int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
static inline int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }

static int log_empty(const olog_t *l) { return l->n == 0; }

static void log_reserve(olog_t *l, uint32_t n)
{
    if (n <= l->cap) return;
    uint32_t c = l->cap ? l->cap * 2 : 4;
    while (c < n) c *= 2;
    l->r = (run_t *)realloc(l->r, (size_t)c * sizeof(run_t));
    if (!l->r) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    l->cap = c;
}

/* RaftLog.get(index).term(): storage/RocksLog.java:122-128. returns 1 and *term when the key exists */
static int log_get(const olog_t *l, int64_t index, int64_t *term)
{
    if (l->n == 0 || index < l->first || index > l->last) return 0;
    uint32_t j = l->n;
    while (j > 0 && l->r[j - 1].start > index) j--;
    if (j == 0) { fprintf(stderr, "oracle: run table corrupt\n"); abort(); }
    *term = l->r[j - 1].term;
    return 1;
}

/* db.put(index, term) for index == last+1, or the first key of an empty log */
static void log_push(olog_t *l, int64_t index, int64_t term)
{
    if (l->n == 0) {
        log_reserve(l, 1);
        l->first = l->last = index;
        l->r[0].start = index; l->r[0].term = term; l->n = 1;
        return;
    }
    if (index != wadd(l->last, 1)) { fprintf(stderr, "oracle: non-contiguous push\n"); abort(); }
    if (l->r[l->n - 1].term != term) {
        log_reserve(l, l->n + 1);
        l->r[l->n].start = index; l->r[l->n].term = term; l->n++;
    }
    l->last = index;
}

/* RaftLog.truncate(index): storage/RocksLog.java:219-225 */
static void log_truncate(olog_t *l, int64_t index)
{
    if (l->n == 0 || l->last < index) return;
    if (index <= l->first) { l->n = 0; return; }
    while (l->n > 0 && l->r[l->n - 1].start >= index) l->n--;
    l->last = index - 1;
}

/* RaftLog.flush(index, term): storage/RocksLog.java:228-242 — deleteRange [epochIndex, index) */
static int log_flush(group_t *g, int64_t index, int64_t term)
{
    if (index < g->epoch_index) return RG_FLUSH_OUT_OF_BOUNDS;
    olog_t *l = &g->log;
    if (l->n != 0) {
        if (index > l->last) {
            l->n = 0;
        } else if (index > l->first) {
            uint32_t drop = 0;     /* runs that end before `index` */
            while (drop + 1 < l->n && l->r[drop + 1].start <= index) drop++;
            if (drop) { memmove(l->r, l->r + drop, (size_t)(l->n - drop) * sizeof(run_t)); l->n -= drop; }
            l->r[0].start = index;
            l->first = index;
        }
    }
    g->epoch_index = index;
    g->epoch_term = term;
    return RG_OK;
}

/* RaftLog.conflict(entries): storage/RocksLog.java:199-216. entries k has index e0+k. returns conflict index or 0 */
static int64_t log_conflict(const olog_t *l, int64_t e0, uint32_t n, const int64_t *terms)
{
    for (uint32_t k = 0; k < n; k++) {
        int64_t idx = wadd(e0, k), t;
        if (!log_get(l, idx, &t)) return 0;        /* NOT_FOUND: everything before matched */
        if (t != terms[k]) return idx;
    }
    return 0;
}

/* RaftLog.append(entries): storage/RocksLog.java:169-196 */
static int log_append(group_t *g, int64_t e0, uint32_t n, const int64_t *terms)
{
    olog_t *l = &g->log;
    int64_t prev_log_index = g->epoch_index;
    int valid = 0;
    if (l->n != 0 && l->first <= e0) {             /* iterator.seekForPrev(entries[0].index) */
        prev_log_index = min64(l->last, e0);
        valid = 1;
    }
    if (!valid && e0 != wadd(prev_log_index, 1)) return RG_A_LOG_NOT_CONTINUOUS;   /* :175-177 */
    int have_prev = 0; int64_t prev_index = 0;
    for (uint32_t k = 0; k < n; k++) {
        int64_t idx = wadd(e0, k);
        if (idx > prev_log_index) {
            if (!have_prev || prev_index == prev_log_index) {
                if (prev_log_index != wsub(idx, 1)) return RG_A_LOG_NOT_CONTINUOUS;  /* :184-188 */
            }
            int64_t t;
            if (log_get(l, idx, &t)) {
                /* overwrite of an existing key: conflict()+truncate() ran first, so it carries the same term */
                if (t != terms[k]) { fprintf(stderr, "oracle: append over a different term\n"); abort(); }
            } else {
                if (l->n != 0 && idx != wadd(l->last, 1)) { fprintf(stderr, "oracle: append leaves a gap\n"); abort(); }
                log_push(l, idx, terms[k]);
            }
        }
        have_prev = 1; prev_index = idx;
    }
    return RG_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* Membership.isBetter: member/Membership.java:74-108                                           */

int orc_is_better(int nr, int64_t nt, int32_t nb, int cr, int64_t ct, int32_t cb)
{
    if (nt != ct) return nt > ct;
    if (nr != cr) {
        if (nr == RG_LEADER) {
            if (cr == RG_CANDIDATE) return 1;
            return -RG_A_LEADER_UNCHANGED;
        }
        return nr == RG_FOLLOWER;
    }
    if (nr == RG_LEADER) return 0;
    if (nr == RG_FOLLOWER) return 1;
    if (nb != cb) return -RG_A_CAND_BALLOT;
    return 0;
}

void orc_is_better_batch(uint32_t n, const int32_t *nr, const int64_t *nt, const int32_t *nb, const int32_t *cr, const int64_t *ct,
                         const int32_t *cb, int32_t *out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = orc_is_better(nr[i], nt[i], nb[i], cr[i], ct[i], cb[i]);
}

void orc_major_indices_batch(uint32_t n, int f, const int64_t *match, int64_t *out)
{
    for (uint32_t i = 0; i < n; i++) orc_major_indices(match + (size_t)i * f, f, out + 2 * (size_t)i);
}

/* RaftContext.switchTo/trySwitchTo -> RaftRoutine.trySwitch + switchTo + convertTo
 * (context/RaftContext.java:195-215, context/RaftRoutine.java:140-216) and the constructors of the
 * new participant (member/RaftMember.java:20-26, Follower.java:26-28, Candidate.java:22-25, Leader.java:25-28).
 * returns 1 converted, 0 not better (participant unchanged), -1 assertion (fx->status set). */
static int switch_to(const orc_table_t *t, group_t *g, fx_t *fx, int role, int64_t term, int32_t ballot)
{
    int b = orc_is_better(role, term, ballot, g->role, g->current_term, g->voted_for);
    if (b < 0) { fx->status = (uint32_t)(-b); return -1; }
    if (!b) return 0;
    g->role = role;
    g->current_term = term;
    g->voted_for = ballot;
    g->role_epoch += 1;                    /* new participant object, old AsyncHead fenced */
    g->timeout_detected = 0;
    g->current_leader = RG_NO_NODE;
    g->votes = 1;
    g->repl_prepared = 0;
    if (role == RG_LEADER)                 /* State objects only exist from prepareReplication on (member/Leader.java:30-50); */
        for (uint32_t j = 0; j < t->followers; j++) {   /* a new Leader therefore starts from zeroed statistics */
            g->peers[j].request_success = g->peers[j].request_failure = 0;
            g->peers[j].recent_failure = 0;
        }
    fx->flags |= RG_F_PERSIST | RG_F_ROLE_CHANGED | RG_F_RESET_TIMER;
    fx->flags &= ~RG_F_EMIT_MASK;
    if (role == RG_CANDIDATE) fx->flags |= RG_EMIT_REQVOTE << RG_F_EMIT_SHIFT;   /* startElection */
    return 1;
}

/* Leader.prepareReplication: member/Leader.java:30-50 */
static void prepare_replication(const orc_table_t *t, group_t *g)
{
    if (g->repl_prepared) return;
    int64_t last_index = log_empty(&g->log) ? g->epoch_index : g->log.last;
    for (uint32_t j = 0; j < t->followers; j++) {
        peer_t *s = &g->peers[j];
        s->last_epoch = g->epoch_index;
        s->next_index = wadd(last_index, 1);
        s->match_index = 0;
        s->rejection = 0;
        s->pending = 0;
        s->request_success = s->request_failure = 0;
        s->recent_failure = 0;
    }
    g->repl_prepared = 1;
}

/* Math.round(Math.log(Math.E + recentRejection)): member/Leadership.java:105.
 * Integer thresholds: step >= k  <=>  r >= ceil(exp(k-0.5) - e). (tests/test_oracle_kat.py re-derives them with math.log) */
int64_t orc_rejection_step(int32_t r)
{
    static const int32_t lo[] = { /* smallest r with step == index+2 */
        2, 10, 31, 88, 242, 663, 1806, 4913, 13358, 36313, 98714, 268335, 729414, 1982757,
        5389696, 14650717, 39824782, 108254986, 294267564, 799902175 };
    if (r < 0) return r == -1 ? 1 : 0;     /* ln(e-1)=0.54 -> 1; ln(e-2)<0.5 -> 0; NaN -> 0 */
    int64_t step = 1;
    for (unsigned i = 0; i < sizeof(lo) / sizeof(lo[0]); i++) if (r >= lo[i]) step = i + 2;
    return step;
}

/* Leadership.State.updateIndex: member/Leadership.java:75-114 */
static int update_index(peer_t *s, int64_t epoch, int64_t index, int success, int snapshot)
{
    if (index < s->match_index) return RG_A_MATCH_ROLLBACK;
    if (epoch < s->last_epoch) return RG_OK;
    if (epoch > s->last_epoch) {
        s->last_epoch = epoch;
        s->next_index = s->next_index > epoch ? s->next_index : epoch;
    }
    if ((s->pending != 0) != (snapshot != 0)) return RG_OK;
    if (s->pending) {
        if (success) {
            s->next_index = max64(s->next_index, wadd(epoch, 1));
            s->pending = 0;
        }
    } else {
        if (success) {
            if (index > s->match_index) {
                s->next_index = wadd(index, 1);
                s->match_index = index;
            }
        } else if (s->match_index == 0) {
            int64_t step = orc_rejection_step(s->rejection);
            int64_t next = max64(wsub(s->next_index, step), wadd(epoch, 1));
            s->next_index = min64(wsub(s->next_index, 1), next);
        }
    }
    if (s->next_index <= epoch && !s->pending) s->pending = 1;
    return RG_OK;
}

/* exposed for differential tests: State.updateIndex on (lastEpoch, nextIndex, matchIndex), recentRejection, pendingInstallation */
int orc_update_index(int64_t st[3], int32_t *rejection, uint8_t *pending, int64_t epoch, int64_t index, int success, int snapshot)
{
    peer_t s;
    memset(&s, 0, sizeof s);
    s.last_epoch = st[0]; s.next_index = st[1]; s.match_index = st[2]; s.rejection = *rejection; s.pending = *pending != 0;
    int rc = update_index(&s, epoch, index, success, snapshot);
    st[0] = s.last_epoch; st[1] = s.next_index; st[2] = s.match_index; *rejection = s.rejection; *pending = s.pending;
    return rc;
}

/* batch forms for differential fuzzing against oracle/_ref (one call, n independent inputs) */
void orc_update_index_batch(uint32_t n, int64_t *st, int32_t *rejection, uint8_t *pending, const int64_t *epoch, const int64_t *index,
                            const uint8_t *success, const uint8_t *snapshot, int32_t *rc)
{
    for (uint32_t i = 0; i < n; i++)
        rc[i] = orc_update_index(st + 3 * (size_t)i, &rejection[i], &pending[i], epoch[i], index[i], success[i], snapshot[i]);
}

/* Leadership.State.majorIndices: member/Leadership.java:116-130 */
void orc_major_indices(const int64_t *match, int n, int64_t out[2])
{
    int64_t s[RG_MAX_CLUSTER];
    for (int i = 0; i < n; i++) s[i] = match[i];
    for (int i = 1; i < n; i++) {              /* Arrays.sort */
        int64_t v = s[i]; int j = i;
        while (j > 0 && s[j - 1] > v) { s[j] = s[j - 1]; j--; }
        s[j] = v;
    }
    out[0] = s[0];
    out[1] = s[n / 2];
}

/* RaftLog.markCommitted: storage/RocksLog.java:100-109 (via RaftContext.commitLog :244-255) */
static int mark_committed(group_t *g, fx_t *fx, int64_t commit_index)
{
    if (commit_index < g->commit_index) return RG_A_COMMIT_ROLLBACK;
    if (commit_index > g->commit_index) {
        g->commit_index = commit_index;
        fx->flags |= RG_F_COMMIT;
    }
    return RG_OK;
}

/* Leader.tryCommit: member/Leader.java:247-280 */
static void try_commit(const orc_table_t *t, group_t *g, fx_t *fx)
{
    int64_t m[RG_MAX_CLUSTER], mi[2];
    for (uint32_t j = 0; j < t->followers; j++) m[j] = g->peers[j].match_index;
    orc_major_indices(m, (int)t->followers, mi);
    int64_t full = mi[0], major = mi[1];
    if (full > major) { fx->status = RG_A_IMPOSSIBLE_REPLICATION; return; }
    if (major == 0) return;
    int64_t mt;
    if (!log_get(&g->log, major, &mt)) { fx->status = RG_NPE_MAJOR_NULL; return; }   /* major.term() on null */
    int64_t commit = (mt == g->current_term) ? major : full;
    if (commit != 0 && commit != g->commit_index) {
        int st = mark_committed(g, fx, commit);
        if (st) fx->status = (uint32_t)st;
    }
}

/* Follower.logContains: member/Follower.java:177-191. returns 1/0, or -1 with fx->status */
static int log_contains(const group_t *g, fx_t *fx, int64_t index, int64_t term)
{
    if (index == 0 && term == 0) return 1;
    if (index == 0 || term == 0) { fx->status = RG_A_PREV_ZERO_MISMATCH; return -1; }
    if (index <= g->epoch_index) {
        if (index == g->epoch_index && term != g->epoch_term) { fx->status = RG_A_EPOCH_TERM_MISMATCH; return -1; }
        return 1;
    }
    int64_t t;
    return log_get(&g->log, index, &t) && t == term;
}

/* Follower.logUpToDate: member/Follower.java:193-207 */
static int log_up_to_date(const group_t *g, fx_t *fx, int64_t index, int64_t term)
{
    if (!log_empty(&g->log)) {
        int64_t last_term = g->log.r[g->log.n - 1].term;
        return term > last_term || (term == last_term && index >= g->log.last);
    }
    if ((index > g->epoch_index && term < g->epoch_term) ||
        (index == g->epoch_index && term != g->epoch_term)) {
        fx->status = RG_A_IMPOSSIBLE_LOG;
        return -1;
    }
    return index >= g->epoch_index;
}

static void reply(fx_t *fx, int64_t term, int success)
{
    fx->resp_term = term;
    fx->flags |= RG_F_REPLIED | (success ? RG_F_SUCCESS : 0);
}

/* ------------------------------------------------------------------------------------------- */
/* RaftParticipant.appendEntries as dispatched on the current role                              */

static void on_append_entries(const orc_table_t *t, group_t *g, fx_t *fx, int64_t term, int32_t leader,
                              int64_t prev_index, int64_t prev_term, uint32_t n, const int64_t *terms,
                              int64_t leader_commit)
{
    if (g->role == RG_LEADER) {                                   /* Leader.appendEntries member/Leader.java:67-86 */
        if (leader == (int32_t)t->self) { fx->status = RG_A_LEADER_SELF_AE; return; }
        if (term < g->current_term) { reply(fx, g->current_term, 0); return; }
        if (term == g->current_term) { fx->status = RG_A_SAME_TERM_LEADER; return; }
        if (switch_to(t, g, fx, RG_FOLLOWER, g->current_term, g->voted_for) < 0) return;
    } else if (g->role == RG_CANDIDATE) {                         /* Candidate.appendEntries member/Candidate.java:28-41 */
        if (term < g->current_term) { reply(fx, g->current_term, 0); return; }
        if (switch_to(t, g, fx, RG_FOLLOWER, term, g->voted_for) < 0) return;
    }
    /* Follower.appendEntries member/Follower.java:35-88 (ctx.participant() is a Follower by now) */
    if (term < g->current_term) { reply(fx, g->current_term, 0); return; }
    fx->flags |= RG_F_RESET_TIMER;                                /* :43 */
    if (term > g->current_term || g->timeout_detected) {          /* :45-47, re-dispatched on the fresh Follower */
        if (switch_to(t, g, fx, RG_FOLLOWER, term, g->voted_for) < 0) return;
    } else if (g->current_leader != RG_NO_NODE && leader != g->current_leader) {
        fx->status = RG_A_TWO_LEADERS;                            /* :48-50: thrown after the mute at :43, outside the try whose */
        fx->flags |= RG_F_TIMER_MUTED; return;                    /* finally un-mutes (:55-85): the timer stays at Long.MAX_VALUE */
    }
    g->current_leader = leader;                                   /* :54 */
    int c = log_contains(g, fx, prev_index, prev_term);           /* :57 */
    if (c < 0) return;
    if (!c) { reply(fx, g->current_term, 0); return; }
    /* purgeEntries :209-221 — entry k has index prev_index+1+k */
    int64_t e0 = wadd(prev_index, 1);
    if (n > 0 && e0 <= g->epoch_index) {
        uint64_t skip = (uint64_t)(g->epoch_index - e0) + 1;
        if (skip >= n) { n = 0; } else { n -= (uint32_t)skip; terms += skip; e0 = wadd(e0, (int64_t)skip); }
    }
    if (n > 0) {                                                  /* :68-74 */
        int64_t conflict = log_conflict(&g->log, e0, n, terms);
        if (conflict) {
            log_truncate(&g->log, conflict);
            fx->flags |= RG_F_LOG_TRUNC;
        }
        int64_t from = log_empty(&g->log) ? e0 : max64(e0, wadd(g->log.last, 1));
        int st = log_append(g, e0, n, terms);
        if (conflict || (st == RG_OK && from <= wadd(e0, (int64_t)n - 1))) {
            fx->log_from = conflict ? conflict : from;
            if (st == RG_OK) fx->flags |= RG_F_LOG_APPEND;
        }
        if (st) { fx->status = (uint32_t)st; return; }
    }
    if (leader_commit > g->epoch_index && !log_empty(&g->log)) {  /* :76-82 */
        int st = mark_committed(g, fx, min64(leader_commit, g->log.last));
        if (st) { fx->status = (uint32_t)st; return; }
    }
    reply(fx, term, 1);                                           /* :87 (Q1: the request term) */
}

/* RaftParticipant.requestVote / preVote as dispatched on the current role */
static void on_vote_request(const orc_table_t *t, group_t *g, fx_t *fx, int pre, int64_t term, int32_t cand,
                            int64_t last_index, int64_t last_term)
{
    if (g->role == RG_LEADER) {
        if (pre) { reply(fx, g->current_term, 0); return; }       /* Leader.preVote member/Leader.java:89-91 */
        /* Leader.requestVote :94-111 */
        if (term < g->current_term) { reply(fx, g->current_term, 0); return; }
        if (term == g->current_term) {
            if (g->voted_for == (int32_t)t->self) { reply(fx, g->current_term, 0); return; }
            fx->status = RG_A_LEADER_NOT_SELF_VOTE; return;
        }
        if (switch_to(t, g, fx, RG_FOLLOWER, g->current_term, cand) < 0) return;
        pre = 0;                                                  /* re-dispatched as requestVote */
    } else if (g->role == RG_CANDIDATE) {                         /* Candidate.preVote == requestVote member/Candidate.java:44-72 */
        if (cand == (int32_t)t->self) { fx->status = RG_A_CAND_SELF_RV; return; }
        if (term < g->current_term) { reply(fx, g->current_term, 0); return; }
        if (term == g->current_term) {
            if (cand != g->voted_for) { reply(fx, g->current_term, 0); return; }
            if (g->voted_for != (int32_t)t->self) { fx->status = RG_A_CAND_NOT_SELF_VOTE; return; }
        }
        if (switch_to(t, g, fx, RG_FOLLOWER, term, cand) < 0) return;     /* Q5: no freshness check */
        pre = 0;                                                  /* ctx.participant().requestVote(...) */
    }
    if (pre) {                                                    /* Follower.preVote member/Follower.java:91-105 */
        if (term <= g->current_term || !g->timeout_detected) { reply(fx, g->current_term, 0); return; }
        fx->flags |= RG_F_RESET_TIMER;
        int ok = log_up_to_date(g, fx, last_index, last_term);
        if (ok < 0) return;
        reply(fx, g->current_term, ok);
        return;
    }
    /* Follower.requestVote member/Follower.java:108-127 */
    if (term < g->current_term) { reply(fx, g->current_term, 0); return; }
    if (term == g->current_term) { reply(fx, g->current_term, cand == g->voted_for); return; }
    fx->flags |= RG_F_RESET_TIMER;                                /* :118 resetTimer(this, true) — nothing un-mutes on this path: */
    int ok = log_up_to_date(g, fx, last_index, last_term);        /* the fresh Follower of :125 gets a new ticket, but a throw */
    if (ok < 0) { fx->flags |= RG_F_TIMER_MUTED; return; }        /* in logUpToDate leaves this one muted */
    if (switch_to(t, g, fx, RG_FOLLOWER, term, ok ? cand : RG_NO_NODE) < 0) return;
    reply(fx, g->current_term, cand == g->voted_for);             /* re-dispatched: term == currentTerm now */
}

/* Leader.replicateLog response callbacks: member/Leader.java:174-188 (snapshot) and :218-237 */
static void on_replicate_ack(const orc_table_t *t, group_t *g, fx_t *fx, int snapshot, uint32_t slot,
                             int64_t resp_term, int success, int64_t epoch_at_send, int64_t last_sent,
                             uint32_t sent_epoch)
{
    if (sent_epoch != g->role_epoch) { fx->status = RG_DROPPED_STALE_ROLE; return; }
    if (g->role != RG_LEADER || !g->repl_prepared) { fx->status = RG_BAD_EVENT; return; }
    uint32_t j = slot < t->self ? slot : slot - 1;
    if (resp_term > g->current_term) {                            /* Q7: ballot = responder */
        switch_to(t, g, fx, RG_FOLLOWER, resp_term, (int32_t)slot);
        return;
    }
    peer_t *s = &g->peers[j];
    if (fx->timed) {                                              /* statSuccess member/Leadership.java:53-57 */
        if (fx->now > s->request_success) s->request_success = fx->now;
        if (s->recent_failure != 0) s->recent_failure = 0;
    }
    if (!success) s->rejection = (int32_t)((uint32_t)s->rejection + 1u);   /* statSuccess member/Leadership.java:58-63 */
    else if (s->rejection != 0) s->rejection = 0;
    int st = update_index(s, epoch_at_send, snapshot ? epoch_at_send : last_sent, success, snapshot);
    if (st) { fx->status = (uint32_t)st; return; }
    if (!snapshot && success) try_commit(t, g, fx);
}

/* vote tallies: Candidate.startElection callback member/Candidate.java:121-134,
 *               Follower.prepareElection callback member/Follower.java:258-270 */
static void on_vote_reply(const orc_table_t *t, group_t *g, fx_t *fx, int pre, uint32_t slot,
                          int64_t resp_term, int granted, uint32_t sent_epoch)
{
    if (sent_epoch == g->role_epoch) {
        if (pre ? !(g->role == RG_FOLLOWER && g->timeout_detected) : g->role != RG_CANDIDATE) {
            fx->status = RG_BAD_EVENT; return;
        }
        int64_t T = pre ? wadd(g->current_term, 1) : g->current_term;
        if (resp_term > T) {
            switch_to(t, g, fx, RG_FOLLOWER, resp_term, (int32_t)slot);
        } else if (granted) {
            g->votes += 1;
            if (g->votes >= t->majority) {
                if (pre) {
                    switch_to(t, g, fx, RG_CANDIDATE, T, (int32_t)t->self);
                } else {
                    g->elected_epoch = g->role_epoch;             /* elected = true: head survives the fencing (Q13) */
                    g->elected_term = g->current_term;
                    switch_to(t, g, fx, RG_LEADER, T, (int32_t)t->self);
                }
            }
        }
        return;
    }
    if (!pre && g->elected_epoch != 0 && sent_epoch == g->elected_epoch) {   /* Q13: late reply on a winner's head */
        int64_t T = g->elected_term;
        if (resp_term > T) {
            g->elected_epoch = 0;                                 /* head.abortRequests() */
            switch_to(t, g, fx, RG_FOLLOWER, resp_term, (int32_t)slot);
        } else if (granted) {
            switch_to(t, g, fx, RG_LEADER, T, (int32_t)t->self); /* votes already >= majority */
        }
        return;
    }
    fx->status = RG_DROPPED_STALE_ROLE;
}

/* RaftParticipant.installSnapshot: member/Follower.java:129-152; Candidate and Leader inherit member/RaftMember.java:61-66.
 * `host_ok` is what RaftContext.installSnapshot returned (download + apply are host work, context/RaftContext.java:270-278). */
static void on_install_snapshot(const orc_table_t *t, group_t *g, fx_t *fx, int64_t term, int host_ok)
{
    if (g->role != RG_FOLLOWER) {                                 /* RaftMember.installSnapshot */
        if (term >= g->current_term) { fx->status = RG_A_INSTALL_BEFORE_AE; return; }
        reply(fx, g->current_term, 0);
        return;
    }
    fx->flags |= RG_F_RESET_TIMER;                                /* :134 resetTimer(this, true) BEFORE the term checks */
    if (term < g->current_term) { fx->flags |= RG_F_TIMER_MUTED; reply(fx, g->current_term, 0); return; }   /* :136-137, never un-muted */
    if (term > g->current_term) { fx->flags |= RG_F_TIMER_MUTED; fx->status = RG_A_INSTALL_BEFORE_AE; return; }   /* :138-139 */
    if (g->timeout_detected) {                                    /* :140-142 refresh, then the fresh Follower runs the same method */
        if (switch_to(t, g, fx, RG_FOLLOWER, g->current_term, g->voted_for) < 0) return;
    }
    reply(fx, g->current_term, host_ok);                          /* :147-152 (the finally un-mutes) */
}

/* RaftParticipant.onTimeout: member/Follower.java:156-168, Candidate.java:82-88, Leader.java:120-126.
 * Only run for the participant whose ticket fired (context/RaftRoutine.java:57,70): `ticket_epoch` 0 = whoever is current. */
static void on_timeout(const orc_table_t *t, group_t *g, fx_t *fx, uint32_t ticket_epoch)
{
    if (ticket_epoch == 0 && t->require_fence) { fx->status = RG_BAD_EVENT; return; }
    if (ticket_epoch != 0 && ticket_epoch != g->role_epoch) { fx->status = RG_DROPPED_STALE_ROLE; return; }
    if (g->role == RG_FOLLOWER) {
        if (t->pre_vote) {
            if (switch_to(t, g, fx, RG_FOLLOWER, g->current_term, g->voted_for) < 0) return;
            g->timeout_detected = 1;                              /* prepareElection :223-279 */
            g->votes = 1;
            fx->flags |= RG_EMIT_PREVOTE << RG_F_EMIT_SHIFT;
        } else {
            switch_to(t, g, fx, RG_CANDIDATE, wadd(g->current_term, 1), (int32_t)t->self);
        }
    } else if (g->role == RG_CANDIDATE) {
        switch_to(t, g, fx, RG_CANDIDATE, wadd(g->current_term, 1), (int32_t)t->self);
    } else {
        fx->flags |= RG_F_RESET_TIMER;                            /* keepAlive context/RaftRoutine.java:53-62 */
        prepare_replication(t, g);
        fx->flags |= RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT;
    }
}

/* RaftStub.process -> Leader.acceptCommand -> RaftContext.acceptCommand -> RaftLog.newEntry, then replicateLog(false)
 * command/RaftStub.java:79-91, member/Leader.java:128-140, context/RaftContext.java:223-237, storage/RocksLog.java:82-89 */
static void on_client_append(const orc_table_t *t, group_t *g, fx_t *fx, uint32_t n)
{
    if (g->role != RG_LEADER) { fx->status = RG_NOT_LEADER; return; }
    if (n == 0) return;
    if (log_empty(&g->log) && g->epoch_index > 0) { fx->status = RG_UNSUPPORTED_LOG_STATE; return; }   /* Q14 */
    for (uint32_t k = 0; k < n; k++) {
        int64_t index = log_empty(&g->log) ? 1 : wadd(g->log.last, 1);
        if (k == 0) fx->log_from = index;
        log_push(&g->log, index, g->current_term);
        prepare_replication(t, g);                                /* replicateLog(false) after every command */
    }
    fx->flags |= RG_F_LOG_APPEND | (RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT);
}

/* ------------------------------------------------------------------------------------------- */

static void step(const orc_table_t *t, group_t *g, const rg_batch_t *in, size_t row,
                 rg_reply_t *rep, rg_logfx_t *lfx, rg_persist_t *per)
{
    const uint32_t hdr = in->head[row].hdr, aux = in->head[row].aux;
    const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), flag = RG_HDR_FLAG(hdr), n = RG_HDR_N(hdr);
    const int64_t a = in->ab[row].x, b = in->ab[row].y, c = in->cd[row].x, d = in->cd[row].y;
    fx_t fx = {0, RG_OK, 0, 0, 0, 0};
    if (t->clock) { fx.now = t->clock[row / in->count]; fx.timed = 1; }

    switch (kind) {
    case RG_EV_NONE:
        break;
    case RG_EV_AE_REQ:
        if (slot >= t->cluster || n > RG_MAX_AE_ENTRIES || (n > 0 && (in->entry_terms == NULL || (uint64_t)aux + n > in->entry_count))) {
            fx.status = RG_BAD_EVENT; break;
        }
        on_append_entries(t, g, &fx, a, (int32_t)slot, b, c, n, n ? in->entry_terms + aux : NULL, d);
        break;
    case RG_EV_AE_ACK:
    case RG_EV_IS_ACK:
        if (slot >= t->cluster || slot == t->self) { fx.status = RG_BAD_EVENT; break; }
        on_replicate_ack(t, g, &fx, kind == RG_EV_IS_ACK, slot, a, (int)flag, b, c, aux);
        break;
    case RG_EV_RV_REQ:
    case RG_EV_PV_REQ:
        if (slot >= t->cluster) { fx.status = RG_BAD_EVENT; break; }
        on_vote_request(t, g, &fx, kind == RG_EV_PV_REQ, a, (int32_t)slot, b, c);
        break;
    case RG_EV_RV_REPLY:
    case RG_EV_PV_REPLY:
        if (slot >= t->cluster || slot == t->self) { fx.status = RG_BAD_EVENT; break; }
        on_vote_reply(t, g, &fx, kind == RG_EV_PV_REPLY, slot, a, (int)flag, aux);
        break;
    case RG_EV_TIMEOUT:
        on_timeout(t, g, &fx, aux);
        break;
    case RG_EV_IS_REQ:
        if (slot >= t->cluster) { fx.status = RG_BAD_EVENT; break; }
        on_install_snapshot(t, g, &fx, a, (int)flag);
        break;
    case RG_EV_CLIENT_APPEND:
        on_client_append(t, g, &fx, n);
        break;
    case RG_EV_LOG_FLUSH: {
        int st = log_flush(g, a, b);
        if (st) fx.status = (uint32_t)st;
        break;
    }
    default:
        fx.status = RG_BAD_EVENT;
        break;
    }

    rep->resp_term = (fx.flags & RG_F_REPLIED) ? fx.resp_term : 0;
    rep->flags = fx.flags | ((uint32_t)g->role << RG_F_ROLE_SHIFT) | (fx.status << RG_F_STATUS_SHIFT);
    rep->role_epoch = g->role_epoch;
    if (fx.flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) {
        lfx->commit_index = g->commit_index;
        lfx->log_from = fx.log_from;
    }
    if (fx.flags & RG_F_PERSIST) {
        per->term = g->current_term;
        per->voted_for = g->voted_for;
        per->role = g->role;
    }
}

/* ------------------------------------------------------------------------------------------- */

orc_table_t *orc_table_create(uint32_t groups, uint32_t cluster, uint32_t self_slot, int pre_vote)
{
    if (groups == 0 || cluster < RG_MIN_CLUSTER || cluster > RG_MAX_CLUSTER || self_slot >= cluster) return NULL;
    orc_table_t *t = (orc_table_t *)calloc(1, sizeof(*t));
    if (!t) return NULL;
    t->groups = groups; t->cluster = cluster; t->self = self_slot; t->followers = cluster - 1;
    t->pre_vote = pre_vote != 0;
    t->majority = (int)(cluster / 2 + 1);
    t->g = (group_t *)calloc(groups, sizeof(group_t));
    t->deadline = (int64_t *)calloc(groups, sizeof(int64_t));
    t->election_ms = 900; t->heartbeat_ms = 300;
    if (!t->g || !t->deadline) { free(t->g); free(t->deadline); free(t); return NULL; }
    for (uint32_t i = 0; i < groups; i++) {
        t->g[i].voted_for = RG_NO_NODE;
        t->g[i].current_leader = RG_NO_NODE;
        t->g[i].role_epoch = 1;
        t->g[i].votes = 1;
    }
    return t;
}

int orc_table_option(orc_table_t *t, int option, int value)
{
    if (!t || option != RG_OPT_REQUIRE_FENCED_TIMEOUTS) return -1;
    t->require_fence = value != 0;
    return 0;
}

void orc_table_destroy(orc_table_t *t)
{
    if (!t) return;
    for (uint32_t i = 0; i < t->groups; i++) free(t->g[i].log.r);
    free(t->g);
    free(t->deadline);
    free(t);
}

int orc_load_state(orc_table_t *t, uint32_t first, uint32_t count, const rg_group_state_t *s)
{
    if (!t || !s || (uint64_t)first + count > t->groups) return -1;
    const uint32_t F = t->followers;
    for (uint32_t i = 0; i < count; i++) {
        group_t *g = &t->g[first + i];
        t->deadline[first + i] = 0;
        g->current_term = s->current_term[i];
        g->voted_for = s->voted_for[i];
        g->role = s->role[i];
        g->current_leader = s->current_leader[i];
        g->timeout_detected = s->timeout_detected[i] != 0;
        g->repl_prepared = s->repl_prepared[i] != 0;
        g->role_epoch = s->role_epoch[i];
        g->votes = s->votes[i];
        g->elected_epoch = s->elected_epoch[i];
        g->elected_term = s->elected_term[i];
        g->commit_index = s->commit_index[i];
        g->epoch_index = s->epoch_index[i];
        g->epoch_term = s->epoch_term[i];
        uint32_t rc = s->run_count[i], ro = s->run_offset[i];
        g->log.n = 0;
        if (rc) {
            log_reserve(&g->log, rc);
            for (uint32_t k = 0; k < rc; k++) {
                g->log.r[k].start = s->run_start[ro + k];
                g->log.r[k].term = s->run_term[ro + k];
            }
            g->log.n = rc;
            g->log.first = s->first_index[i];
            g->log.last = s->last_index[i];
            if (g->log.r[0].start != g->log.first || g->log.last < g->log.r[rc - 1].start) return -2;
            if (g->log.first != g->epoch_index && g->log.first != g->epoch_index + 1) return -3;
        }
        for (uint32_t j = 0; j < F; j++) {
            peer_t *p = &g->peers[j];
            p->last_epoch = s->peer_last_epoch[(size_t)i * F + j];
            p->next_index = s->peer_next_index[(size_t)i * F + j];
            p->match_index = s->peer_match_index[(size_t)i * F + j];
            p->rejection = s->peer_rejection[(size_t)i * F + j];
            p->pending = s->peer_pending[(size_t)i * F + j] != 0;
            p->request_success = p->request_failure = 0; p->recent_failure = 0;   /* statistics are not part of the snapshot */
        }
    }
    return 0;
}

int orc_read_state(orc_table_t *t, uint32_t first, uint32_t count, rg_group_state_t *d)
{
    if (!t || !d || (uint64_t)first + count > t->groups) return -1;
    const uint32_t F = t->followers;
    for (uint32_t i = 0; i < count; i++) {
        const group_t *g = &t->g[first + i];
        d->current_term[i] = g->current_term;
        d->voted_for[i] = g->voted_for;
        d->role[i] = g->role;
        d->current_leader[i] = g->current_leader;
        d->timeout_detected[i] = g->timeout_detected;
        d->repl_prepared[i] = g->repl_prepared;
        d->role_epoch[i] = g->role_epoch;
        d->votes[i] = g->votes;
        d->elected_epoch[i] = g->elected_epoch;
        d->elected_term[i] = g->elected_term;
        d->commit_index[i] = g->commit_index;
        d->epoch_index[i] = g->epoch_index;
        d->epoch_term[i] = g->epoch_term;
        uint32_t rc = g->log.n > RG_TERM_RUNS ? RG_TERM_RUNS : g->log.n;
        d->run_count[i] = rc;
        d->run_offset[i] = i * RG_TERM_RUNS;
        d->first_index[i] = g->log.n ? g->log.first : 0;
        d->last_index[i] = g->log.n ? g->log.last : 0;
        for (uint32_t k = 0; k < RG_TERM_RUNS; k++) {
            int have = k < rc;
            d->run_start[(size_t)i * RG_TERM_RUNS + k] = have ? g->log.r[g->log.n - rc + k].start : 0;
            d->run_term[(size_t)i * RG_TERM_RUNS + k] = have ? g->log.r[g->log.n - rc + k].term : 0;
        }
        for (uint32_t j = 0; j < F; j++) {
            const peer_t *p = &g->peers[j];
            d->peer_last_epoch[(size_t)i * F + j] = p->last_epoch;
            d->peer_next_index[(size_t)i * F + j] = p->next_index;
            d->peer_match_index[(size_t)i * F + j] = p->match_index;
            d->peer_rejection[(size_t)i * F + j] = p->rejection;
            d->peer_pending[(size_t)i * F + j] = p->pending;
        }
    }
    return 0;
}

static int check_batch(const orc_table_t *t, const rg_batch_t *in, const rg_outcome_t *out)
{
    if (!t || !in || !out || !in->head || !in->ab || !in->cd || !out->reply || !out->logfx || !out->persist) return -1;
    if (in->rounds == 0) return -1;
    if (in->gid) {
        if (in->rounds != 1 || in->count > t->groups) return -1;
        for (uint32_t i = 0; i < in->count; i++) {
            if (in->gid[i] >= t->groups) return -1;
            if (i && in->gid[i] <= in->gid[i - 1]) return -1;
        }
    } else if (in->count != t->groups) return -1;
    return 0;
}

int orc_submit(orc_table_t *t, const rg_batch_t *in, const rg_outcome_t *out)
{
    if (check_batch(t, in, out)) return -1;
    for (uint32_t r = 0; r < in->rounds; r++)
        for (uint32_t i = 0; i < in->count; i++) {
            size_t row = (size_t)r * in->count + i;
            uint32_t gid = in->gid ? in->gid[i] : i;
            step(t, &t->g[gid], in, row, &out->reply[row], &out->logfx[row], &out->persist[row]);
        }
    return 0;
}

/* Leader.replicateLog: member/Leader.java:142-245, with RaftLog.batch: storage/RocksLog.java:131-166 */
int orc_replicate(orc_table_t *t, uint32_t count, const uint32_t *gid, const uint8_t *heartbeat, const uint16_t *in_flight,
                  rg_send_head_t *head, rg_send_t *send)
{
    if (!t || !head || !send) return -1;
    if (gid ? count > t->groups : count != t->groups) return -1;
    const uint32_t F = t->followers;
    for (uint32_t i = 0; i < count; i++) {
        if (gid && (gid[i] >= t->groups || (i && gid[i] <= gid[i - 1]))) return -1;
        group_t *g = &t->g[gid ? gid[i] : i];
        rg_send_head_t *h = &head[i];
        rg_send_t *out = send + i;                       /* follower-major: (j, row) at j * count + row */
        const size_t os = count;
        h->term = g->current_term; h->leader_commit = g->commit_index;
        h->epoch_index = g->epoch_index; h->epoch_term = g->epoch_term;
        h->role_epoch = g->role_epoch; h->is_leader = g->role == RG_LEADER; h->reserved = 0;
        if (g->role != RG_LEADER) {
            for (uint32_t j = 0; j < F; j++) out[j * os] = (rg_send_t){0, 0, 0, 0, RG_SEND_NONE};
            continue;
        }
        prepare_replication(t, g);                                               /* :146 */
        const int hb = heartbeat && heartbeat[i];
        const uint32_t limit = (uint32_t)(RG_IN_FLIGHT_LIMIT / (hb ? 10 : 1));      /* :162 */
        const int64_t fetch = RG_REPLICATE_LIMIT >> (hb ? 1 : 0);                  /* :194 */
        for (uint32_t j = 0; j < F; j++) {
            const peer_t *s = &g->peers[j];
            rg_send_t o = {g->epoch_index, g->epoch_term, g->epoch_index, 0, RG_SEND_APPEND};
            const uint32_t fl = in_flight ? in_flight[(size_t)j * os + i] : 0;
            if (fl > limit) { o.kind = RG_SEND_GATED; out[j * os] = o; continue; }     /* :163-166 */
            if (s->pending) { o.kind = RG_SEND_SNAPSHOT; out[j * os] = o; continue; }  /* :168-190 */
            const int64_t next = max64(wsub(s->next_index, 1), g->epoch_index);    /* :193 */
            /* entries = log.batch(next, fetch + 1) */
            int64_t idx = next, len = fetch + 1;
            if (idx == g->epoch_index) { idx = wadd(idx, 1); len -= 1; }
            int64_t got = 0, first_term = 0, t_;
            while (got < len && log_get(&g->log, wadd(idx, got), &t_)) { if (got == 0) first_term = t_; got++; }
            if (got > 0) {
                if (idx == next) {                                                /* prevEntry.index() == nextIndex :197-200 */
                    o.prev_index = next; o.prev_term = first_term;
                    o.count = (uint32_t)(got - 1);
                } else {
                    o.count = (uint32_t)got;                                      /* prevEntry.index() == epoch.index()+1 :201-203 */
                }
                o.last_index = o.count == 0 ? o.prev_index : wadd(o.prev_index, o.count);   /* :204-208 */
            }
            out[j * os] = o;
        }
    }
    return 0;
}

/* ---- N4 timers: RaftRoutine.resetTimer / electionTimeout / keepAlive  context/RaftRoutine.java:53-130 ---------- */

static uint64_t timer_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* RaftConfig.electionTimeout: uniform in [E, 2E] (support/RaftConfig.java:187-190); the draw itself is ours */
static int64_t election_timeout(uint64_t seed, uint32_t gid, uint32_t role_epoch, int64_t now, int64_t E)
{
    uint64_t h = timer_mix(seed ^ timer_mix((uint64_t)gid * 0xD1342543DE82EF95ull ^ ((uint64_t)role_epoch << 32) ^ (uint64_t)now));
    return E + (int64_t)(h % (uint64_t)(E + 1));
}

/* the deadline resetTimer leaves behind: a handler mutes (deadline MAX) and un-mutes, so the un-muted reset sees
 * moment == MAX and lands on now + timeout (:105-107); a Leader is re-scheduled heartbeatInterval ahead, at once
 * when the ticket is new (:117-118); a ticket that already fired (moment < 0) is not replaced (:96-98) */
static int64_t rearm(const orc_table_t *t, int64_t d, uint32_t gid, int role, int fresh, int muted, uint32_t role_epoch, int64_t now)
{
    if (fresh) d = 0;                                            /* convertTo: ticketHolder.set(null) :198 */
    if (role == RG_LEADER) return d == 0 ? now : wadd(now, t->heartbeat_ms);
    if (d < 0) return d;
    if (muted) return INT64_MAX;                                 /* resetTimer(.., true) with no un-muting call after it :101-107 */
    return wadd(now, election_timeout(t->timer_seed, gid, role_epoch, now, t->election_ms));
}

int orc_timers_configure(orc_table_t *t, int64_t election_ms, int64_t heartbeat_ms, uint64_t seed)
{
    if (!t || election_ms <= 0 || heartbeat_ms <= 0) return -1;
    t->election_ms = election_ms; t->heartbeat_ms = heartbeat_ms; t->timer_seed = seed;
    return 0;
}

int orc_timers_update(orc_table_t *t, uint32_t rounds, uint32_t count, const uint32_t *gid, const rg_reply_t *reply, const int64_t *now)
{
    if (!t || !reply || !now || rounds == 0) return -1;
    if (gid ? (rounds != 1 || count > t->groups) : count != t->groups) return -1;
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t g = gid ? gid[i] : i;
        if (g >= t->groups) return -1;
        int64_t d = t->deadline[g];
        for (uint32_t r = 0; r < rounds; r++) {
            const rg_reply_t *rep = &reply[(size_t)r * count + i];
            if (rep->flags & RG_F_RESET_TIMER)
                d = rearm(t, d, g, (int)RG_F_ROLE(rep->flags), (rep->flags & RG_F_ROLE_CHANGED) != 0,
                          (rep->flags & RG_F_TIMER_MUTED) != 0, rep->role_epoch, now[r]);
        }
        t->deadline[g] = d;
    }
    return 0;
}

int orc_timers_arm(orc_table_t *t, int64_t now)
{
    if (!t) return -1;
    for (uint32_t g = 0; g < t->groups; g++)
        if (t->deadline[g] == 0) t->deadline[g] = rearm(t, 0, g, t->g[g].role, 1, 0, t->g[g].role_epoch, now);
    return 0;
}

/* electionTimeout / keepAlive firing: deadline reached -> CAS to TimerTicket.TIMEOUT, onTimeout gets queued (:53-77) */
int orc_timers_expired_epochs(orc_table_t *t, int64_t now, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, uint32_t *out_count)
{
    if (!t || !out_count) return -1;
    uint32_t n = 0;
    for (uint32_t g = 0; g < t->groups; g++) {
        if (t->deadline[g] > 0 && t->deadline[g] <= now) {
            if (n < capacity) { out_gid[n] = g; if (out_epoch) out_epoch[n] = t->g[g].role_epoch; t->deadline[g] = -1; }
            n++;
        }
    }
    *out_count = n;
    return 0;
}

int orc_timers_expired(orc_table_t *t, int64_t now, uint32_t *out_gid, uint32_t capacity, uint32_t *out_count)
{
    return orc_timers_expired_epochs(t, now, out_gid, NULL, capacity, out_count);
}

int orc_timers_read(orc_table_t *t, uint32_t first, uint32_t count, int64_t *deadline)
{
    if (!t || !deadline || (uint64_t)first + count > t->groups) return -1;
    memcpy(deadline, t->deadline + first, (size_t)count * sizeof(int64_t));
    return 0;
}

/* ---- N4b: follower health and Leader.isReady ---------------------------------------------------- */

/* the wall clock the following orc_submit calls see, one value per round (NULL: statistics are not kept) */
int orc_health_clock(orc_table_t *t, const int64_t *now_per_round)
{
    if (!t) return -1;
    t->clock = now_per_round;
    return 0;
}

/* State.statFailure(now, unreachable, reject): member/Leadership.java:65-73; callers member/Leader.java:187,235,240.
 * Rows for groups that are not prepared leaders have no State object to land on and are ignored. */
int orc_health_failure(orc_table_t *t, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, int64_t now)
{
    if (!t || (n && (!gid || !slot || !flags))) return -1;
    for (uint32_t i = 0; i < n; i++) {
        if (gid[i] >= t->groups || slot[i] >= t->cluster || slot[i] == t->self) continue;
        group_t *g = &t->g[gid[i]];
        if (g->role != RG_LEADER || !g->repl_prepared) continue;
        peer_t *s = &g->peers[slot[i] < t->self ? slot[i] : slot[i] - 1];
        if (now > s->request_failure) s->request_failure = now;
        if (flags[i] & 1u) s->recent_failure = (int32_t)((uint32_t)s->recent_failure + 1u);
        if (flags[i] & 2u) s->rejection = (int32_t)((uint32_t)s->rejection + 1u);
    }
    return 0;
}

/* State.isUnhealthy / State.isReady: member/Leadership.java:43-51 */
static int state_ready(const peer_t *s, int32_t critical_point, int64_t cool_down, int64_t now)
{
    int unhealthy = (critical_point > 0 && (uint32_t)s->recent_failure > (uint32_t)critical_point) ||
                    (cool_down > 0 && wsub(now, s->request_failure) < cool_down);
    return s->request_success != 0 && !(s->pending || unhealthy);
}

/* Leader.isReady: member/Leader.java:52-64 (false for every other role: command/RaftStub.java:80-87) */
int orc_ready(orc_table_t *t, int64_t now, int32_t critical_point, int64_t cool_down_ms, uint8_t *ready)
{
    if (!t || !ready) return -1;
    for (uint32_t i = 0; i < t->groups; i++) {
        const group_t *g = &t->g[i];
        ready[i] = 0;
        if (g->role != RG_LEADER || !g->repl_prepared) continue;
        int n = 1, half = (int)t->followers / 2;
        for (uint32_t j = 0; j < t->followers; j++)
            if (state_ready(&g->peers[j], critical_point, cool_down_ms, now) && ++n > half) { ready[i] = 1; break; }
    }
    return 0;
}

int orc_health_read(orc_table_t *t, uint32_t first, uint32_t count, int64_t *request_success, int64_t *request_failure,
                    int32_t *recent_failure)
{
    if (!t || !request_success || !request_failure || !recent_failure || (uint64_t)first + count > t->groups) return -1;
    const uint32_t F = t->followers;
    for (uint32_t i = 0; i < count; i++)
        for (uint32_t j = 0; j < F; j++) {
            const peer_t *s = &t->g[first + i].peers[j];
            request_success[(size_t)i * F + j] = s->request_success;
            request_failure[(size_t)i * F + j] = s->request_failure;
            recent_failure[(size_t)i * F + j] = s->recent_failure;
        }
    return 0;
}

int orc_log_term(const orc_table_t *t, uint32_t gid, int64_t index, int64_t *term)
{
    if (!t || gid >= t->groups) return 0;
    return log_get(&t->g[gid].log, index, term);
}

int64_t orc_log_conflict(const orc_table_t *t, uint32_t gid, int64_t e0, uint32_t n, const int64_t *terms)
{
    if (!t || gid >= t->groups) return 0;
    return log_conflict(&t->g[gid].log, e0, n, terms);
}

/* ---- threaded CPU baseline ------------------------------------------------------------------- */

typedef struct {
    orc_table_t *t; const rg_batch_t *in; const rg_outcome_t *out;
    int tid, threads; pthread_barrier_t *bar;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    pthread_barrier_wait(j->bar);
    /* contexts are dealt round-robin to the loops (EventLoopGroup.next), here in chunks of 64 so that two threads
     * never write the same cache line of the packed outcome arrays (an artefact of the batch format, not of the path) */
    const uint32_t chunk = 64, stride = chunk * (uint32_t)j->threads;
    for (uint32_t r = 0; r < j->in->rounds; r++)
        for (uint32_t base = (uint32_t)j->tid * chunk; base < j->in->count; base += stride)
            for (uint32_t i = base; i < base + chunk && i < j->in->count; i++) {
                size_t row = (size_t)r * j->in->count + i;
                step(j->t, &j->t->g[i], j->in, row, &j->out->reply[row], &j->out->logfx[row], &j->out->persist[row]);
            }
    pthread_barrier_wait(j->bar);
    return NULL;
}

double orc_submit_threads(orc_table_t *t, const rg_batch_t *in, const rg_outcome_t *out, int threads)
{
    if (check_batch(t, in, out) || in->gid || threads < 1 || threads > 256) return -1.0;
    pthread_t th[256]; job_t jobs[256];
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1);
    for (int i = 0; i < threads; i++) {
        jobs[i] = (job_t){t, in, out, i, threads, &bar};
        pthread_create(&th[i], NULL, worker, &jobs[i]);
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    pthread_barrier_destroy(&bar);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
