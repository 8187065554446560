// rg_device.hpp — device-side decision logic of the batched multi-Raft engine (gfx950 / CDNA4).
//
// One LANE owns one raft group for the whole launch: the group's scalar state lives in VGPRs, its
// per-follower replication state (Leadership.State) is staged in LDS as [follower][lane] columns so
// the responder slot of an ack — a runtime value — indexes LDS instead of forcing the register file
// through scratch.  Integer work only; no MFMA.
//
// What is decided here, and the reference code it stands in for (paths relative to
// /root/reference/src/main/java/io/lubricant/consensus/raft/):
//   switch_to            RaftRoutine.trySwitch/switchTo/convertTo  context/RaftRoutine.java:140-216
//                        Membership.isBetter                       context/member/Membership.java:74-108
//   on_append_entries    Follower/Candidate/Leader.appendEntries   member/Follower.java:35-88,177-221
//                                                                  member/Candidate.java:28-41  member/Leader.java:67-86
//   on_vote_request      requestVote/preVote of the three roles    member/Follower.java:91-127,193-207
//                                                                  member/Candidate.java:44-72  member/Leader.java:89-111
//   on_replicate_ack     Leader.replicateLog callbacks             member/Leader.java:174-188,218-237
//                        State.updateIndex/majorIndices, tryCommit member/Leadership.java:53-63,75-130  member/Leader.java:247-280
//   on_vote_reply        election / pre-election tallies           member/Candidate.java:121-134  member/Follower.java:258-270
//   on_timeout           onTimeout of the three roles              member/Follower.java:156-168  member/Candidate.java:82-88  member/Leader.java:120-126
//   log cache            RaftLog.get/last/conflict/truncate/append/flush/newEntry/markCommitted
//                                                                  command/storage/RocksLog.java:82-128,169-242
// The log itself stays with the host (RocksDB); the device keeps the newest RG_TERM_RUNS maximal
// equal-term runs of it, which answers RaftLog.get(i).term() exactly for every index at or above the
// oldest cached run.  A lookup below that is a cache miss: the row is left unapplied
// (RG_NEED_HOST) unless the host attached the answer as a hint.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/raftgpu.h"

namespace rg {

constexpr int K = RG_TERM_RUNS;
constexpr int BLOCK = 64;          // one wavefront per workgroup: lanes never share LDS columns, no barriers

struct alignas(16) I64x2 { int64_t x, y; };
struct alignas(16) Ident { int32_t voted_for, leader; uint32_t role_epoch, meta; };
struct alignas(16) Elect { int64_t elected_term; uint32_t elected_epoch; int32_t votes; };
struct alignas(16) Match { int64_t match_index; int32_t rejection; int32_t pad; };

// meta word: role[1:0] timeoutDetected[2] replPrepared[3] runCount[6:4] pendingInstallation bit per follower [14:8]
constexpr uint32_t META_ROLE = 3u, META_TD = 1u << 2, META_PREP = 1u << 3;
constexpr int META_RC_SHIFT = 4, META_PEND_SHIFT = 8;

// HBM layout of a table: structure of 16-byte structs, one column per struct kind, so every lane
// moves 16 B per load/store and a wavefront touches 1 KiB of one column at a time.
struct DevTable {
    I64x2 *term_commit;   // [G] {currentTerm, commitIndex}
    I64x2 *epoch;         // [G] {epoch.index, epoch.term}
    I64x2 *window;        // [G] {firstIndex, lastIndex}
    Ident *ident;         // [G]
    Elect *elect;         // [G]
    I64x2 *runs;          // [K][G] {start, term}
    I64x2 *peer_en;       // [F][G] {lastEpoch, nextIndex}
    Match *peer_m;        // [F][G]
    uint32_t groups;
};

struct StepParams {
    DevTable t;
    uint32_t rounds, count;
    const uint32_t *gid;            // sparse only
    const rg_ev_head_t *head;
    const I64x2 *ab, *cd, *hint;
    const int64_t *entry_terms;
    uint64_t entry_count;
    rg_reply_t *reply;
    I64x2 *logfx;
    rg_persist_t *persist;
    unsigned long long *counters;   // [RG_NUM_COUNTERS]
    int32_t self, cluster, majority, pre_vote;
    int32_t fast_paths;             // 0: general handlers only
};

struct ReplicateParams {             // N1: Leader.replicateLog for many groups (rg_kernels.hip: replicate_kernel)
    DevTable t;
    uint32_t count;
    const uint32_t *gid;
    const uint8_t *heartbeat;
    const uint16_t *in_flight;
    rg_send_head_t *head;
    rg_send_t *send;
};

struct TimerParams {                 // N4: RaftRoutine.resetTimer / electionTimeout for many groups (rg_kernels.hip)
    int64_t *deadline;               // [G] 0 = no ticket, -1 = fired (TimerTicket.TIMEOUT), >0 = armed
    const Ident *ident;              // role / role epoch of the table (arm only)
    uint32_t groups, rounds, count;
    const uint32_t *gid;
    const rg_reply_t *reply;
    int64_t now[64];                 // per-round timestamps of one update launch (<= 64 rounds per launch)
    int64_t election_ms, heartbeat_ms;
    uint64_t seed;
};

struct HealthParams {                // N4b: Leadership.State health fields + Leader.isReady (rg_kernels.hip)
    int64_t *ok, *fail;              // [F][G] requestSuccess / requestFailure
    int32_t *recent;                 // [F][G] recentFailure
    DevTable t;
    uint32_t rounds, count, followers, self;
    const uint32_t *gid;
    const rg_ev_head_t *head;
    const rg_reply_t *reply;
    int64_t now[64];
};

__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__device__ __forceinline__ int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
__device__ __forceinline__ int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }

// State.majorIndices (member/Leadership.java:116-130): sorted[0] ("full") and sorted[F/2] ("major") of the F follower
// matchIndex values. One 64-bit compare per compare-exchange; for F = 4 (five nodes) only the two order statistics
// that are used get completed: pairs, then min of the minima / max of the maxima, then the larger of the middle two.
__device__ __forceinline__ void cmp_exchange(int64_t &a, int64_t &b)
{
    const bool sw = a > b;
    const int64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}
template <int F>
__device__ __forceinline__ void major_indices(int64_t (&m)[F], int64_t &full, int64_t &major)
{
    if constexpr (F == 4) {
        cmp_exchange(m[0], m[1]); cmp_exchange(m[2], m[3]);
        cmp_exchange(m[0], m[2]); cmp_exchange(m[1], m[3]);
        full = m[0]; major = m[1] > m[2] ? m[1] : m[2];
    } else {
#pragma unroll
        for (int a = 1; a < F; a++) {                                   // insertion network, F <= 6
#pragma unroll
            for (int b = a; b > 0; b--) cmp_exchange(m[b - 1], m[b]);
        }
        full = m[0]; major = m[F / 2];
    }
}

template <int F, class V>
__device__ __forceinline__ void major_indices_v(V (&m)[F], V &full, V &major)
{
    if constexpr (sizeof(V) == 8) {
        major_indices<F>(reinterpret_cast<int64_t (&)[F]>(m), reinterpret_cast<int64_t &>(full), reinterpret_cast<int64_t &>(major));
    } else {                                                            // 32-bit: the ISA has integer min / max
        auto cx = [](V &x, V &y) { const V lo = x < y ? x : y, hi = x < y ? y : x; x = lo; y = hi; };
        if constexpr (F == 4) {
            cx(m[0], m[1]); cx(m[2], m[3]); cx(m[0], m[2]); cx(m[1], m[3]);
            full = m[0]; major = m[1] > m[2] ? m[1] : m[2];
        } else {
#pragma unroll
            for (int x = 1; x < F; x++) {
#pragma unroll
                for (int y = x; y > 0; y--) cx(m[y - 1], m[y]);
            }
            full = m[0]; major = m[F / 2];
        }
    }
}

// Math.round(Math.log(Math.E + r)) as integer thresholds (member/Leadership.java:105).
__device__ __forceinline__ int64_t rejection_step(int32_t r)
{
    if (r < 0) return r == -1 ? 1 : 0;
    int64_t s = 1;
    s += r >= 2; s += r >= 10; s += r >= 31; s += r >= 88; s += r >= 242; s += r >= 663; s += r >= 1806;
    s += r >= 4913; s += r >= 13358; s += r >= 36313; s += r >= 98714; s += r >= 268335; s += r >= 729414;
    s += r >= 1982757; s += r >= 5389696; s += r >= 14650717; s += r >= 39824782; s += r >= 108254986;
    s += r >= 294267564; s += r >= 799902175;
    return s;
}

// The 32-bit tier. Raft terms and log indices are Java longs, but a group whose values all fit in 30 bits — every real
// deployment for years — can be decided with 32-bit compares, selects and adds: half the VALU instructions of the 64-bit
// forms (a 64-bit select is two v_cndmask, a 64-bit min/max a compare plus two, an add two). try_fast therefore exists in
// two instantiations; the narrow one runs when EVERY lane of the wavefront has a narrow event and a narrow group (one
// ballot), writes only the low words back (the high words are zero and stay zero: inputs below 2^30 plus increments below
// 2^21 cannot reach 2^31), and is bit-for-bit the wide one on that domain — the differential tests run both.
constexpr uint64_t NARROW_LIMIT = 1ull << 30;
__device__ __forceinline__ bool is_narrow(int64_t v) { return (uint64_t)v < NARROW_LIMIT; }
__device__ __forceinline__ int64_t with_lo(int64_t old, uint32_t lo) { return (int64_t)(((uint64_t)old & 0xFFFFFFFF00000000ull) | lo); }

// effects of one row
struct Fx {
    uint32_t flags;
    uint32_t status;
    int64_t resp_term;
    int64_t log_from;
};

// Two facts about an AppendEntries request that do not depend on the group's state — "the carried entry terms are
// readable" and "the first min(n, 4) entry terms are all equal" — are worked out where the event is loaded. The
// two-wavefront kernel hands them to the deciding wavefront in the two unused bits of its LDS copy of the header.
constexpr uint32_t HDR_SAME = 1u << 10, HDR_ENTRIES_OK = 1u << 11;
__device__ __forceinline__ bool entries_readable(const StepParams &p, uint32_t hdr, uint32_t aux)
{
    const uint32_t n = RG_HDR_N(hdr);
    return (n == 0) | ((n <= 4u) & (p.entry_terms != nullptr) & ((uint64_t)aux + n <= p.entry_count));
}
__device__ __forceinline__ bool entries_same_term(uint32_t hdr, int64_t e0, int64_t e1, int64_t e2, int64_t e3)
{
    const uint32_t n = RG_HDR_N(hdr);
    return ((n < 2u) | (e1 == e0)) & ((n < 3u) | (e2 == e0)) & ((n < 4u) | (e3 == e0));
}

template <int F>
struct Peers {                       // LDS columns of this lane
    int64_t *last_epoch, *next_index, *match_index;
    int32_t *rejection;
};

// Register image of one group.  The K=4 cached term runs are SCALAR members on purpose: with arrays the
// optimiser folds the select chains below back into dynamically indexed accesses, which pins the whole
// struct in scratch memory (measured: 168 B/lane of scratch, every field access a scratch_load).
struct Group {
    int64_t term, commit, epoch_index, epoch_term, first, last, elected_term;
    int64_t s0, s1, s2, s3;          // run starts, ascending; valid [0, rc)
    int64_t t0, t1, t2, t3;          // run terms
    int32_t voted_for, leader, votes, role, rc;
    uint32_t role_epoch, elected_epoch, pending;
    bool td, prepared, log_dirty, peers_dirty;
    bool narrow;                     // every 64-bit value of the group (and of its LDS follower columns) is in [0, NARROW_LIMIT): see try_fast
    static_assert(K == 4, "run cache is hand-unrolled for 4 runs");

    // NOTE on style: every method first copies the fields it needs into locals, computes with value
    // selects, and stores back unconditionally.  Member functions are optimised BEFORE they are inlined
    // into the kernel, while `this` is still an opaque pointer; a load inside a conditional arm there is
    // folded into "load of a selected address", which later defeats scalar replacement of the struct.
    __device__ __forceinline__ bool has_log() const { return rc > 0; }
    __device__ __forceinline__ bool present(int64_t i) const
    {
        const int n = rc; const int64_t f = first, l = last;
        return n > 0 && i >= f && i <= l;
    }
    __device__ __forceinline__ bool cached(int64_t i) const { return i >= s0; }
    __device__ __forceinline__ int64_t last_term() const
    {
        const int n = rc; const int64_t a0 = t0, a1 = t1, a2 = t2, a3 = t3;
        int64_t t = a0;
        t = n > 1 ? a1 : t;
        t = n > 2 ? a2 : t;
        t = n > 3 ? a3 : t;
        return t;
    }
    // term of a PRESENT and CACHED index
    __device__ __forceinline__ int64_t term_at(int64_t i) const
    {
        const int n = rc; const int64_t a0 = t0, a1 = t1, a2 = t2, a3 = t3, b1 = s1, b2 = s2, b3 = s3;
        int64_t t = a0;
        t = (n > 1 && b1 <= i) ? a1 : t;
        t = (n > 2 && b2 <= i) ? a2 : t;
        t = (n > 3 && b3 <= i) ? a3 : t;
        return t;
    }
    // db.put(last+1, t) — or the first key of an empty log
    __device__ __forceinline__ void push(int64_t index, int64_t t)
    {
        int n = rc;
        int64_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, y1 = t1, y2 = t2, y3 = t3;
        const int64_t f = first;
        int64_t lt = y0;
        lt = n > 1 ? y1 : lt;
        lt = n > 2 ? y2 : lt;
        lt = n > 3 ? y3 : lt;
        const bool empty = n == 0;
        const bool newrun = empty || lt != t;
        const bool shift = newrun && n == K;     // cache full: forget the oldest run (lookups into it become misses)
        x0 = shift ? x1 : x0; y0 = shift ? y1 : y0;
        x1 = shift ? x2 : x1; y1 = shift ? y2 : y1;
        x2 = shift ? x3 : x2; y2 = shift ? y3 : y2;
        n = shift ? K - 1 : n;
        const bool w0 = newrun && n == 0, w1 = newrun && n == 1, w2 = newrun && n == 2, w3 = newrun && n == 3;
        x0 = w0 ? index : x0; y0 = w0 ? t : y0;
        x1 = w1 ? index : x1; y1 = w1 ? t : y1;
        x2 = w2 ? index : x2; y2 = w2 ? t : y2;
        x3 = w3 ? index : x3; y3 = w3 ? t : y3;
        n += newrun ? 1 : 0;
        s0 = x0; s1 = x1; s2 = x2; s3 = x3; t0 = y0; t1 = y1; t2 = y2; t3 = y3;
        rc = n;
        first = empty ? index : f;
        last = index;
        log_dirty = true;
    }
    // RaftLog.truncate(index): storage/RocksLog.java:219-225. `keep_term` = term of index-1, used only when
    // the cut lands below the cached runs (hint-resolved conflict) and the cache must be re-seeded.
    __device__ __forceinline__ void truncate(int64_t index, int64_t keep_term)
    {
        const int n0 = rc;
        const int64_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, f = first, l = last;
        const bool dirty0 = log_dirty;
        const bool act = n0 != 0 && l >= index;
        const bool wipe = act && index <= f;
        int n = (x0 < index ? 1 : 0) + ((n0 > 1 && x1 < index) ? 1 : 0) + ((n0 > 2 && x2 < index) ? 1 : 0) +
                ((n0 > 3 && x3 < index) ? 1 : 0);
        const bool reseed = act && !wipe && n == 0;
        n = reseed ? 1 : n;
        s0 = reseed ? index - 1 : x0;
        t0 = reseed ? keep_term : y0;
        rc = act ? (wipe ? 0 : n) : n0;
        last = (act && !wipe) ? index - 1 : l;
        log_dirty = dirty0 || act;
    }
    // RaftLog.flush(index, term): storage/RocksLog.java:228-242
    __device__ __forceinline__ uint32_t flush(int64_t index, int64_t term_)
    {
        int n = rc;
        int64_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, y1 = t1, y2 = t2, y3 = t3;
        const int64_t f = first, l = last, ei = epoch_index;
        const bool dirty0 = log_dirty;
        if (index < ei) return RG_FLUSH_OUT_OF_BOUNDS;
        const bool have = n != 0;
        const bool wipe = have && index > l;
        const bool trim = have && !wipe && index > f;
        // runs that end before `index` (their successor starts at or below it) are gone
        const int drop = trim ? (((n > 1 && x1 <= index) ? 1 : 0) + ((n > 2 && x2 <= index) ? 1 : 0) +
                                 ((n > 3 && x3 <= index) ? 1 : 0)) : 0;
#pragma unroll
        for (int s = 0; s < K - 1; s++) {
            const bool sh = s < drop;
            x0 = sh ? x1 : x0; y0 = sh ? y1 : y0;
            x1 = sh ? x2 : x1; y1 = sh ? y2 : y1;
            x2 = sh ? x3 : x2; y2 = sh ? y3 : y2;
        }
        n = wipe ? 0 : n - drop;
        x0 = (trim && x0 < index) ? index : x0;
        s0 = x0; s1 = x1; s2 = x2; s3 = x3; t0 = y0; t1 = y1; t2 = y2; t3 = y3;
        rc = n;
        first = trim ? index : f;
        log_dirty = dirty0 || wipe || trim;
        epoch_index = index;
        epoch_term = term_;
        return RG_OK;
    }
};

template <int F>
struct Stepper {
    const StepParams &p;
    Group &g;
    Peers<F> pe;
    Fx fx;
    bool narrow_tier = true;             // false: this kernel never takes the 32-bit tier (a compile-time constant after inlining)

    __device__ __forceinline__ Stepper(const StepParams &p_, Group &g_, const Peers<F> &pe_) : p(p_), g(g_), pe(pe_) {}

    __device__ __forceinline__ void reply(int64_t term, bool success)
    {
        fx.resp_term = term;
        fx.flags |= RG_F_REPLIED | (success ? RG_F_SUCCESS : 0u);
    }

    // returns 1 converted, 0 not better, -1 assertion
    __device__ __forceinline__ int switch_to(int role, int64_t term, int32_t ballot)
    {
        bool better;
        if (term != g.term) {
            better = term > g.term;
        } else if (role != g.role) {
            if (role == RG_LEADER) {
                if (g.role != RG_CANDIDATE) { fx.status = RG_A_LEADER_UNCHANGED; return -1; }
                better = true;
            } else {
                better = role == RG_FOLLOWER;
            }
        } else if (role == RG_LEADER) {
            better = false;
        } else if (role == RG_FOLLOWER) {
            better = true;
        } else {
            if (ballot != g.voted_for) { fx.status = RG_A_CAND_BALLOT; return -1; }
            better = false;
        }
        if (!better) return 0;
        g.role = role; g.term = term; g.voted_for = ballot;
        g.role_epoch += 1u;
        g.td = false; g.leader = RG_NO_NODE; g.votes = 1; g.prepared = false;
        fx.flags |= RG_F_PERSIST | RG_F_ROLE_CHANGED | RG_F_RESET_TIMER;
        fx.flags &= ~RG_F_EMIT_MASK;
        if (role == RG_CANDIDATE) fx.flags |= RG_EMIT_REQVOTE << RG_F_EMIT_SHIFT;
        return 1;
    }

    __device__ __forceinline__ void prepare_replication()
    {
        if (g.prepared) return;
        const int64_t l_ = g.last, e_ = g.epoch_index;
        const int64_t next = wadd(g.rc > 0 ? l_ : e_, 1);
#pragma unroll
        for (int j = 0; j < F; j++) {
            pe.last_epoch[j * BLOCK] = g.epoch_index;
            pe.next_index[j * BLOCK] = next;
            pe.match_index[j * BLOCK] = 0;
            pe.rejection[j * BLOCK] = 0;
        }
        g.pending = 0;
        g.prepared = true;
        g.peers_dirty = true;
    }

    __device__ __forceinline__ uint32_t mark_committed(int64_t index)
    {
        if (index < g.commit) return RG_A_COMMIT_ROLLBACK;
        if (index > g.commit) { g.commit = index; fx.flags |= RG_F_COMMIT; }
        return RG_OK;
    }

    // ---- appendEntries --------------------------------------------------------------------------
    // entry term k of the request: the first PREFETCHED_ENTRIES arrive in registers with the event
    // (loaded one round ahead), the rest is read from the entry stream on demand
    struct Pre { int64_t e0, e1, e2, e3; };          // scalars, not an array: see the note in Group
    __device__ __forceinline__ int64_t entry_term(const int64_t *terms, const Pre pre, uint64_t k) const
    {
        int64_t t = pre.e0;
        t = k == 1 ? pre.e1 : t;
        t = k == 2 ? pre.e2 : t;
        t = k == 3 ? pre.e3 : t;
        if (k >= 4) t = terms[k];
        return t;
    }

    __device__ __forceinline__ void on_append_entries(int64_t term, int32_t leader, int64_t prev_index,
                                                      int64_t prev_term, uint32_t n, const int64_t *terms,
                                                      const Pre pre,
                                                      int64_t leader_commit, bool hinted, int64_t hint_prev_term,
                                                      int64_t hint_conflict)
    {
        // --- cache-miss pre-check, BEFORE any mutation, so a NEED_HOST row is a no-op -------------
        bool use_hint = false;
        if (term >= g.term && g.has_log()) {
            int64_t need = INT64_MAX;
            if (prev_index > g.epoch_index && prev_index <= g.last) need = prev_index;
            if (n > 0) {
                const int64_t scan = max64(wadd(prev_index, 1), wadd(g.epoch_index, 1));
                const int64_t e_last = wadd(prev_index, (int64_t)n);
                if (scan <= g.last && scan <= e_last) need = min64(need, scan);
            }
            if (need != INT64_MAX && !g.cached(need)) {
                if (!hinted) { fx.status = RG_NEED_HOST; fx.log_from = need; return; }
                use_hint = true;
            }
        }

        if (g.role == RG_LEADER) {
            if (leader == p.self) { fx.status = RG_A_LEADER_SELF_AE; return; }
            if (term < g.term) { reply(g.term, false); return; }
            if (term == g.term) { fx.status = RG_A_SAME_TERM_LEADER; return; }
            if (switch_to(RG_FOLLOWER, g.term, g.voted_for) < 0) return;
        } else if (g.role == RG_CANDIDATE) {
            if (term < g.term) { reply(g.term, false); return; }
            if (switch_to(RG_FOLLOWER, term, g.voted_for) < 0) return;
        }
        if (term < g.term) { reply(g.term, false); return; }
        fx.flags |= RG_F_RESET_TIMER;
        if (term > g.term || g.td) {
            if (switch_to(RG_FOLLOWER, term, g.voted_for) < 0) return;
        } else if (g.leader != RG_NO_NODE && leader != g.leader) {
            fx.status = RG_A_TWO_LEADERS;                 // thrown after the mute (:43), outside the try whose finally un-mutes
            fx.flags |= RG_F_TIMER_MUTED; return;
        }
        g.leader = leader;

        // logContains
        bool contains;
        if (prev_index == 0 && prev_term == 0) {
            contains = true;
        } else if (prev_index == 0 || prev_term == 0) {
            fx.status = RG_A_PREV_ZERO_MISMATCH; return;
        } else if (prev_index <= g.epoch_index) {
            if (prev_index == g.epoch_index && prev_term != g.epoch_term) { fx.status = RG_A_EPOCH_TERM_MISMATCH; return; }
            contains = true;
        } else if (!g.present(prev_index)) {
            contains = false;
        } else {
            const int64_t t = g.cached(prev_index) ? g.term_at(prev_index) : hint_prev_term;
            contains = t == prev_term;
        }
        if (!contains) { reply(g.term, false); return; }

        // purgeEntries: entry k has index prev_index+1+k
        int64_t e0 = wadd(prev_index, 1);
        const int64_t e_first = e0;                   // index of entry 0 of the request, before the purge
        bool purged = false;
        if (n > 0 && e0 <= g.epoch_index) {
            const uint64_t skip = (uint64_t)(g.epoch_index - e0) + 1u;
            purged = true;
            if (skip >= n) { n = 0; } else { n -= (uint32_t)skip; e0 = wadd(e0, (int64_t)skip); }
        }
        if (n > 0) {
            const int64_t e_last = wadd(e0, (int64_t)n - 1);
            // RaftLog.conflict: walk the overlap with the stored keys
            int64_t conflict = 0;
            if (use_hint) {
                conflict = hint_conflict;
            } else if (g.has_log() && e0 <= g.last) {
                const int64_t stop = min64(e_last, g.last);
                for (int64_t idx = e0; idx <= stop; idx++) {
                    if (g.term_at(idx) != entry_term(terms, pre, (uint64_t)(idx - e_first))) { conflict = idx; break; }
                }
            }
            if (conflict) {
                // term of the key just below the cut; only needed when the cut empties the cached runs
                // (hint-resolved conflict below the cache) while the log itself stays non-empty
                int64_t keep = 0;
                if (g.has_log() && conflict > g.first && conflict <= g.s0)
                    keep = (conflict == e0) ? (purged ? g.epoch_term : prev_term)
                                            : entry_term(terms, pre, (uint64_t)(conflict - 1 - e_first));
                g.truncate(conflict, keep);
                fx.flags |= RG_F_LOG_TRUNC;
            }
            // RaftLog.append: only keys above the greatest stored key <= entries[0].index are written
            int64_t from;
            if (!g.has_log()) {
                if (e0 != wadd(g.epoch_index, 1)) { fx.status = RG_A_LOG_NOT_CONTINUOUS; return; }
                from = e0;
            } else {
                if (g.last < wsub(e0, 1)) { fx.status = RG_A_LOG_NOT_CONTINUOUS; return; }
                from = max64(e0, wadd(g.last, 1));
            }
            if (conflict || from <= e_last) {
                fx.log_from = conflict ? conflict : from;
                fx.flags |= RG_F_LOG_APPEND;
            }
            for (int64_t idx = from; idx <= e_last; idx++) g.push(idx, entry_term(terms, pre, (uint64_t)(idx - e_first)));
        }
        if (leader_commit > g.epoch_index && g.has_log()) {
            const uint32_t st = mark_committed(min64(leader_commit, g.last));
            if (st) { fx.status = st; return; }
        }
        reply(term, true);
    }

    // ---- requestVote / preVote ------------------------------------------------------------------
    // returns 1/0, -1 on assertion
    __device__ __forceinline__ int log_up_to_date(int64_t index, int64_t term)
    {
        if (g.has_log()) {
            const int64_t lt = g.last_term();
            return (term > lt || (term == lt && index >= g.last)) ? 1 : 0;
        }
        if ((index > g.epoch_index && term < g.epoch_term) || (index == g.epoch_index && term != g.epoch_term)) {
            fx.status = RG_A_IMPOSSIBLE_LOG;
            return -1;
        }
        return index >= g.epoch_index ? 1 : 0;
    }

    __device__ __forceinline__ void on_vote_request(bool pre, int64_t term, int32_t cand, int64_t last_index,
                                                    int64_t last_term)
    {
        if (g.role == RG_LEADER) {
            if (pre) { reply(g.term, false); return; }
            if (term < g.term) { reply(g.term, false); return; }
            if (term == g.term) {
                if (g.voted_for == p.self) { reply(g.term, false); return; }
                fx.status = RG_A_LEADER_NOT_SELF_VOTE; return;
            }
            if (switch_to(RG_FOLLOWER, g.term, cand) < 0) return;
            pre = false;
        } else if (g.role == RG_CANDIDATE) {
            if (cand == p.self) { fx.status = RG_A_CAND_SELF_RV; return; }
            if (term < g.term) { reply(g.term, false); return; }
            if (term == g.term) {
                if (cand != g.voted_for) { reply(g.term, false); return; }
                if (g.voted_for != p.self) { fx.status = RG_A_CAND_NOT_SELF_VOTE; return; }
            }
            if (switch_to(RG_FOLLOWER, term, cand) < 0) return;
            pre = false;
        }
        if (pre) {
            if (term <= g.term || !g.td) { reply(g.term, false); return; }
            fx.flags |= RG_F_RESET_TIMER;
            const int ok = log_up_to_date(last_index, last_term);
            if (ok < 0) return;
            reply(g.term, ok != 0);
            return;
        }
        if (term < g.term) { reply(g.term, false); return; }
        if (term == g.term) { reply(g.term, cand == g.voted_for); return; }
        fx.flags |= RG_F_RESET_TIMER;                     // :118 resetTimer(this, true); only the fresh Follower of :125 gets a live ticket
        const int ok = log_up_to_date(last_index, last_term);
        if (ok < 0) { fx.flags |= RG_F_TIMER_MUTED; return; }
        if (switch_to(RG_FOLLOWER, term, ok ? cand : RG_NO_NODE) < 0) return;
        reply(g.term, cand == g.voted_for);
    }

    // ---- Leader ack path ------------------------------------------------------------------------
    __device__ __forceinline__ void on_replicate_ack(bool snapshot, uint32_t slot, int64_t resp_term, bool success,
                                                     int64_t epoch_at_send, int64_t last_sent, uint32_t sent_epoch,
                                                     bool hinted, int64_t hint_index, int64_t hint_term)
    {
        if (sent_epoch != g.role_epoch) { fx.status = RG_DROPPED_STALE_ROLE; return; }
        if (g.role != RG_LEADER || !g.prepared) { fx.status = RG_BAD_EVENT; return; }
        if (resp_term > g.term) { switch_to(RG_FOLLOWER, resp_term, (int32_t)slot); return; }
        const int j = (int)(slot < (uint32_t)p.self ? slot : slot - 1u);

        // work on a register copy of this follower's State; committed to LDS only if the row applies
        int64_t s_epoch = pe.last_epoch[j * BLOCK], s_next = pe.next_index[j * BLOCK], s_match = pe.match_index[j * BLOCK];
        int32_t s_rej = pe.rejection[j * BLOCK];
        bool s_pend = (g.pending >> j) & 1u;

        s_rej = success ? 0 : (int32_t)((uint32_t)s_rej + 1u);          // statSuccess runs before updateIndex
        const int64_t index = snapshot ? epoch_at_send : last_sent;
        bool rollback = false;
        if (index < s_match) {
            rollback = true;                                            // AbstractMethodError: only the counter moved
        } else if (epoch_at_send >= s_epoch) {
            if (epoch_at_send > s_epoch) { s_epoch = epoch_at_send; s_next = s_next > epoch_at_send ? s_next : epoch_at_send; }
            if (s_pend == snapshot) {
                if (s_pend) {
                    if (success) { s_next = max64(s_next, wadd(epoch_at_send, 1)); s_pend = false; }
                } else if (success) {
                    if (index > s_match) { s_next = wadd(index, 1); s_match = index; }
                } else if (s_match == 0) {
                    const int64_t next = max64(wsub(s_next, rejection_step(s_rej)), wadd(epoch_at_send, 1));
                    s_next = min64(wsub(s_next, 1), next);
                }
                if (s_next <= epoch_at_send && !s_pend) s_pend = true;
            }
        }

        // tryCommit on the would-be matchIndex vector
        int64_t commit_to = 0;
        uint32_t commit_status = RG_OK;
        if (!rollback && !snapshot && success) {
            int64_t m[F];
#pragma unroll
            for (int i = 0; i < F; i++) m[i] = (i == j) ? s_match : pe.match_index[i * BLOCK];
            int64_t full, major;
            major_indices<F>(m, full, major);
            if (major != 0) {
                if (!g.present(major)) {
                    commit_status = RG_NPE_MAJOR_NULL;
                } else {
                    int64_t mt;
                    if (g.cached(major)) {
                        mt = g.term_at(major);
                    } else if (hinted && hint_index == major) {
                        mt = hint_term;
                    } else {
                        fx.status = RG_NEED_HOST; fx.log_from = major; return;   // nothing was written yet
                    }
                    commit_to = (mt == g.term) ? major : full;
                }
            }
        }

        pe.rejection[j * BLOCK] = s_rej;
        g.peers_dirty = true;
        if (rollback) { fx.status = RG_A_MATCH_ROLLBACK; return; }
        pe.last_epoch[j * BLOCK] = s_epoch;
        pe.next_index[j * BLOCK] = s_next;
        pe.match_index[j * BLOCK] = s_match;
        g.pending = (g.pending & ~(1u << j)) | ((s_pend ? 1u : 0u) << j);
        if (commit_status) { fx.status = commit_status; return; }
        if (commit_to != 0 && commit_to != g.commit) {
            const uint32_t st = mark_committed(commit_to);
            if (st) fx.status = st;
        }
    }

    // ---- tallies --------------------------------------------------------------------------------
    __device__ __forceinline__ void on_vote_reply(bool pre, uint32_t slot, int64_t resp_term, bool granted,
                                                  uint32_t sent_epoch)
    {
        if (sent_epoch == g.role_epoch) {
            const bool sender_ok = pre ? (g.role == RG_FOLLOWER && g.td) : (g.role == RG_CANDIDATE);
            if (!sender_ok) { fx.status = RG_BAD_EVENT; return; }
            const int64_t T = pre ? wadd(g.term, 1) : g.term;
            if (resp_term > T) {
                switch_to(RG_FOLLOWER, resp_term, (int32_t)slot);
            } else if (granted) {
                g.votes += 1;
                if (g.votes >= p.majority) {
                    if (pre) {
                        switch_to(RG_CANDIDATE, T, p.self);
                    } else {
                        g.elected_epoch = g.role_epoch;       // the winner's AsyncHead is never aborted (Candidate.java:75-79)
                        g.elected_term = g.term;
                        switch_to(RG_LEADER, T, p.self);
                    }
                }
            }
            return;
        }
        if (!pre && g.elected_epoch != 0u && sent_epoch == g.elected_epoch) {
            const int64_t T = g.elected_term;
            if (resp_term > T) {
                g.elected_epoch = 0u;
                switch_to(RG_FOLLOWER, resp_term, (int32_t)slot);
            } else if (granted) {
                switch_to(RG_LEADER, T, p.self);
            }
            return;
        }
        fx.status = RG_DROPPED_STALE_ROLE;
    }

    // RaftParticipant.installSnapshot: member/Follower.java:129-152, member/RaftMember.java:61-66 (Candidate, Leader)
    __device__ __forceinline__ void on_install_snapshot(int64_t term, bool host_ok)
    {
        if (g.role != RG_FOLLOWER) {
            if (term >= g.term) { fx.status = RG_A_INSTALL_BEFORE_AE; return; }
            reply(g.term, false);
            return;
        }
        fx.flags |= RG_F_RESET_TIMER;                     // :134 mutes BEFORE the term checks; :136-139 never un-mute
        if (term < g.term) { fx.flags |= RG_F_TIMER_MUTED; reply(g.term, false); return; }
        if (term > g.term) { fx.flags |= RG_F_TIMER_MUTED; fx.status = RG_A_INSTALL_BEFORE_AE; return; }
        if (g.td) { if (switch_to(RG_FOLLOWER, g.term, g.voted_for) < 0) return; }
        reply(g.term, host_ok);
    }

    __device__ __forceinline__ void on_timeout(uint32_t ticket_epoch)
    {
        // context/RaftRoutine.java:57,70: only the participant whose ticket fired runs onTimeout (0 = whoever is current)
        if (ticket_epoch != 0u && ticket_epoch != g.role_epoch) { fx.status = RG_DROPPED_STALE_ROLE; return; }
        if (g.role == RG_FOLLOWER) {
            if (p.pre_vote) {
                if (switch_to(RG_FOLLOWER, g.term, g.voted_for) < 0) return;
                g.td = true; g.votes = 1;
                fx.flags |= RG_EMIT_PREVOTE << RG_F_EMIT_SHIFT;
            } else {
                switch_to(RG_CANDIDATE, wadd(g.term, 1), p.self);
            }
        } else if (g.role == RG_CANDIDATE) {
            switch_to(RG_CANDIDATE, wadd(g.term, 1), p.self);
        } else {
            fx.flags |= RG_F_RESET_TIMER;
            prepare_replication();
            fx.flags |= RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT;
        }
    }

    __device__ __forceinline__ void on_client_append(uint32_t n)
    {
        if (g.role != RG_LEADER) { fx.status = RG_NOT_LEADER; return; }
        if (n == 0) return;
        if (!g.has_log() && g.epoch_index > 0) { fx.status = RG_UNSUPPORTED_LOG_STATE; return; }
        // n x RaftLog.newEntry(currentTerm); replicateLog(false) after each, so prepareReplication sees exactly one new entry
        const int64_t first_new = g.has_log() ? wadd(g.last, 1) : 1;
        fx.log_from = first_new;
        g.push(first_new, g.term);
        prepare_replication();
        if (n > 1) { g.last = wadd(g.last, (int64_t)n - 1); }
        fx.flags |= RG_F_LOG_APPEND | (RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT);
    }

    // ---- tier 1: branch-free fast paths -----------------------------------------------------------
    // A lone wavefront per SIMD pays ~40 cycles for every divergent branch of the general handlers below
    // (measured: ~3000 cycles for a plain heartbeat), so the common rows are decided here with selects only,
    // under explicit preconditions that make them a strict special case of the general code (which stays the
    // single source of truth for everything else):
    //   * AppendEntries at a Follower (term >= currentTerm; a higher term or a pending pre-vote refreshes the
    //     Follower first, member/Follower.java:45-47) whose prevLog is the log tail: no conflict, no purge,
    //     entries of one term (member/Follower.java:35-88);
    //   * AppendEntries ack at a prepared Leader: same role epoch, no term change, no epoch move, not pending
    //     (member/Leader.java:218-237, member/Leadership.java:75-114, member/Leader.java:247-280);
    //   * client append at a Leader with a non-empty log (member/Leader.java:128-140);
    //   * vote replies that only count (member/Candidate.java:127-128, member/Follower.java:264-265), late
    //     replies to a won election that change nothing (Q13), and responses to a fenced participant.
    // The two rare sub-cases that need more than selects (a new term run at the log tail, the first
    // prepareReplication of a new Leader) sit behind ONE wave-level branch. Any row that misses a precondition
    // is left untouched and goes to run().
    // NOTE: bool operands are combined with & and | (never && / ||): short-circuit operators are compiled back
    // into exec-mask branches, which is exactly what this tier exists to avoid.
    // pe0 = the first carried entry term; entries_ok / same = entries_readable() / entries_same_term() of the row;
    // ev_narrow = a, b, c, d, pe0 are all in [0, NARROW_LIMIT) (worked out where the event is loaded)
    __device__ __forceinline__ bool try_fast(bool allow, uint32_t hdr, uint32_t aux, int64_t a, int64_t b, int64_t c,
                                             int64_t d, int64_t pe0, bool entries_ok, bool same, bool ev_narrow)
    {
        if (narrow_tier && __builtin_amdgcn_ballot_w64(!(ev_narrow & g.narrow)) == 0)       // wave-uniform: all 64 rows fit the 32-bit tier
            return try_fast_v<int32_t>(allow, hdr, aux, (int32_t)a, (int32_t)b, (int32_t)c, (int32_t)d, (int32_t)pe0, entries_ok, same);
        const bool done = try_fast_v<int64_t>(allow, hdr, aux, a, b, c, d, pe0, entries_ok, same);
        refresh_narrow();                                                    // 64-bit values went in: the group may have left the domain
        return done;
    }

    // is every 64-bit value of the group (registers and LDS follower columns) in [0, NARROW_LIMIT)?
    __device__ __forceinline__ void refresh_narrow()
    {
        if (!narrow_tier) return;
        uint64_t w = (uint64_t)g.term | (uint64_t)g.commit | (uint64_t)g.epoch_index | (uint64_t)g.epoch_term | (uint64_t)g.first | (uint64_t)g.last |
                     (uint64_t)g.elected_term | (uint64_t)g.s0 | (uint64_t)g.s1 | (uint64_t)g.s2 | (uint64_t)g.s3 | (uint64_t)g.t0 | (uint64_t)g.t1 |
                     (uint64_t)g.t2 | (uint64_t)g.t3;
        if (g.prepared) {
#pragma unroll
            for (int i = 0; i < F; i++) w |= (uint64_t)pe.last_epoch[i * BLOCK] | (uint64_t)pe.next_index[i * BLOCK] | (uint64_t)pe.match_index[i * BLOCK];
        }
        g.narrow = w < NARROW_LIMIT;
    }

    // V = int64_t: any values. V = int32_t: the narrow tier (see NARROW_LIMIT) — same statements, 32-bit arithmetic.
    template <class V>
    __device__ __forceinline__ bool try_fast_v(bool allow, uint32_t hdr, uint32_t aux, V a, V b, V c, V d, V pe0, bool entries_ok, bool same)
    {
        constexpr bool NARROW = sizeof(V) == 4;
        auto vadd = [](V x, V y) -> V { typedef typename std::make_unsigned<V>::type U; return (V)((U)x + (U)y); };
        auto vmin = [](V x, V y) -> V { return x < y ? x : y; };
        auto keep = [](int64_t old, V v) -> int64_t { if constexpr (sizeof(V) == 4) return with_lo(old, (uint32_t)v); else return (int64_t)v; };
        const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr);
        const bool flag = RG_HDR_FLAG(hdr) != 0;
        const uint32_t P = (uint32_t)p.cluster, self = (uint32_t)p.self;
        const V g_term = (V)g.term, g_last = (V)g.last, g_commit = (V)g.commit, g_epoch = (V)g.epoch_index, g_first = (V)g.first;
        const V g_s0 = (V)g.s0, el_term = (V)g.elected_term;
        const V q0 = (V)g.t0, q1 = (V)g.t1, q2 = (V)g.t2, q3 = (V)g.t3, b1 = (V)g.s1, b2 = (V)g.s2, b3 = (V)g.s3;
        V lt = q0;                                                       // Group::last_term()
        lt = g.rc > 1 ? q1 : lt; lt = g.rc > 2 ? q2 : lt; lt = g.rc > 3 ? q3 : lt;
        const int32_t rc = g.rc, role = g.role, g_leader = g.leader, g_votes = g.votes;
        const uint32_t g_repoch = g.role_epoch, el_epoch = g.elected_epoch;
        const bool has_log = rc > 0, g_td = g.td, g_prep = g.prepared;
        const bool peer_ok = (slot < P) & (slot != self);

        // ---- AppendEntries request at a follower --------------------------------------------------
        const bool contains = c == lt;                                   // prevLogTerm == term of the tail
        const bool refresh = (a > g_term) | g_td;                        // switchTo(Follower, term, lastCandidate)
        const V ae_last = contains ? vadd(b, (V)n) : g_last;
        const bool want_commit = contains & (d > g_epoch);
        const V ae_x = vmin(d, ae_last);
        const bool fa = allow & (kind == RG_EV_AE_REQ) & (slot < P) & (role == RG_FOLLOWER) & (a >= g_term) &
                        (refresh | (g_leader == RG_NO_NODE) | (g_leader == (int32_t)slot)) & has_log & (b == g_last) &
                        (b > g_epoch) & (c != 0) & entries_ok & (!contains | (n == 0) | same) &
                        !(want_commit & (ae_x < g_commit));
        const bool ae_refresh = fa & refresh;
        const bool ae_commit = fa & want_commit & (ae_x > g_commit);
        const bool ae_append = fa & contains & (n > 0);
        const bool ae_newrun = ae_append & (pe0 != lt);

        // ---- AppendEntries ack at a leader ----------------------------------------------------------
        const bool ack_kind = (kind == RG_EV_AE_ACK) | (kind == RG_EV_IS_ACK);
        const bool ack_shape = allow & (kind == RG_EV_AE_ACK) & peer_ok;
        const uint32_t j = ack_shape ? (slot < self ? slot : slot - 1u) : 0u;
        const V s_epoch = (V)pe.last_epoch[j * BLOCK], s_next = (V)pe.next_index[j * BLOCK], s_match = (V)pe.match_index[j * BLOCK];
        const int32_t s_rej = pe.rejection[j * BLOCK];
        const bool s_pend = ((g.pending >> j) & 1u) != 0;
        const bool adv = flag & (c > s_match);
        const V n_match = adv ? c : s_match;
        const V n_next = adv ? vadd(c, 1) : s_next;
        V m[F];
#pragma unroll
        for (int i = 0; i < F; i++) {
            const V mi = (V)pe.match_index[i * BLOCK];
            m[i] = ((uint32_t)i == j) ? n_match : mi;
        }
        V full, major;
        major_indices_v<F, V>(m, full, major);
        const bool lookup = flag & (major != 0);
        const bool major_ok = has_log & (major >= g_first) & (major <= g_last) & (major >= g_s0);   // present and cached
        V mt = q0;                                                       // Group::term_at(major)
        mt = ((rc > 1) & (b1 <= major)) ? q1 : mt; mt = ((rc > 2) & (b2 <= major)) ? q2 : mt; mt = ((rc > 3) & (b3 <= major)) ? q3 : mt;
        const V commit_to = lookup ? (mt == g_term ? major : full) : (V)0;
        const bool do_commit = (commit_to != 0) & (commit_to != g_commit);
        const bool fk = ack_shape & (aux == g_repoch) & (role == RG_LEADER) & g_prep & (a <= g_term) &
                        (b == s_epoch) & !s_pend & (c >= s_match) & (flag | (s_match != 0)) & (n_next > b) &
                        (!lookup | major_ok) & !(do_commit & (commit_to < g_commit));
        const bool ack_commit = fk & do_commit;
        const bool ack_drop = allow & ack_kind & peer_ok & (aux != g_repoch);       // AsyncHead aborted: response dropped

        // ---- client append at a leader ----------------------------------------------------------------
        const bool fc = allow & (kind == RG_EV_CLIENT_APPEND) & (role == RG_LEADER) & (n >= 1u) & has_log;
        const bool fc_newrun = fc & (lt != g_term);
        const bool fc_prepare = fc & !g_prep;

        // ---- vote replies that only count, change nothing, or reach a fenced participant --------------
        const bool is_pv = kind == RG_EV_PV_REPLY;
        const bool vr_shape = allow & ((kind == RG_EV_RV_REPLY) | is_pv) & peer_ok;
        bool count_only = false, late_noop = false, vote_drop = false;
        if (__builtin_amdgcn_ballot_w64(vr_shape) != 0) {                // wave-uniform: steady replication carries no vote replies
            const bool cur_epoch = aux == g_repoch;
            const bool sender_ok = is_pv ? ((role == RG_FOLLOWER) & g_td) : (role == RG_CANDIDATE);
            const V T = is_pv ? vadd(g_term, 1) : g_term;
            count_only = vr_shape & cur_epoch & sender_ok & (a <= T) & (!flag | (g_votes + 1 < p.majority));
            const bool late = !is_pv & !cur_epoch & (el_epoch != 0u) & (aux == el_epoch);
            late_noop = vr_shape & late & (a <= el_term) &
                        (!flag | (el_term < g_term) | ((el_term == g_term) & (role == RG_LEADER)));
            vote_drop = vr_shape & !cur_epoch & !late;
        }
        const bool fv = count_only | late_noop | vote_drop;

        const bool fast = fa | fk | fc | fv | ack_drop;
        if (fk) {
            pe.rejection[j * BLOCK] = flag ? 0 : (int32_t)((uint32_t)s_rej + 1u);
            if constexpr (NARROW) {                       // the high words in LDS are zero already
                reinterpret_cast<int32_t *>(&pe.next_index[j * BLOCK])[0] = n_next;
                reinterpret_cast<int32_t *>(&pe.match_index[j * BLOCK])[0] = n_match;
            } else {
                pe.next_index[j * BLOCK] = n_next;
                pe.match_index[j * BLOCK] = n_match;
            }
        }
        if (ae_newrun | fc_newrun | fc_prepare) {          // rare: one wave-level branch for both
            if (ae_newrun | fc_newrun) g.push((int64_t)vadd(g_last, 1), (int64_t)(ae_newrun ? pe0 : g_term));
            if (fc_prepare) {                             // Leader.prepareReplication after the FIRST new entry
                const int64_t next = (int64_t)vadd(g_last, 2);
#pragma unroll
                for (int i = 0; i < F; i++) {
                    pe.last_epoch[i * BLOCK] = (int64_t)g_epoch; pe.next_index[i * BLOCK] = next;
                    pe.match_index[i * BLOCK] = 0; pe.rejection[i * BLOCK] = 0;
                }
                g.pending = 0;
            }
        }
        g.prepared = g_prep | fc_prepare;
        g.peers_dirty = g.peers_dirty | fk | fc_prepare;
        g.term = keep(g.term, ae_refresh ? a : g_term);
        g.role_epoch = g_repoch + (ae_refresh ? 1u : 0u);
        g.td = g_td & !ae_refresh;
        g.votes = ae_refresh ? 1 : (g_votes + ((count_only & flag) ? 1 : 0));
        g.leader = fa ? (int32_t)slot : g_leader;
        {
            const V cur_last = (V)g.last;                 // (a new run pushed above has already moved it)
            g.last = keep(g.last, ae_append ? ae_last : (fc ? vadd(g_last, (V)n) : cur_last));
        }
        g.log_dirty = g.log_dirty | ae_append | fc;
        g.commit = keep(g.commit, ae_commit ? ae_x : (ack_commit ? commit_to : g_commit));
        if (fast) {
            fx.status = (ack_drop | vote_drop) ? RG_DROPPED_STALE_ROLE : RG_OK;
            fx.resp_term = (int64_t)a;                   // only read for AppendEntries: the request term (== currentTerm by now)
            fx.log_from = (int64_t)vadd(g_last, 1);
            fx.flags = (fa ? (RG_F_RESET_TIMER | RG_F_REPLIED | (contains ? RG_F_SUCCESS : 0u)) : 0u) |
                       (ae_refresh ? (RG_F_PERSIST | RG_F_ROLE_CHANGED) : 0u) |
                       ((ae_append | fc) ? RG_F_LOG_APPEND : 0u) | ((ae_commit | ack_commit) ? RG_F_COMMIT : 0u) |
                       (fc ? (RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT) : 0u);
        }
        return fast;
    }

    // ---- one row (tier 2: the general handlers) ---------------------------------------------------
    __device__ __forceinline__ void run(uint32_t hdr, uint32_t aux, int64_t a, int64_t b, int64_t c, int64_t d,
                                        int64_t hx, int64_t hy, int64_t pe0, int64_t pe1, int64_t pe2, int64_t pe3)
    {
        fx = Fx{0u, RG_OK, 0, 0};
        const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr);
        const bool flag = RG_HDR_FLAG(hdr) != 0, hinted = (RG_HDR_HINT(hdr) != 0) & (p.hint != nullptr);   // no hint column: the bit means nothing
        const uint32_t P = (uint32_t)p.cluster;
        switch (kind) {
        case RG_EV_NONE:
            break;
        case RG_EV_AE_REQ:
            if (slot >= P || n > RG_MAX_AE_ENTRIES || (n > 0 && (p.entry_terms == nullptr || (uint64_t)aux + n > p.entry_count))) {
                fx.status = RG_BAD_EVENT; break;
            }
            on_append_entries(a, (int32_t)slot, b, c, n, p.entry_terms + aux, Pre{pe0, pe1, pe2, pe3}, d, hinted, hx, hy);
            break;
        case RG_EV_AE_ACK:
        case RG_EV_IS_ACK:
            if (slot >= P || slot == (uint32_t)p.self) { fx.status = RG_BAD_EVENT; break; }
            on_replicate_ack(kind == RG_EV_IS_ACK, slot, a, flag, b, c, aux, hinted, hx, hy);
            break;
        case RG_EV_RV_REQ:
        case RG_EV_PV_REQ:
            if (slot >= P) { fx.status = RG_BAD_EVENT; break; }
            on_vote_request(kind == RG_EV_PV_REQ, a, (int32_t)slot, b, c);
            break;
        case RG_EV_RV_REPLY:
        case RG_EV_PV_REPLY:
            if (slot >= P || slot == (uint32_t)p.self) { fx.status = RG_BAD_EVENT; break; }
            on_vote_reply(kind == RG_EV_PV_REPLY, slot, a, flag, aux);
            break;
        case RG_EV_TIMEOUT:
            on_timeout(aux);
            break;
        case RG_EV_IS_REQ:
            if (slot >= P) { fx.status = RG_BAD_EVENT; break; }
            on_install_snapshot(a, flag);
            break;
        case RG_EV_CLIENT_APPEND:
            on_client_append(n);
            break;
        case RG_EV_LOG_FLUSH: {
            const uint32_t st = g.flush(a, b);
            if (st) fx.status = st;
            break;
        }
        default:
            fx.status = RG_BAD_EVENT;
            break;
        }
    }
};

}  // namespace rg
