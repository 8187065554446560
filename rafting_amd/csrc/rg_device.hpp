// rg_device.hpp — device-side decision logic of the batched multi-Raft engine (gfx950 / CDNA4).
//
// One LANE owns one raft group for the whole launch: the group's scalar state lives in VGPRs, its
// per-follower replication state (Leadership.State) is staged in LDS so the responder slot of an ack —
// a runtime value — indexes LDS instead of forcing the register file through scratch.  Integer work
// only; no MFMA.
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
//
// Two tiers decide a row:
//   tier 2  Stepper::run — the general handlers, one function per reference method, branchy, 64-bit: the single
//           source of truth;
//   tier 1  tier1<V> — the rows a cluster produces in operation (AppendEntries at a follower whose prevLog is the log tail,
//           acks at a prepared leader, client appends, the whole election traffic: timeouts, vote requests at a follower,
//           vote replies that count / win / carry a higher term / arrive late, a leader stepping down on a higher-term ack)
//           with selects only, under preconditions that make every one of them a strict special case of tier 2. Instantiated at
//           V = int64_t by the 64-bit bodies. The compact-format kernel's 32-bit body has its own statement of the same classes, in
//           sign words: rg_tier1n.hpp (tier1n), used while every value of a workgroup's groups and rows is below 2^30.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/raftgpu.h"

namespace rg {

constexpr int K = RG_TERM_RUNS;
constexpr int BLOCK = 64;          // raft groups per workgroup = lanes of a wavefront

struct alignas(16) I64x2 { int64_t x, y; };
struct alignas(16) I32x4 { int32_t x, y, z, w; };
struct alignas(8) U32x2 { uint32_t x, y; };
struct alignas(16) Ident { int32_t voted_for, leader; uint32_t role_epoch, meta; };
struct alignas(16) Elect { int64_t elected_term; uint32_t elected_epoch; int32_t votes; };
struct alignas(16) Match { int64_t match_index; int32_t rejection; int32_t pad; };

// meta word: role[1:0] timeoutDetected[2] replPrepared[3] runCount[6:4] pendingInstallation bit per follower [14:8]
constexpr uint32_t META_ROLE = 3u, META_TD = 1u << 2, META_PREP = 1u << 3;
constexpr int META_RC_SHIFT = 4, META_PEND_SHIFT = 8;
constexpr uint32_t META_PEND_MASK = (1u << (RG_MAX_CLUSTER - 1)) - 1u;      // pendingInstallation, one bit per follower (bits 8 .. 21 of meta)

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
    int64_t *ibase;       // [G] index base of the compact formats (rg_index_base_set; 0 unless the host set one): see to_rel()
    uint32_t groups;
};

struct StepParams {
    DevTable t;
    uint32_t rounds, count;
    const uint32_t *gid;            // sparse only
    const rg_ev_head_t *head;
    const I64x2 *ab, *cd, *hint;    // wide rows (rg_batch_t)
    const int64_t *entry_terms;
    const I32x4 *abcd32;            // compact rows (rg_batch32_t): a, b, c, d as int32; entry_terms32 instead of entry_terms
    const int32_t *entry_terms32;
    uint64_t entry_count;
    rg_reply_t *reply;
    I64x2 *logfx;
    rg_persist_t *persist;
    I32x4 *out32, *persist32;       // compact outcome rows (rg_submit32c): rg_out32_t always, rg_persist32_t iff PERSIST; reply / logfx / persist are then
                                    // the optional overflow columns of rows flagged RG_F_WIDE_VALUES (all three, or all null)
    unsigned long long *counters;   // [workgroups][RG_NUM_COUNTERS]
    unsigned long long *wide_bodies;// one word: workgroups of compact-row launches that were decided by the 64-bit body (rg_wide_body_workgroups)
    int32_t self, cluster, majority, pre_vote;
    int32_t fast_paths;             // 0: general handlers only
    int32_t force_wide;             // compact-format kernel: skip the 32-bit body (tests)
    int32_t require_fence;          // RG_OPT_REQUIRE_FENCED_TIMEOUTS: a TIMEOUT row with aux == 0 is RG_BAD_EVENT
    int32_t has_bases;              // the host has set a non-zero index base at some time (rg_index_base_set): 0 = the ibase column is all zero and is not read
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
    uint32_t *epoch;                 // [G] role epoch of the group after the last batch the timers saw (ABI 5): what a compact outcome row, which
                                     //     names its role epoch only where a conversion happened, is chained from
    const Ident *ident;              // role / role epoch of the table (arm only)
    uint32_t groups, rounds, count;
    const uint32_t *gid;
    const rg_reply_t *reply;
    const I32x4 *out32, *persist32;  // compact outcome rows (rg_timers_update32): rg_out32_t / rg_persist32_t, dense
    const int64_t *now_mem;          // non-null: the per-round timestamps are READ from device-visible memory (a recorded tick is replayed with new clocks)
    int64_t now[64];                 // else: per-round timestamps of one update launch (<= 64 rounds per launch)
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
    const I32x4 *out32;              // compact outcome rows instead of `reply` (rg_health_update32)
    const int64_t *now_mem;          // as TimerParams.now_mem
    int64_t now[64];
};

struct TickFoldParams {              // the device-resident tick's second kernel (rg_tick2: tick_fold_kernel): timers + health + the list of fired tickets
    TimerParams tp;                  // out32 / persist32 / now_mem set; rounds, count = G
    HealthParams hp;                 // head / out32 / now_mem set
    const int64_t *now_last;         // &now[rounds - 1]: the clock of the expiry
    unsigned long long *masks;       // [waves] which lanes of a wavefront expired (written by every workgroup, read by the last one)
    uint32_t *ticket;                // [1] workgroups done; the last one to arrive emits the list and puts it back to 0
    uint32_t *out_gid, *out_epoch, *out_count;
    uint32_t capacity;
    int expire;                      // 0: no expiry step
};
struct TickTailParams {              // everything a recorded tick does after the decisions, one launch (rg_kernels.hip: tick_tail_kernel)
    TickFoldParams fp;
    ReplicateParams qp;              // head == nullptr: no send side
    HealthParams rp;                 // now_mem = &now[rounds - 1]
    int32_t critical_point;
    int64_t cool_down;
    uint8_t *ready;                  // nullptr: no readiness column
};

__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__device__ __forceinline__ int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
__device__ __forceinline__ int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }
// two's-complement add in the width of V (Java long arithmetic wraps; the 32-bit tier never gets near its limit)
template <class V> __device__ __forceinline__ V vadd(V x, V y) { typedef typename std::make_unsigned<V>::type U; return (V)((U)x + (U)y); }
template <class V> __device__ __forceinline__ V vmin(V x, V y) { return x < y ? x : y; }
template <class V> __device__ __forceinline__ V vmax(V x, V y) { return x > y ? x : y; }

// The 32-bit tier's domain. Event fields must lie in [0, EV_LIMIT), group state in [0, STATE_LIMIT): tier 1 adds at most
// RG_MAX_ENTRIES (< 2^20) to a value per row, so nothing it computes from such inputs can reach 2^31, and every sum it forms equals
// the 64-bit one. A row or a state outside the domain sends its WORKGROUP back to the 64-bit body (step32_kernel).
constexpr uint32_t EV_LIMIT = 1u << 30;
constexpr uint32_t STATE_LIMIT = (1u << 30) + (1u << 29);

// State.majorIndices (member/Leadership.java:116-130): sorted[0] ("full") and sorted[F/2] ("major") of the F follower
// matchIndex values. One 64-bit compare per compare-exchange; for F = 4 (five nodes) only the two order statistics
// that are used get completed: pairs, then min of the minima / max of the maxima, then the larger of the middle two.
template <class V>
__device__ __forceinline__ void cmp_exchange(V &a, V &b)
{
    const bool sw = a > b;
    const V lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}
template <int F, class V>
__device__ __forceinline__ void major_indices(V (&m)[F], V &full, V &major)
{
    if constexpr (F == 4 && sizeof(V) == 4) {                           // 32-bit: the ISA has integer min / max / max3
        const V lo1 = vmin(m[0], m[1]), hi1 = vmax(m[0], m[1]), lo2 = vmin(m[2], m[3]), hi2 = vmax(m[2], m[3]);
        full = vmin(lo1, lo2);
        major = vmax(vmax(lo1, lo2), vmin(hi1, hi2));                   // second largest of four
    } else if constexpr (F == 4) {
        cmp_exchange(m[0], m[1]); cmp_exchange(m[2], m[3]);
        cmp_exchange(m[0], m[2]); cmp_exchange(m[1], m[3]);
        full = m[0]; major = m[1] > m[2] ? m[1] : m[2];
    } else {
#pragma unroll
        for (int a = 1; a < F; a++) {                                   // insertion network (any F: clusters of up to 15 nodes)
#pragma unroll
            for (int b = a; b > 0; b--) cmp_exchange(m[b - 1], m[b]);
        }
        full = m[0]; major = m[F / 2];
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

// effects of one row
template <class V>
struct FxT {
    uint32_t flags;
    uint32_t status;
    V resp_term;
    V log_from;
};
typedef FxT<int64_t> Fx;

// Facts about a row that do not depend on the group's state are worked out where the event is loaded (the I/O wavefront of the
// two-wavefront kernels) and handed over in header bits 10, 11 of the LDS / register copy (on the wire bit 10 is RG_HDR_SAME_TERM,
// bit 11 is unused; the loader consumes and replaces them):
//   HDR_AE_OK    AppendEntries request that tier 1 may decide: leader slot in range, prevLogTerm != 0, and either no entries or
//                <= RG_MAX_AE_ENTRIES entries that are readable and all of one term (the term itself travels as `pe0`)
//   HDR_PEER_OK  slot names a remote peer (slot < P, slot != self)
//   HDR_WIDE     compact rows only (they carry no hints, so the copy reuses the hint bit): a field lies outside [0, EV_LIMIT)
constexpr uint32_t HDR_AE_OK = 1u << 10, HDR_PEER_OK = 1u << 11, HDR_WIDE = 1u << 9;

// Register image of one group, in the width V of its terms and indices.  The K=4 cached term runs are SCALAR members on
// purpose: with arrays the optimiser folds the select chains below back into dynamically indexed accesses, which pins the
// whole struct in scratch memory (measured: 168 B/lane of scratch, every field access a scratch_load).
template <class V>
struct GroupT {
    V term, commit, epoch_index, epoch_term, first, last, elected_term;
    V s0, s1, s2, s3;                // run starts, ascending; valid [0, rc)
    V t0, t1, t2, t3;                // run terms
    V lt, top;                       // term / start of the NEWEST run (== t[rc-1], s[rc-1]); kept by push / truncate / flush, meaningless while rc == 0
    int32_t voted_for, leader, votes, role, rc;
    uint32_t role_epoch, elected_epoch, pending;
    bool td, prepared, log_dirty, peers_dirty;
    static_assert(K == 4, "run cache is hand-unrolled for 4 runs");

    // NOTE on style: every method first copies the fields it needs into locals, computes with value
    // selects, and stores back unconditionally.  Member functions are optimised BEFORE they are inlined
    // into the kernel, while `this` is still an opaque pointer; a load inside a conditional arm there is
    // folded into "load of a selected address", which later defeats scalar replacement of the struct.
    __device__ __forceinline__ bool has_log() const { return rc > 0; }
    __device__ __forceinline__ bool present(V i) const
    {
        const int n = rc; const V f = first, l = last;
        return n > 0 && i >= f && i <= l;
    }
    __device__ __forceinline__ bool cached(V i) const { return i >= s0; }
    __device__ __forceinline__ V last_term() const { return lt; }
    __device__ __forceinline__ void refresh_tail()
    {
        const int n = rc; const V a0 = t0, a1 = t1, a2 = t2, a3 = t3, b0 = s0, b1 = s1, b2 = s2, b3 = s3;
        V t = a0, s = b0;
        t = n > 1 ? a1 : t; s = n > 1 ? b1 : s;
        t = n > 2 ? a2 : t; s = n > 2 ? b2 : s;
        t = n > 3 ? a3 : t; s = n > 3 ? b3 : s;
        lt = t; top = s;
    }
    // term of a PRESENT and CACHED index
    __device__ __forceinline__ V term_at(V i) const
    {
        const int n = rc; const V a0 = t0, a1 = t1, a2 = t2, a3 = t3, b1 = s1, b2 = s2, b3 = s3;
        V t = a0;
        t = (n > 1 && b1 <= i) ? a1 : t;
        t = (n > 2 && b2 <= i) ? a2 : t;
        t = (n > 3 && b3 <= i) ? a3 : t;
        return t;
    }
    // db.put(last+1, t) — or the first key of an empty log
    __device__ __forceinline__ void push(V index, V t)
    {
        int n = rc;
        V x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, y1 = t1, y2 = t2, y3 = t3;
        const V f = first, cur_lt = lt, cur_top = top;
        const bool empty = n == 0;
        const bool newrun = empty || cur_lt != t;
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
        lt = t; top = newrun ? index : cur_top;
        rc = n;
        first = empty ? index : f;
        last = index;
        log_dirty = true;
    }
    // RaftLog.truncate(index): storage/RocksLog.java:219-225. `keep_term` = term of index-1, used only when
    // the cut lands below the cached runs (hint-resolved conflict) and the cache must be re-seeded.
    __device__ __forceinline__ void truncate(V index, V keep_term)
    {
        const int n0 = rc;
        const V x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, f = first, l = last;
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
        refresh_tail();
    }
    // RaftLog.flush(index, term): storage/RocksLog.java:228-242
    __device__ __forceinline__ uint32_t flush(V index, V term_)
    {
        int n = rc;
        V x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, y1 = t1, y2 = t2, y3 = t3;
        const V f = first, l = last, ei = epoch_index;
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
        refresh_tail();
        return RG_OK;
    }
};
typedef GroupT<int64_t> Group;
// every term / index of the image is in [0, limit)
__device__ __forceinline__ bool fits32(const Group &g, uint32_t limit)
{
    // a value is in range iff it is non-negative and below the limit; the maximum of the unsigned images tells both at once
    auto mx = [](uint64_t x, uint64_t y) { return x > y ? x : y; };
    uint64_t w = mx(mx(mx((uint64_t)g.term, (uint64_t)g.commit), mx((uint64_t)g.epoch_index, (uint64_t)g.epoch_term)),
                    mx(mx((uint64_t)g.first, (uint64_t)g.last), (uint64_t)g.elected_term));
    w = mx(w, mx(mx(mx((uint64_t)g.s0, (uint64_t)g.s1), mx((uint64_t)g.s2, (uint64_t)g.s3)), mx(mx((uint64_t)g.t0, (uint64_t)g.t1), mx((uint64_t)g.t2, (uint64_t)g.t3))));
    return w < (uint64_t)limit;
}
// ---- the index base of the compact formats (ABI 4) -------------------------------------------------------------------------
// Every quantity of the path is a Java long (command/RaftLog.java:72-132), and a group that lives long enough pushes its log indices past
// 2^30 — which would send its workgroup to the 64-bit body for good. Terms stay small (one per election); only INDICES grow. So the compact
// formats and the 32-bit image carry an index x of group g RELATIVE to a base the host sets for that group (rg_index_base_set):
//     0 travels as 0 ("none": matchIndex of a follower that has not answered, leaderCommit of a fresh leader, prevLogIndex of an empty log),
//     any other x as x - base[g], which must lie in [1, 2^30) for the 32-bit body (in [1, 2^31) for the format).
// Order, equality, "is it zero" and adding a positive count to a non-zero index all survive the mapping as long as EVERY non-zero index of
// the group lies above the base — what tier 1 of the 32-bit body (rg_tier1n.hpp) does with indices is exactly that; the one place that
// computes from a possibly-zero index, prepareReplication's `epoch.index + 1`, is covered by requiring epoch.index > base where base != 0.
// The general handlers run on ABSOLUTE values (converted on the way in and out), the table holds absolute int64 state, and anything that does not
// fit sends the workgroup to the 64-bit body as before. base = 0 (the default) is the format of ABI 3, bit for bit.
__device__ __forceinline__ int64_t to_rel(int64_t x, int64_t base) { const int64_t d = x - base; return x == 0 ? 0 : (d == 0 ? -1 : d); }      // (x == base != 0 has no image: -1 fails every range check)
__device__ __forceinline__ int64_t to_abs(int64_t r, int64_t base) { return r == 0 ? 0 : r + base; }
template <class Fn> __device__ __forceinline__ Group map_indices(const Group &g, Fn f)
{
    Group r = g;
    r.commit = f(g.commit); r.epoch_index = f(g.epoch_index); r.first = f(g.first); r.last = f(g.last);
    r.s0 = f(g.s0); r.s1 = f(g.s1); r.s2 = f(g.s2); r.s3 = f(g.s3); r.top = f(g.top);
    return r;
}
__device__ __forceinline__ Group group_to_rel(const Group &g, int64_t base) { return map_indices(g, [base](int64_t x) { return to_rel(x, base); }); }
__device__ __forceinline__ Group group_to_abs(const Group &g, int64_t base) { return map_indices(g, [base](int64_t x) { return to_abs(x, base); }); }
// which of a row's fields a, b, c, d (bits 0..3) are log indices, by event kind: AE_REQ b d, AE_ACK b c, IS_ACK b, RV_REQ / PV_REQ b, LOG_FLUSH a, IS_REQ b
__device__ __forceinline__ uint32_t index_fields(uint32_t kind)
{
    constexpr uint64_t LUT = (0xAull << (4 * RG_EV_AE_REQ)) | (0x6ull << (4 * RG_EV_AE_ACK)) | (0x2ull << (4 * RG_EV_IS_ACK)) | (0x2ull << (4 * RG_EV_RV_REQ)) |
                             (0x2ull << (4 * RG_EV_PV_REQ)) | (0x1ull << (4 * RG_EV_LOG_FLUSH)) | (0x2ull << (4 * RG_EV_IS_REQ));
    return (uint32_t)(LUT >> (4u * (kind & 15u))) & 15u;
}

// ---- Leadership.State of this lane's group in LDS -------------------------------------------------------------------
// Wide: four [follower][lane] columns of 64-bit values. Narrow (32-bit body): one 16-byte record {lastEpoch, nextIndex, matchIndex,
// recentRejection} per follower as [follower][lane], plus the F matchIndex values again as one [lane] row of 16 (F <= 4) or 32
// bytes, so that the quorum select reads them with one or two ds_read_b128 instead of F. The general handlers use the scalar
// accessors; tier 1 the bulk ones.
template <int F>
struct PeersWide {
    int64_t *e, *n, *m;              // + lane already applied
    int32_t *r;
    bool overflow;                   // (narrow only; kept so both have the same shape)
    __device__ __forceinline__ int64_t last_epoch(int j) const { return e[j * BLOCK]; }
    __device__ __forceinline__ int64_t next_index(int j) const { return n[j * BLOCK]; }
    __device__ __forceinline__ int64_t match_index(int j) const { return m[j * BLOCK]; }
    __device__ __forceinline__ int32_t rejection(int j) const { return r[j * BLOCK]; }
    __device__ __forceinline__ void set_last_epoch(int j, int64_t v) { e[j * BLOCK] = v; }
    __device__ __forceinline__ void set_next_index(int j, int64_t v) { n[j * BLOCK] = v; }
    __device__ __forceinline__ void set_match_index(int j, int64_t v) { m[j * BLOCK] = v; }
    __device__ __forceinline__ void set_rejection(int j, int32_t v) { r[j * BLOCK] = v; }
    // tier 1
    __device__ __forceinline__ void load_state(uint32_t j, int64_t &ep, int64_t &nx, int64_t &mt, int32_t &rj) const
    {
        ep = e[j * BLOCK]; nx = n[j * BLOCK]; mt = m[j * BLOCK]; rj = r[j * BLOCK];
    }
    __device__ __forceinline__ void load_matches(int64_t (&mm)[F]) const
    {
#pragma unroll
        for (int i = 0; i < F; i++) mm[i] = m[i * BLOCK];
    }
    __device__ __forceinline__ void store_ack(uint32_t j, int64_t ep, int64_t nx, int64_t mt, int32_t rj)
    {
        (void)ep; n[j * BLOCK] = nx; m[j * BLOCK] = mt; r[j * BLOCK] = rj;
    }
    __device__ __forceinline__ void store_prepare(int64_t ep, int64_t nx)
    {
#pragma unroll
        for (int i = 0; i < F; i++) { e[i * BLOCK] = ep; n[i * BLOCK] = nx; m[i * BLOCK] = 0; r[i * BLOCK] = 0; }
    }
};

template <int F>
struct PeersNarrow {
    static constexpr int MV = (F + 3) / 4;      // 16-byte pieces of the matchIndex row
    I32x4 *rec;                                 // [F][BLOCK], + lane applied
    int32_t *mv;                                // [MV][BLOCK][4], + lane * 4 applied; element i at mv[(i / 4) * BLOCK * 4 + (i % 4)]
    bool overflow;                              // a general handler stored a value that does not fit: the workgroup redoes the launch in 64 bits
    int64_t base;                               // the records hold indices relative to the group's index base (to_rel); the scalar accessors the general handlers use speak absolute values
    __device__ __forceinline__ int32_t *mslot(int i) const { return mv + (i >> 2) * (BLOCK * 4) + (i & 3); }
    __device__ __forceinline__ int64_t last_epoch(int j) const { return to_abs(rec[j * BLOCK].x, base); }
    __device__ __forceinline__ int64_t next_index(int j) const { return to_abs(rec[j * BLOCK].y, base); }
    __device__ __forceinline__ int64_t match_index(int j) const { return to_abs(rec[j * BLOCK].z, base); }
    __device__ __forceinline__ int32_t rejection(int j) const { return rec[j * BLOCK].w; }
    __device__ __forceinline__ int32_t fit(int64_t v) { const int64_t r = to_rel(v, base); overflow = overflow | ((uint64_t)r >= (uint64_t)STATE_LIMIT); return (int32_t)r; }
    __device__ __forceinline__ void set_last_epoch(int j, int64_t v) { rec[j * BLOCK].x = fit(v); }
    __device__ __forceinline__ void set_next_index(int j, int64_t v) { rec[j * BLOCK].y = fit(v); }
    __device__ __forceinline__ void set_match_index(int j, int64_t v) { const int32_t w = fit(v); rec[j * BLOCK].z = w; *mslot(j) = w; }
    __device__ __forceinline__ void set_rejection(int j, int32_t v) { rec[j * BLOCK].w = v; }
    // tier 1
    __device__ __forceinline__ void load_state(uint32_t j, int32_t &ep, int32_t &nx, int32_t &mt, int32_t &rj) const
    {
        const I32x4 v = rec[j * BLOCK];
        ep = v.x; nx = v.y; mt = v.z; rj = v.w;
    }
    __device__ __forceinline__ void load_matches(int32_t (&mm)[F]) const
    {
#pragma unroll
        for (int q = 0; q < MV; q++) {
            const I32x4 v = *reinterpret_cast<const I32x4 *>(mv + q * (BLOCK * 4));
            if (q * 4 + 0 < F) mm[q * 4 + 0] = v.x;
            if (q * 4 + 1 < F) mm[q * 4 + 1] = v.y;
            if (q * 4 + 2 < F) mm[q * 4 + 2] = v.z;
            if (q * 4 + 3 < F) mm[q * 4 + 3] = v.w;
        }
    }
    __device__ __forceinline__ void store_ack(uint32_t j, int32_t ep, int32_t nx, int32_t mt, int32_t rj)
    {
        rec[j * BLOCK] = I32x4{ep, nx, mt, rj};
        *mslot((int)j) = mt;
    }
    __device__ __forceinline__ void store_prepare(int32_t ep, int32_t nx)
    {
#pragma unroll
        for (int i = 0; i < F; i++) rec[i * BLOCK] = I32x4{ep, nx, 0, 0};
#pragma unroll
        for (int q = 0; q < MV; q++) *reinterpret_cast<I32x4 *>(mv + q * (BLOCK * 4)) = I32x4{0, 0, 0, 0};
    }
};

// entry terms of an AppendEntries request as the general handlers read them: `same` = every carried entry has term e0 (nothing else is
// read); otherwise the first four may have been prefetched with the event (wide rows) and the rest is read from the batch's entry stream,
// 64-bit (rg_batch_t) or 32-bit (rg_batch32_t)
struct Entries {
    const int64_t *t64;
    const int32_t *t32;
    int64_t e0, e1, e2, e3;          // scalars, not an array: see the note in GroupT
    bool same, prefetched;
    __device__ __forceinline__ int64_t term(uint64_t k) const
    {
        // fields to locals first, value selects after (the note in GroupT: a select between member loads becomes an indexed load of `this`)
        const int64_t a0 = e0, a1 = e1, a2 = e2, a3 = e3;
        const int64_t *p64 = t64;
        const int32_t *p32 = t32;
        const bool sm = same, pf = prefetched;
        int64_t t = a0;
        t = k == 1 ? a1 : t;
        t = k == 2 ? a2 : t;
        t = k == 3 ? a3 : t;
        if (!sm && !(pf && k < 4)) t = p64 ? p64[k] : (int64_t)p32[k];
        return sm ? a0 : t;
    }
};

template <int F, class PE>
struct Stepper {
    const StepParams &p;
    Group &g;
    PE &pe;
    Fx fx;

    __device__ __forceinline__ Stepper(const StepParams &p_, Group &g_, PE &pe_) : p(p_), g(g_), pe(pe_) {}

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
            pe.set_last_epoch(j, g.epoch_index);
            pe.set_next_index(j, next);
            pe.set_match_index(j, 0);
            pe.set_rejection(j, 0);
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
    __device__ __forceinline__ void on_append_entries(int64_t term, int32_t leader, int64_t prev_index,
                                                      int64_t prev_term, uint32_t n, const Entries en,
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
                    if (g.term_at(idx) != en.term((uint64_t)(idx - e_first))) { conflict = idx; break; }
                }
            }
            if (conflict) {
                // term of the key just below the cut; only needed when the cut empties the cached runs
                // (hint-resolved conflict below the cache) while the log itself stays non-empty
                int64_t keep = 0;
                if (g.has_log() && conflict > g.first && conflict <= g.s0)
                    keep = (conflict == e0) ? (purged ? g.epoch_term : prev_term)
                                            : en.term((uint64_t)(conflict - 1 - e_first));
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
            for (int64_t idx = from; idx <= e_last; idx++) g.push(idx, en.term((uint64_t)(idx - e_first)));
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
        int64_t s_epoch = pe.last_epoch(j), s_next = pe.next_index(j), s_match = pe.match_index(j);
        int32_t s_rej = pe.rejection(j);
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
            for (int i = 0; i < F; i++) m[i] = (i == j) ? s_match : pe.match_index(i);
            int64_t full, major;
            major_indices<F, int64_t>(m, full, major);
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

        pe.set_rejection(j, s_rej);
        g.peers_dirty = true;
        if (rollback) { fx.status = RG_A_MATCH_ROLLBACK; return; }
        pe.set_last_epoch(j, s_epoch);
        pe.set_next_index(j, s_next);
        pe.set_match_index(j, s_match);
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
        // context/RaftRoutine.java:57,70: only the participant whose ticket fired runs onTimeout (0 = whoever is current — refused when the table
        // requires fenced timeouts: RG_OPT_REQUIRE_FENCED_TIMEOUTS)
        if (ticket_epoch == 0u && p.require_fence != 0) { fx.status = RG_BAD_EVENT; return; }
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

    // ---- one row (tier 2: the general handlers) ---------------------------------------------------
    // hdr: kind / slot / flag / n as on the wire (other bits are not looked at); en: how this row's entry terms are read
    __device__ __forceinline__ void run(uint32_t hdr, uint32_t aux, int64_t a, int64_t b, int64_t c, int64_t d,
                                        bool hinted, int64_t hx, int64_t hy, const Entries en, bool entries_readable)
    {
        fx = Fx{0u, RG_OK, 0, 0};
        const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr);
        const bool flag = RG_HDR_FLAG(hdr) != 0;
        const uint32_t P = (uint32_t)p.cluster;
        switch (kind) {
        case RG_EV_NONE:
            break;
        case RG_EV_AE_REQ:
            if (slot >= P || n > RG_MAX_AE_ENTRIES || (n > 0 && !entries_readable)) {
                fx.status = RG_BAD_EVENT; break;
            }
            on_append_entries(a, (int32_t)slot, b, c, n, en, d, hinted, hx, hy);
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

// ---- tier 1: the rows of a cluster in operation, with selects only ----------------------------------------------------------
// A lone wavefront per SIMD pays ~40 cycles for every taken branch of the general handlers (measured in round 2: ~1 700 ticks for a
// visit by ONE lane, which its 63 neighbours wait for), so every row class a healthy or an electing cluster produces is decided here
// under explicit preconditions that make it a strict special case of the general code:
//   * AppendEntries at a Follower (term >= currentTerm; a higher term or a pending pre-vote refreshes the Follower first,
//     member/Follower.java:45-47) whose prevLog is the log tail: no conflict, no purge, entries of one term (member/Follower.java:35-88);
//   * AppendEntries ack at a prepared Leader: same role epoch, no term change, no epoch move, not pending, the quorum index inside the
//     newest term run (member/Leader.java:218-237, member/Leadership.java:75-114, member/Leader.java:247-280);
//   * client append at a Leader with a non-empty log (member/Leader.java:128-140);
//   * responses to a fenced participant (transport/rpc/Async.java:157-171);
//   and, behind ONE wave-uniform branch that steady replication never takes (`election`):
//   * vote replies: counted, winning (-> Candidate / Leader), carrying a higher term (-> Follower, votedFor = responder), late for a won
//     election (Q13) (member/Candidate.java:121-134, member/Follower.java:258-270);
//   * election / heartbeat timeouts of the three roles (member/Follower.java:156-168, Candidate.java:82-88, Leader.java:120-126);
//   * RequestVote / PreVote at a Follower that has a log (member/Follower.java:91-127, 193-207);
//   * a Leader stepping down on a higher-term replication response (member/Leader.java:224-226, 178-181).
// Every conversion these rows make is one application of RaftRoutine.convertTo + RaftMember.<init> at the end (`conv_all`).
// The two rare sub-cases that need more than selects (a new term run at the log tail, prepareReplication of a new Leader) sit behind one
// more wave-level branch. Any row that misses a precondition is left untouched — `false` — and goes to Stepper::run().
// MUST be called by every lane of the wavefront (converged code).
// NOTE: bool operands are combined with & and | (never && / ||): short-circuit operators are compiled back
// into exec-mask branches, which is exactly what this tier exists to avoid.
// hdr: header with HDR_AE_OK / HDR_PEER_OK worked out by the loader; pe0: the term shared by the carried entries (if any)
template <int F, class V, class PE>
__device__ __forceinline__ bool tier1(const StepParams &p, GroupT<V> &g, PE &pe, FxT<V> &fx, bool allow, uint32_t hdr, uint32_t aux,
                                      V a, V b, V c, V d, V pe0)
{
    const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr);
    const bool flag = RG_HDR_FLAG(hdr) != 0;
    const uint32_t self = (uint32_t)p.self;
    const V g_term = g.term, g_last = g.last, g_commit = g.commit, g_epoch = g.epoch_index, lt = g.lt, top = g.top;
    const int32_t rc = g.rc, role = g.role, g_leader = g.leader, g_votes = g.votes, g_voted = g.voted_for;
    const uint32_t g_repoch = g.role_epoch;
    const bool has_log = rc > 0, g_td = g.td, g_prep = g.prepared;
    const bool peer_ok = (hdr & HDR_PEER_OK) != 0;

    // ---- AppendEntries request at a follower --------------------------------------------------
    const bool contains = c == lt;                                   // prevLogTerm == term of the tail
    const bool refresh = (a > g_term) | g_td;                        // switchTo(Follower, term, lastCandidate)
    const V ae_last = contains ? vadd<V>(b, (V)n) : g_last;
    const bool want_commit = contains & (d > g_epoch);
    const V ae_x = vmin<V>(d, ae_last);
    const bool fa = allow & ((hdr & HDR_AE_OK) != 0) & (role == RG_FOLLOWER) & (a >= g_term) &
                    (refresh | (g_leader == RG_NO_NODE) | (g_leader == (int32_t)slot)) & has_log & (b == g_last) &
                    (b > g_epoch) & !(want_commit & (ae_x < g_commit));
    const bool ae_refresh = fa & refresh;
    const bool ae_commit = fa & want_commit & (ae_x > g_commit);
    const bool ae_append = fa & contains & (n > 0);
    const bool ae_newrun = ae_append & (pe0 != lt);

    // ---- AppendEntries ack at a leader ----------------------------------------------------------
    const bool ack_kind = (kind == RG_EV_AE_ACK) | (kind == RG_EV_IS_ACK);
    const bool ack_shape = allow & (kind == RG_EV_AE_ACK) & peer_ok;
    const uint32_t j = ack_shape ? (slot < self ? slot : slot - 1u) : 0u;
    V s_epoch, s_next, s_match;
    int32_t s_rej;
    pe.load_state(j, s_epoch, s_next, s_match, s_rej);
    const bool s_pend = ((g.pending >> j) & 1u) != 0;
    const bool adv = flag & (c > s_match);
    const V n_match = adv ? c : s_match;
    const V n_next = adv ? vadd<V>(c, 1) : s_next;
    V m[F];
    pe.load_matches(m);
#pragma unroll
    for (int i = 0; i < F; i++) m[i] = ((uint32_t)i == j) ? n_match : m[i];
    V full, major;
    major_indices<F, V>(m, full, major);
    const bool lookup = flag & (major != 0);
    const bool major_ok = has_log & (major >= top) & (major <= g_last);     // inside the newest run: present, cached, its term is lt
    const V commit_to = lookup ? (lt == g_term ? major : full) : (V)0;
    const bool do_commit = (commit_to != 0) & (commit_to != g_commit);
    const bool lead_ok = (aux == g_repoch) & (role == RG_LEADER) & g_prep;
    const bool fk = ack_shape & lead_ok & (a <= g_term) &
                    (b == s_epoch) & !s_pend & (c >= s_match) & (flag | (s_match != 0)) & (n_next > b) &
                    (!lookup | major_ok) & !(do_commit & (commit_to < g_commit));
    const bool ack_commit = fk & do_commit;
    const bool ack_any = allow & ack_kind & peer_ok;
    const bool ack_drop = ack_any & (aux != g_repoch);               // AsyncHead aborted: response dropped

    // ---- client append at a leader ----------------------------------------------------------------
    const bool fc = allow & (kind == RG_EV_CLIENT_APPEND) & (role == RG_LEADER) & (n >= 1u) & has_log;
    const bool fc_newrun = fc & (lt != g_term);
    const bool fc_prepare = fc & !g_prep;

    // ---- the rows above: apply ---------------------------------------------------------------------------
    // (of these only the AppendEntries refresh converts: Follower(request term, lastCandidate) — role and vote stay what they are)
    const bool fast_main = fa | fk | fc | ack_drop;
    if (fk) pe.store_ack(j, s_epoch, n_next, n_match, flag ? 0 : (int32_t)((uint32_t)s_rej + 1u));
    if (ae_newrun | fc_newrun | fc_prepare) {                        // rare: one wave-level branch for both
        if (ae_newrun | fc_newrun) g.push(vadd<V>(g_last, 1), ae_newrun ? pe0 : g_term);
        if (fc_prepare) {                                            // Leader.prepareReplication after the FIRST new entry: nextIndex = that entry + 1
            pe.store_prepare(g_epoch, vadd<V>(g_last, 2));
            g.pending = 0;
        }
    }
    g.prepared = (g_prep & !ae_refresh) | fc_prepare;
    g.peers_dirty = g.peers_dirty | fk | fc_prepare;
    g.term = ae_refresh ? a : g_term;
    g.role_epoch = g_repoch + (ae_refresh ? 1u : 0u);
    g.td = g_td & !ae_refresh;
    g.votes = ae_refresh ? 1 : g_votes;
    g.leader = fa ? (int32_t)slot : g_leader;
    {
        const V cur_last = g.last;                                   // (a new run pushed above has already moved it)
        g.last = ae_append ? ae_last : (fc ? vadd<V>(g_last, (V)n) : cur_last);
    }
    g.log_dirty = g.log_dirty | ae_append | fc;
    g.commit = ae_commit ? ae_x : (ack_commit ? commit_to : g_commit);
    if (fast_main) {
        fx.status = ack_drop ? RG_DROPPED_STALE_ROLE : RG_OK;
        fx.resp_term = a;                                            // only read for AppendEntries (== currentTerm by now)
        fx.log_from = vadd<V>(g_last, 1);                            // only read with RG_F_LOG_APPEND
        fx.flags = (fa ? (RG_F_RESET_TIMER | RG_F_REPLIED | (contains ? RG_F_SUCCESS : 0u)) : 0u) |
                   (ae_refresh ? (RG_F_PERSIST | RG_F_ROLE_CHANGED) : 0u) |
                   ((ae_append | fc) ? RG_F_LOG_APPEND : 0u) | ((ae_commit | ack_commit) ? RG_F_COMMIT : 0u) |
                   (fc ? (RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT) : 0u);
    }

    // ---- election traffic: decided AND applied behind one wave-uniform branch ---------------------------------
    // (a lane is in at most one class, and the classes above left the state of every other lane as it was: the snapshot is still valid)
    const bool ack_down = ack_any & lead_ok & (a > g_term);          // Leader -> Follower(result.term, responder)
    const bool election = (allow & (kind - (uint32_t)RG_EV_RV_REQ <= (uint32_t)(RG_EV_TIMEOUT - RG_EV_RV_REQ))) | ack_down;
    bool el_fast = false;
    if (__builtin_amdgcn_ballot_w64(election) != 0) {               // wave-uniform: steady replication carries no such rows
        const V term1 = vadd<V>(g_term, 1), el_term = g.elected_term;
        const uint32_t el_epoch = g.elected_epoch;
        const bool cur_epoch = aux == g_repoch;
        const bool is_pv = kind == RG_EV_PV_REPLY;
        const bool is_pvq = kind == RG_EV_PV_REQ;
        // vote replies
        const bool vr_shape = allow & ((kind == RG_EV_RV_REPLY) | is_pv) & peer_ok;
        const bool sender_ok = is_pv ? ((role == RG_FOLLOWER) & g_td) : (role == RG_CANDIDATE);
        const V T = is_pv ? term1 : g_term;
        const bool vr_cur = vr_shape & cur_epoch & sender_ok & ((term1 > g_term) | !is_pv);     // (currentTerm + 1 wrapped: the general handlers)
        const bool vr_higher = vr_cur & (a > T);
        const bool vr_grant = vr_cur & (a <= T) & flag;
        const bool vr_win = vr_grant & (g_votes + 1 >= p.majority);
        const bool win_rv = vr_win & !is_pv;
        const bool vr_quiet = vr_cur & (a <= T) & !vr_win;            // counted, or refused: nothing else changes
        const bool late = vr_shape & !is_pv & !cur_epoch & (el_epoch != 0u) & (aux == el_epoch);
        const bool late_higher = late & (a > el_term);                // head.abortRequests(); Follower if that is "better"
        const bool late_noop = late & (a <= el_term) & (!flag | (el_term < g_term) | ((el_term == g_term) & (role == RG_LEADER)));
        const bool vote_drop = vr_shape & !cur_epoch & !late;
        // timeouts (aux 0 = whoever is current; context/RaftRoutine.java:70)
        const bool to_kind = allow & (kind == RG_EV_TIMEOUT) & ((aux != 0u) | (p.require_fence == 0));      // (an un-fenced row where fences are required: general handlers, RG_BAD_EVENT)
        const bool to_stale = to_kind & (aux != 0u) & (aux != g_repoch);
        const bool to_live = to_kind & !to_stale;
        const bool to_pre = to_live & (role == RG_FOLLOWER) & (p.pre_vote != 0);
        const bool to_cand = to_live & (((role == RG_FOLLOWER) & (p.pre_vote == 0)) | (role == RG_CANDIDATE)) & (term1 > g_term);
        const bool to_lead = to_live & (role == RG_LEADER);
        // RequestVote / PreVote at a Follower that has a log
        const bool vq = allow & ((kind == RG_EV_RV_REQ) | is_pvq) & (slot < (uint32_t)p.cluster) & (role == RG_FOLLOWER) & has_log;
        const bool utd = (c > lt) | ((c == lt) & (b >= g_last));      // Follower.logUpToDate with a last entry
        const bool pv_judge = vq & is_pvq & (a > g_term) & g_td;      // else failure(currentTerm), no timer touched
        const bool rv_new = vq & !is_pvq & (a > g_term);
        const bool rv_same = vq & !is_pvq & (a == g_term);
        const bool vq_success = (pv_judge & utd) | (rv_same & ((int32_t)slot == g_voted)) | (rv_new & utd);

        // RaftRoutine.convertTo + RaftMember.<init> for the lanes in `conv`
        const bool conv_self = vr_win | to_cand;                      // ballot = self
        const bool conv = vr_higher | conv_self | (late_higher & (a >= g_term)) | to_pre | rv_new | ack_down;
        const int32_t new_role = vr_win ? (is_pv ? RG_CANDIDATE : RG_LEADER) : (to_cand ? RG_CANDIDATE : RG_FOLLOWER);
        const V new_term = (to_cand | (vr_win & is_pv)) ? term1 : ((win_rv | to_pre) ? g_term : a);
        const int32_t new_vote = conv_self ? (int32_t)self : (to_pre ? g_voted : ((rv_new & !utd) ? RG_NO_NODE : (int32_t)slot));
        const bool lead_prepare = to_lead & !g_prep;                  // a new Leader's first tick: Leader.prepareReplication (member/Leader.java:30-50)
        if (lead_prepare) {
            pe.store_prepare(g_epoch, vadd<V>(has_log ? g_last : g_epoch, 1));
            g.pending = 0;
        }
        el_fast = vr_higher | vr_win | vr_quiet | late_higher | late_noop | vote_drop | to_stale | to_pre | to_cand | to_lead | vq | ack_down;
        g.elected_epoch = win_rv ? g_repoch : (late_higher ? 0u : el_epoch);      // Candidate.java:75-79 / head.abortRequests()
        g.elected_term = win_rv ? g_term : el_term;
        g.term = conv ? new_term : g.term;
        g.role = conv ? new_role : g.role;
        g.voted_for = conv ? new_vote : g.voted_for;
        g.role_epoch = g.role_epoch + (conv ? 1u : 0u);
        g.td = (g.td & !conv) | to_pre;
        g.votes = conv ? 1 : (g.votes + ((vr_grant & !vr_win) ? 1 : 0));
        g.leader = conv ? RG_NO_NODE : g.leader;
        g.prepared = (g.prepared & !conv) | lead_prepare;
        g.peers_dirty = g.peers_dirty | lead_prepare;
        if (el_fast) {
            fx.status = (vote_drop | to_stale) ? RG_DROPPED_STALE_ROLE : RG_OK;
            fx.resp_term = rv_new ? a : g_term;                       // only read for the vote requests
            fx.log_from = 0;
            fx.flags = (vq ? RG_F_REPLIED : 0u) | (vq_success ? RG_F_SUCCESS : 0u) | ((pv_judge | to_lead) ? RG_F_RESET_TIMER : 0u) |
                       (conv ? (RG_F_PERSIST | RG_F_ROLE_CHANGED | RG_F_RESET_TIMER) : 0u) |
                       (to_lead ? (RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT) : 0u) | (to_pre ? (RG_EMIT_PREVOTE << RG_F_EMIT_SHIFT) : 0u) |
                       ((conv & (new_role == RG_CANDIDATE)) ? (RG_EMIT_REQVOTE << RG_F_EMIT_SHIFT) : 0u);
        }
    }
    return fast_main | el_fast;
}

// ---- tier 1.5 (round 6), the 64-bit bodies' statement of it (the 32-bit body's is rg_tier1n.hpp: tier15, where the four classes are described) -------------
// Rows tier 1 leaves open that an election or a cache miss puts into EVERY stream — a rejected ack before the follower's first match, a vote request of a
// higher term at a Candidate, a new leader's entries over the uncommitted tail of the newest run, an AppendEntries whose prevLogIndex lies below the cached
// runs (no hint with it) — decided here, each the statements Stepper::run executes on that input, so that a wave-round whose open rows are all of these
// classes does not run the general handlers. Called for the lanes tier 1 left open (divergent is fine: no wave-level operation inside); `allow` as for tier1.
template <int F, class V, class PE>
__device__ __forceinline__ bool tier15w(const StepParams &p, GroupT<V> &g, PE &pe, FxT<V> &fx, bool allow, uint32_t hdr, uint32_t aux, bool hinted,
                                        V a, V b, V c, V d, V pe0)
{
    if (!allow) return false;
    const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr);
    const bool flag = RG_HDR_FLAG(hdr) != 0, peer_ok = (hdr & HDR_PEER_OK) != 0;
    const uint32_t self = (uint32_t)p.self;
    // a rejected ack while nothing of that follower has matched (on_replicate_ack: updateIndex's back-off)
    if ((kind == RG_EV_AE_ACK) & peer_ok & !flag & (aux == g.role_epoch) & (g.role == RG_LEADER) & g.prepared & (a <= g.term)) {
        const uint32_t j = slot < self ? slot : slot - 1u;
        V s_epoch, s_next, s_match;
        int32_t s_rej;
        pe.load_state(j, s_epoch, s_next, s_match, s_rej);
        const bool pend = ((g.pending >> j) & 1u) != 0;
        if ((s_match == 0) & (b == s_epoch) & !pend & (c >= 0)) {
            const int32_t rej = (int32_t)((uint32_t)s_rej + 1u);
            const V next = vmax<V>((V)(s_next - (V)rejection_step(rej)), vadd<V>(b, 1));
            const V nn = vmin<V>((V)(s_next - 1), next);
            pe.store_ack(j, s_epoch, nn, (V)0, rej);
            g.pending = g.pending | ((nn <= b) ? (1u << j) : 0u);
            g.peers_dirty = true;
            fx = FxT<V>{0u, RG_OK, 0, 0};
            return true;
        }
        return false;
    }
    // RequestVote / PreVote of a higher term at a Candidate (on_vote_request: switchTo(Follower, term, candidate), then the same-term answer of that Follower)
    if (((kind == RG_EV_RV_REQ) | (kind == RG_EV_PV_REQ)) & (slot < (uint32_t)p.cluster) & (slot != self) & (g.role == RG_CANDIDATE) & (a > g.term)) {
        g.role = RG_FOLLOWER; g.term = a; g.voted_for = (int32_t)slot;
        g.role_epoch += 1u;
        g.td = false; g.leader = RG_NO_NODE; g.votes = 1; g.prepared = false;
        fx = FxT<V>{RG_F_PERSIST | RG_F_ROLE_CHANGED | RG_F_RESET_TIMER | RG_F_REPLIED | RG_F_SUCCESS, RG_OK, a, 0};
        return true;
    }
    if (((hdr & HDR_AE_OK) != 0) & (g.role == RG_FOLLOWER) & !g.prepared & (g.rc > 0) & (g.epoch_index < g.last) & (a >= g.term) & (b > g.epoch_index) & (b <= g.last)) {
        // the cache miss of on_append_entries' pre-check: nothing is touched
        if (b < g.s0) {
            if (hinted) return false;
            fx = FxT<V>{0u, RG_NEED_HOST, 0, b};
            return true;
        }
        // a new leader's entries over the uncommitted tail of the newest run: truncate(prev + 1), a run of the entries' term, commit, reply
        const V lt = g.lt, commit = g.commit;
        const bool refresh = (a > g.term) | g.td;
        const V new_last = vadd<V>(b, (V)n);
        const V commit_to = vmin<V>(d, new_last);
        const bool with_commit = d > g.epoch_index;
        if (((g.leader == (int32_t)slot) | refresh) & (b < g.last) & (b >= g.top) & (c == lt) & (n >= 1u) & (pe0 != lt) & (b >= commit) & !(with_commit & (commit_to < commit))) {
            g.term = a;
            g.role_epoch += refresh ? 1u : 0u;
            g.td = false;
            g.votes = refresh ? 1 : g.votes;
            g.leader = (int32_t)slot;
            g.truncate(vadd<V>(b, 1), (V)0);
            g.push(vadd<V>(b, 1), pe0);
            g.last = new_last;
            const V new_commit = with_commit ? vmax<V>(commit, commit_to) : commit;
            g.commit = new_commit;
            fx = FxT<V>{RG_F_RESET_TIMER | RG_F_REPLIED | RG_F_SUCCESS | RG_F_LOG_TRUNC | RG_F_LOG_APPEND | (refresh ? (RG_F_PERSIST | RG_F_ROLE_CHANGED) : 0u) |
                        ((new_commit > commit) ? RG_F_COMMIT : 0u), RG_OK, a, vadd<V>(b, 1)};
            return true;
        }
    }
    return false;
}

}  // namespace rg
