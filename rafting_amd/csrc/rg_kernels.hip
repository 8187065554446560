// rg_kernels.hip — gfx950 kernels of the batched multi-Raft decision engine: the step kernels (rg_step.hpp: the replacement of the
// reference EventLoop drain) and, below, the send side (N1), the timers and the health / readiness gate (N4), the measured-copy
// yardstick and the compact transfer formats of the pipelined host path.
#include "rg_device.hpp"
#include "rg_step.hpp"

namespace rg {

// this wavefront's earlier LDS writes are visible to its later LDS reads (other wavefronts are not involved)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}
// outputs nobody on the device reads again: write-through, do not keep them in the caches
__device__ __forceinline__ void stream_store(uint4 *dst, const uint4 v)
{
    const u32x4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(dst));
}

// N1 — Leader.replicateLog (member/Leader.java:142-245) for many groups: one lane per row, F sends per lane.
// HBM-bound integer scan. Reads: the four 16-byte scalar columns of the row; for a Leader also the F {lastEpoch, nextIndex}
// pairs, its in-flight counts and the term runs it needs (run 0 always, runs 1-3 only when the log has more than one run).
// Writes: 48 B head + 32 F B sends per row. The outputs are arrays of structs (what the host reads), so a lane's own struct
// is not a coalesced unit; every wavefront therefore transposes through LDS and stores WHOLE 1 KiB lines, 16 B per lane:
// three store instructions for the 64 heads, two per follower for the 64 sends of that follower.
// (the rows of ONE wavefront: i0 = its first row, st = its own 3 KiB of LDS; replicate_kernel and the fused tail of a recorded tick call it)
template <int F>
__device__ __forceinline__ void replicate_wave(const ReplicateParams &p, uint4 *st, const uint32_t i0, const uint32_t lane)
{
    const uint32_t i = i0 + lane;
    const bool active = i < p.count;
    const uint32_t ir = active ? i : p.count - 1u;              // lanes past the end shadow the last row; their stores are clipped
    const uint32_t gi = p.gid ? p.gid[ir] : ir;
    const uint32_t G = p.t.groups;
    const I64x2 tc = p.t.term_commit[gi], ep = p.t.epoch[gi], w = p.t.window[gi];
    Ident id = p.t.ident[gi];
    const bool leader = (id.meta & META_ROLE) == RG_LEADER;
    const int rc = (int)((id.meta >> META_RC_SHIFT) & 7u);
    const bool has_log = rc > 0;
    const int64_t first = w.x, last = w.y;
    const bool prepared = (id.meta & META_PREP) != 0;
    // issue every load a Leader needs before anything is decided (one round trip instead of three dependent ones)
    I64x2 en[F];
    uint32_t fl[F];
#pragma unroll
    for (int j = 0; j < F; j++) {
        en[j] = I64x2{0, 0}; fl[j] = 0u;
        if (leader & prepared) en[j] = p.t.peer_en[(size_t)j * G + gi];
        if (leader && p.in_flight) fl[j] = p.in_flight[(size_t)j * p.count + ir];
    }
    I64x2 r0{0, 0}, r1{0, 0}, r2{0, 0}, r3{0, 0};
    if (leader & has_log) {
        r0 = p.t.runs[gi];
        if (rc > 1) { r1 = p.t.runs[(size_t)G + gi]; }
        if (rc > 2) { r2 = p.t.runs[(size_t)2 * G + gi]; }
        if (rc > 3) { r3 = p.t.runs[(size_t)3 * G + gi]; }
    }
    const bool hb = p.heartbeat != nullptr && p.heartbeat[ir] != 0;

    rg_send_head_t h;
    h.term = tc.x; h.leader_commit = tc.y; h.epoch_index = ep.x; h.epoch_term = ep.y;
    h.role_epoch = id.role_epoch; h.is_leader = leader ? 1u : 0u; h.reserved = 0;

    const uint32_t limit = hb ? RG_IN_FLIGHT_LIMIT / 10 : RG_IN_FLIGHT_LIMIT;
    const int64_t fetch = hb ? RG_REPLICATE_LIMIT / 2 : RG_REPLICATE_LIMIT;
    uint32_t pend = (id.meta >> META_PEND_SHIFT) & META_PEND_MASK;
    if (leader & !prepared & active) {                // Leader.prepareReplication :30-50
        const int64_t next0 = wadd(has_log ? last : ep.x, 1);
#pragma unroll
        for (int j = 0; j < F; j++) {
            p.t.peer_en[(size_t)j * G + gi] = I64x2{ep.x, next0};
            p.t.peer_m[(size_t)j * G + gi] = Match{0, 0, 0};
        }
        id.meta = (id.meta & ~(META_PEND_MASK << META_PEND_SHIFT)) | META_PREP;
        p.t.ident[gi] = id;
    }
    if (!prepared) pend = 0;
    rg_send_t sends[F];
#pragma unroll
    for (int j = 0; j < F; j++) {
        rg_send_t s{ep.x, ep.y, ep.x, 0u, RG_SEND_APPEND};
        if (!leader) {
            s = rg_send_t{0, 0, 0, 0u, RG_SEND_NONE};
        } else if (fl[j] > limit) {
            s.kind = RG_SEND_GATED;
        } else if ((pend >> j) & 1u) {
            s.kind = RG_SEND_SNAPSHOT;
        } else {
            const int64_t next_index = prepared ? en[j].y : wadd(has_log ? last : ep.x, 1);
            const int64_t next = max64(wsub(next_index, 1), ep.x);
            int64_t idx = next, len = fetch + 1;                  // RaftLog.batch(next, fetch + 1)  storage/RocksLog.java:131-166
            if (idx == ep.x) { idx = wadd(idx, 1); len -= 1; }
            int64_t avail = 0;
            if (has_log && idx >= first && idx <= last) avail = min64(len, wsub(last, idx) + 1);
            if (avail > 0) {
                if (idx == next) {                                // entries[0] is the prevLog entry
                    s.prev_index = next;
                    if (next >= r0.x) {
                        int64_t t = r0.y;
                        t = (rc > 1 && r1.x <= next) ? r1.y : t;
                        t = (rc > 2 && r2.x <= next) ? r2.y : t;
                        t = (rc > 3 && r3.x <= next) ? r3.y : t;
                        s.prev_term = t;
                    } else {
                        s.prev_term = 0;
                        s.kind = RG_SEND_NEED_HOST;
                    }
                    s.count = (uint32_t)(avail - 1);
                } else {
                    s.count = (uint32_t)avail;
                }
                s.last_index = s.count == 0 ? s.prev_index : wadd(s.prev_index, (int64_t)s.count);
            }
        }
        sends[j] = s;
    }

    // ---- transposed, line-sized stores -------------------------------------------------------------------------------
    // each wavefront owns its slice of `stage`: LDS operations of one wavefront execute in order, so a wave-level
    // scheduling barrier + lgkmcnt(0) is all the synchronisation the transpose needs (no workgroup barrier)
    const uint32_t rows_here = p.count > i0 ? (p.count - i0 < 64u ? p.count - i0 : 64u) : 0u;
    {
        const uint4 *hv = reinterpret_cast<const uint4 *>(&h);
        st[lane * 3 + 0] = hv[0]; st[lane * 3 + 1] = hv[1]; st[lane * 3 + 2] = hv[2];
        wave_lds_sync();
        uint4 *dst = reinterpret_cast<uint4 *>(p.head + i0);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t c = (uint32_t)k * 64u + lane;              // 16-byte chunk of the wavefront's 3 KiB
            if (c < rows_here * 3u) stream_store(dst + c, st[c]);
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int j = 0; j < F; j++) {
        const uint4 *sv = reinterpret_cast<const uint4 *>(&sends[j]);
        st[lane * 2 + 0] = sv[0]; st[lane * 2 + 1] = sv[1];
        wave_lds_sync();
        uint4 *dst = reinterpret_cast<uint4 *>(p.send + (size_t)j * p.count + i0);   // follower-major: element (j, row) at j * count + row
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t c = (uint32_t)k * 64u + lane;
            if (c < rows_here * 2u) stream_store(dst + c, st[c]);
        }
        wave_lds_sync();
    }
}

template <int F>
__global__ __launch_bounds__(256) void replicate_kernel(const ReplicateParams p)
{
    __shared__ uint4 stage[4][3 * 64];                // per wavefront: 64 heads (3 x 16 B) or 64 sends (2 x 16 B)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    replicate_wave<F>(p, stage[wave], blockIdx.x * blockDim.x + wave * 64u, lane);
}

hipError_t launch_replicate(const ReplicateParams &p, int followers, hipStream_t s)
{
    const uint32_t blocks = (p.count + 255) / 256;
    if (blocks == 0) return hipSuccess;
    switch (followers) {
    case 1: hipLaunchKernelGGL(replicate_kernel<1>, dim3(blocks), dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL(replicate_kernel<2>, dim3(blocks), dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL(replicate_kernel<3>, dim3(blocks), dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL(replicate_kernel<4>, dim3(blocks), dim3(256), 0, s, p); break;
    case 5: hipLaunchKernelGGL(replicate_kernel<5>, dim3(blocks), dim3(256), 0, s, p); break;
    case 6: hipLaunchKernelGGL(replicate_kernel<6>, dim3(blocks), dim3(256), 0, s, p); break;
#define RG_REPLICATE_CASE(F_) case F_: hipLaunchKernelGGL(replicate_kernel<F_>, dim3(blocks), dim3(256), 0, s, p); break;
    RG_REPLICATE_CASE(7) RG_REPLICATE_CASE(8) RG_REPLICATE_CASE(9) RG_REPLICATE_CASE(10) RG_REPLICATE_CASE(11) RG_REPLICATE_CASE(12) RG_REPLICATE_CASE(13) RG_REPLICATE_CASE(14)
#undef RG_REPLICATE_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- N4: timers ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t timer_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// RaftConfig.electionTimeout(): uniform in [E, 2E] (support/RaftConfig.java:187-190)
__host__ __device__ __forceinline__ int64_t election_timeout(uint64_t seed, uint32_t gid, uint32_t role_epoch, int64_t now, int64_t E)
{
    const uint64_t h = timer_mix(seed ^ timer_mix((uint64_t)gid * 0xD1342543DE82EF95ull ^ ((uint64_t)role_epoch << 32) ^ (uint64_t)now));
    return E + (int64_t)(h % (uint64_t)(E + 1));
}
// what RaftRoutine.resetTimer leaves behind for a participant of `role` (context/RaftRoutine.java:86-130)
__device__ __forceinline__ int64_t rearm(const TimerParams &p, int64_t d, uint32_t g, int role, bool fresh, bool muted, uint32_t role_epoch, int64_t now)
{
    if (fresh) d = 0;                                            // convertTo: ticketHolder.set(null)
    if (role == RG_LEADER) return d == 0 ? now : wadd(now, p.heartbeat_ms);   // keepAlive: schedule(exist == null ? 0 : timeout)
    if (d < 0) return d;                                         // moment < 0: the fired ticket stays
    if (muted) return INT64_MAX;                                 // resetTimer(.., true) and no un-muting call after it (:101-107)
    return wadd(now, election_timeout(p.seed, g, role_epoch, now, p.election_ms));
}

template <class P> __device__ __forceinline__ int64_t now_of(const P &p, uint32_t r) { return p.now_mem ? p.now_mem[r] : p.now[r]; }

__global__ __launch_bounds__(256) void timers_update_kernel(const TimerParams p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.count) return;
    const uint32_t g = p.gid ? p.gid[i] : i;
    int64_t d = p.deadline[g];
    uint32_t e = p.epoch[g];
    for (uint32_t r = 0; r < p.rounds; r++) {
        const rg_reply_t rep = p.reply[(size_t)r * p.count + i];
        e = rep.role_epoch;
        if (rep.flags & RG_F_RESET_TIMER)
            d = rearm(p, d, g, (int)RG_F_ROLE(rep.flags), (rep.flags & RG_F_ROLE_CHANGED) != 0, (rep.flags & RG_F_TIMER_MUTED) != 0,
                      rep.role_epoch, now_of(p, r));
    }
    p.deadline[g] = d;
    p.epoch[g] = e;
}

// The same from COMPACT outcome rows (rg_out32_t: flags in .y; rg_persist32_t: the role epoch after a conversion in .z). A row without a
// conversion keeps the epoch of the row before it; the first rows of a batch keep the epoch the group had after the previous batch: p.epoch.
__global__ __launch_bounds__(256) void timers_update32_kernel(const TimerParams p)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.count) return;
    int64_t d = p.deadline[g];
    uint32_t e = p.epoch[g];
    for (uint32_t r = 0; r < p.rounds; r++) {
        const size_t row = (size_t)r * p.count + g;
        const uint32_t flags = (uint32_t)p.out32[row].y;
        if (flags & RG_F_PERSIST) e = (uint32_t)p.persist32[row].z;
        if (flags & RG_F_RESET_TIMER)
            d = rearm(p, d, g, (int)RG_F_ROLE(flags), (flags & RG_F_ROLE_CHANGED) != 0, (flags & RG_F_TIMER_MUTED) != 0, e, now_of(p, r));
    }
    p.deadline[g] = d;
    p.epoch[g] = e;
}

__global__ __launch_bounds__(256) void timers_arm_kernel(const TimerParams p)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.groups) return;
    const Ident id = p.ident[g];
    p.epoch[g] = id.role_epoch;
    if (p.deadline[g] != 0) return;
    p.deadline[g] = rearm(p, 0, g, (int)(id.meta & META_ROLE), true, false, id.role_epoch, p.now[0]);
}

// Expired groups in ascending order, three passes so the list is deterministic:
//  1. per wavefront: ballot of (0 < deadline <= now), popcount -> counts[wave]
//  2. one block: exclusive prefix sum over the wavefront counts
//  3. per wavefront: every expired lane writes its gid at offset[wave] + popcount(ballot below its lane) and marks
//     the ticket fired (electionTimeout's CAS deadline -> TimerTicket.TIMEOUT)
__global__ __launch_bounds__(256) void timers_count_kernel(const int64_t *deadline, uint32_t groups, int64_t now, const int64_t *now_mem, uint32_t *counts)
{
    if (now_mem) now = *now_mem;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t d = g < groups ? deadline[g] : 0;
    const unsigned long long m = __ballot(d > 0 && d <= now);
    if ((threadIdx.x & 63u) == 0) counts[g >> 6] = (uint32_t)__popcll(m);
}

__global__ __launch_bounds__(1024) void timers_scan_kernel(uint32_t *counts, uint32_t waves, uint32_t *total)
{
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, per = (waves + 1023u) / 1024u;
    const uint32_t lo = tid * per, hi = lo + per < waves ? lo + per : waves;
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += counts[k];
    part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {              // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint32_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t k = lo; k < hi; k++) { const uint32_t c = counts[k]; counts[k] = run; run += c; }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void timers_emit_kernel(int64_t *deadline, const Ident *ident, uint32_t groups, int64_t now, const int64_t *now_mem,
                                                          const uint32_t *offsets, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity)
{
    if (now_mem) now = *now_mem;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t d = g < groups ? deadline[g] : 0;
    const bool exp = d > 0 && d <= now;
    const unsigned long long m = __ballot(exp);
    if (!exp) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pos = offsets[g >> 6] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (pos < capacity) {
        out_gid[pos] = g;
        if (out_epoch) out_epoch[pos] = ident[g].role_epoch;      // the participant whose ticket fired (RG_EV_TIMEOUT.aux)
        deadline[g] = -1;
    }
}

// ---- the device-resident tick, folded (rg_tick2) ---------------------------------------------------------------------------------------------
// A single-round tick over 65 536 groups is launch-bound, so what follows the decisions is folded: per group the batch's flags into the deadline
// (timers_update32_kernel) and into the followers' statistics (health_update_kernel), then the expiry: the list of the tickets that fired, in ascending group
// order — what timers_count / _scan / _emit do in three launches — by a single-pass prefix sum over the workgroups (decoupled look-back): every workgroup
// counts its fired tickets, publishes the count in its own status word, adds up the words of the workgroups before it until it meets one that already
// carries an inclusive prefix, publishes its own inclusive prefix, and writes ITS OWN groups at that offset — gid, role epoch, and the -1 in its own
// deadline column. Nothing but the status words crosses workgroups (and, on this part, XCDs — each with an L2 of its own), and they do so through agent-scope
// atomic loads and stores: written through to the device's coherence point, never served from a stale line; no workgroup executes an agent-scope release or
// acquire (those write back and invalidate the XCD's whole L2: with one per workgroup the one-kernel tick took 84 us, profiles/
// r06j_bench_tick_recordings_before_fence_fix.json), and no workgroup is left to emit the whole list alone (the last-workgroup scan that came next spent
// 10 us of a 13.6-us kernel walking masks and epochs one load after the other: profiles/r06n_tick_kernel_trace.txt).
// A status word: state (2 bits: 0 none, 1 the workgroup's own count, 2 inclusive prefix) | generation (30 bits) | value (32 bits). The generation is read
// from `ticket` and moved on by the LAST workgroup of the grid once its look-back is complete — by then every workgroup has published, hence read it — so the
// words of the previous launch are simply not of this generation and nothing is ever reset. Workgroups are dispatched in index order, so the ones a
// workgroup waits for are resident or done (the assumption every single-pass scan makes); the wait is bounded all the same: a launch that runs into the bound
// reports 0xFFFFFFFF fired tickets instead of hanging.
#ifndef RG_AGENT_LOAD               // (the host emulation runs workgroups one after the other, in index order, on plain memory)
#define RG_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RG_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
// what the batch did to the timer and the follower statistics of group g (timers_update32_kernel + health_update_kernel); returns the deadline it leaves
__device__ __forceinline__ int64_t fold_group(const TickFoldParams &p, const uint32_t g)
{
    const uint32_t G = p.tp.count;
    // RaftRoutine.resetTimer for the rows of this group (timers_update32_kernel)
    int64_t d = p.tp.deadline[g];
    uint32_t e = p.tp.epoch[g];
    const size_t GG = p.hp.t.groups;
    for (uint32_t r = 0; r < p.tp.rounds; r++) {
        const size_t row = (size_t)r * G + g;
        const uint32_t flags = (uint32_t)p.tp.out32[row].y;
        const int64_t now = p.tp.now_mem[r];
        if (flags & RG_F_PERSIST) e = (uint32_t)p.tp.persist32[row].z;
        if (flags & RG_F_RESET_TIMER)
            d = rearm(p.tp, d, g, (int)RG_F_ROLE(flags), (flags & RG_F_ROLE_CHANGED) != 0, (flags & RG_F_TIMER_MUTED) != 0, e, now);
        // Leadership.State.statSuccess for an ack that was applied (health_update_kernel)
        if ((flags & RG_F_ROLE_CHANGED) && RG_F_ROLE(flags) == RG_LEADER) {
            for (uint32_t j = 0; j < p.hp.followers; j++) { p.hp.ok[j * GG + g] = 0; p.hp.fail[j * GG + g] = 0; p.hp.recent[j * GG + g] = 0; }
            continue;
        }
        const uint32_t hdr = p.hp.head[row].hdr, kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), st = RG_F_STATUS(flags);
        const bool ack = kind == RG_EV_AE_ACK || kind == RG_EV_IS_ACK;
        const bool reached = st == RG_OK || st == RG_A_MATCH_ROLLBACK || st == RG_NPE_MAJOR_NULL || st == RG_A_COMMIT_ROLLBACK;
        if (!ack || !reached || (flags & RG_F_ROLE_CHANGED) || slot >= p.hp.followers + 1 || slot == p.hp.self) continue;
        const uint32_t j = slot < p.hp.self ? slot : slot - 1;
        if (now > p.hp.ok[j * GG + g]) p.hp.ok[j * GG + g] = now;
        p.hp.recent[j * GG + g] = 0;
    }
    p.tp.deadline[g] = d;
    p.tp.epoch[g] = e;
    return d;
}

// The fired tickets of the whole table, in ascending gid order (see above). EVERY thread of the grid calls it (the barriers); d = the deadline of the lane's
// group g, `holds` (wave-uniform) = this wavefront has groups at all (lane = group); `part` = at least 136 words of the workgroup's LDS.
__device__ __forceinline__ void expire_tail(const TickFoldParams &p, const int64_t d, const uint32_t g, const bool holds, const bool active, uint32_t *part)
{
    constexpr unsigned long long ST_COUNT = 1ull << 62, ST_PREFIX = 2ull << 62;
    constexpr uint32_t GEN_MASK = 0x3FFFFFFFu;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int64_t now = *p.now_last;
    const bool fired = holds && active && d > 0 && d <= now;
    const unsigned long long m = __ballot(fired);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t epoch = fired ? p.tp.ident[g].role_epoch : 0u;      // the participant whose ticket fired (RG_EV_TIMEOUT.aux): this workgroup's own group
    if (lane == 0) part[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < nwaves; w++) { const uint32_t c = part[w]; before += w < wave ? c : 0u; total += c; }
    __syncthreads();
    if (wave == 0) {
        uint32_t *val = part + 8, *state = part + 72;                  // one look-back window: 64 predecessors
        const uint32_t gen = RG_AGENT_LOAD(p.ticket) & GEN_MASK;
        const unsigned long long tag = (unsigned long long)gen << 32;
        uint32_t excl = 0;
        bool lost = false;
        if (blockIdx.x != 0) {
            if (lane == 0) RG_AGENT_STORE(p.masks + blockIdx.x, ST_COUNT | tag | total);
            int64_t hi = (int64_t)blockIdx.x - 1;                       // nearest predecessor not added yet
            for (;;) {
                const int64_t idx = hi - (int64_t)lane;
                unsigned long long w = ST_PREFIX | tag;                 // (before workgroup 0: an inclusive prefix of 0)
                if (idx >= 0) {
                    uint32_t spins = 0;
                    for (;;) {
                        w = RG_AGENT_LOAD(p.masks + idx);
                        if ((w >> 62) != 0ull && (uint32_t)((w >> 32) & GEN_MASK) == gen) break;
                        if (++spins > (1u << 20)) { lost = true; w = ST_PREFIX | tag; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                val[lane] = (uint32_t)w; state[lane] = (uint32_t)(w >> 62);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                const unsigned long long found = __ballot((w >> 62) == 2ull);
                const uint32_t stop = found ? (uint32_t)__builtin_ctzll(found) : 63u;      // the nearest predecessor that carries an inclusive prefix
                uint32_t add = 0;
                for (uint32_t k = 0; k <= stop; k++) add += val[k];    // (every lane walks the window: at most 64 LDS reads, no cross-lane reduction needed)
                excl += add;
                __builtin_amdgcn_wave_barrier();
                if (found) break;
                hi -= 64;
            }
        }
        lost = __ballot(lost) != 0;
        if (lane == 0) {
            RG_AGENT_STORE(p.masks + blockIdx.x, ST_PREFIX | tag | (unsigned long long)(excl + total));
            part[0] = excl;
            if (blockIdx.x == gridDim.x - 1u) {
                *p.out_count = lost ? 0xFFFFFFFFu : excl + total;
                RG_AGENT_STORE(p.ticket, (gen + 1u) & GEN_MASK);
            }
        }
    }
    __syncthreads();
    if (fired) {
        const uint32_t pos = part[0] + before + rank;
        if (pos < p.capacity) {
            p.out_gid[pos] = g;
            if (p.out_epoch) p.out_epoch[pos] = epoch;
            p.tp.deadline[g] = -1;                                    // electionTimeout's CAS: deadline -> TimerTicket.TIMEOUT
        }
    }
}

__global__ __launch_bounds__(256) void tick_fold_kernel(const TickFoldParams p)
{
    __shared__ uint32_t part[136];
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const bool active = g < p.tp.count;
    const int64_t d = active ? fold_group(p, g) : 0;
    if (p.expire) expire_tail(p, d, g, true, active, part);
}

hipError_t launch_tick_fold(const TickFoldParams &p, hipStream_t s)
{
    if (p.tp.count == 0) return hipSuccess;
    hipLaunchKernelGGL(tick_fold_kernel, dim3((p.tp.count + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_timers_update(const TimerParams &p, hipStream_t s)
{
    if (p.count == 0) return hipSuccess;
    if (p.out32) hipLaunchKernelGGL(timers_update32_kernel, dim3((p.count + 255) / 256), dim3(256), 0, s, p);
    else         hipLaunchKernelGGL(timers_update_kernel, dim3((p.count + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_timers_arm(const TimerParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(timers_arm_kernel, dim3((p.groups + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}
// now_mem non-null: the clock is read from device-visible memory when the kernels RUN (a recorded tick, rg_tick2); total: where the number of expired
// groups goes (device memory, or page-locked host memory)
hipError_t launch_timers_expired(int64_t *deadline, const Ident *ident, uint32_t groups, int64_t now, const int64_t *now_mem, uint32_t *counts, uint32_t *total,
                                 uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, hipStream_t s)
{
    const uint32_t blocks = (groups + 255) / 256, waves = (groups + 63) / 64;
    hipLaunchKernelGGL(timers_count_kernel, dim3(blocks), dim3(256), 0, s, deadline, groups, now, now_mem, counts);
    hipLaunchKernelGGL(timers_scan_kernel, dim3(1), dim3(1024), 0, s, counts, waves, total);
    hipLaunchKernelGGL(timers_emit_kernel, dim3(blocks), dim3(256), 0, s, deadline, ident, groups, now, now_mem, counts, out_gid, out_epoch, capacity);
    return hipGetLastError();
}

// ---- N4b: follower health + Leader.isReady -----------------------------------------------------------------
__global__ __launch_bounds__(256) void health_update_kernel(const HealthParams p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.count) return;
    const uint32_t g = p.gid ? p.gid[i] : i;
    const size_t G = p.t.groups;
    for (uint32_t r = 0; r < p.rounds; r++) {
        const size_t row = (size_t)r * p.count + i;
        const uint32_t flags = p.out32 ? (uint32_t)p.out32[row].y : p.reply[row].flags;      // (rg_out32_t.flags are rg_reply_t.flags)
        if ((flags & RG_F_ROLE_CHANGED) && RG_F_ROLE(flags) == RG_LEADER) {      // new Leader: new State objects
            for (uint32_t j = 0; j < p.followers; j++) { p.ok[j * G + g] = 0; p.fail[j * G + g] = 0; p.recent[j * G + g] = 0; }
            continue;
        }
        const uint32_t hdr = p.head[row].hdr, kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), st = RG_F_STATUS(flags);
        const bool ack = kind == RG_EV_AE_ACK || kind == RG_EV_IS_ACK;
        // statSuccess ran iff the callback got past the fence and the term check and the row was applied
        const bool reached = st == RG_OK || st == RG_A_MATCH_ROLLBACK || st == RG_NPE_MAJOR_NULL || st == RG_A_COMMIT_ROLLBACK;
        if (!ack || !reached || (flags & RG_F_ROLE_CHANGED) || slot >= p.followers + 1 || slot == p.self) continue;
        const uint32_t j = slot < p.self ? slot : slot - 1;
        const int64_t cur = p.ok[j * G + g], now = now_of(p, r);
        if (now > cur) p.ok[j * G + g] = now;                                    // increaseMono
        p.recent[j * G + g] = 0;
    }
}

__global__ __launch_bounds__(256) void health_failure_kernel(const HealthParams p, uint32_t n, const uint32_t *gid, const uint8_t *slot,
                                                             const uint8_t *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = gid[i], s = slot[i];
    if (g >= p.t.groups || s > p.followers || s == p.self) return;
    const uint32_t meta = p.t.ident[g].meta;
    if ((meta & META_ROLE) != RG_LEADER || !(meta & META_PREP)) return;        // no State object to land on
    const uint32_t j = s < p.self ? s : s - 1;
    const size_t G = p.t.groups, at = j * G + g;
    if (p.now[0] > p.fail[at]) p.fail[at] = p.now[0];                            // statFailure: increaseMono(requestFailure)
    if (flags[i] & 1u) atomicAdd(&p.recent[at], 1);                              // unreachable (rows may repeat a (group, follower))
    if (flags[i] & 2u) atomicAdd(&p.t.peer_m[at].rejection, 1);                  // reject
}

// Leader.isReady: ready = 1; for every State that isReady(...): ++ready > followers/2 -> true
__device__ __forceinline__ uint8_t ready_of(const HealthParams &p, const int64_t now, const int32_t critical_point, const int64_t cool_down, const uint32_t g)
{
    const Ident id = p.t.ident[g];
    uint8_t out = 0;
    if ((id.meta & META_ROLE) == RG_LEADER && (id.meta & META_PREP)) {
        const uint32_t pend = (id.meta >> META_PEND_SHIFT) & META_PEND_MASK;
        const size_t G = p.t.groups;
        uint32_t n = 1;
        for (uint32_t j = 0; j < p.followers; j++) {
            const int64_t ok = p.ok[j * G + g], fail = p.fail[j * G + g];
            const uint32_t recent = (uint32_t)p.recent[j * G + g];
            const bool unhealthy = (critical_point > 0 && recent > (uint32_t)critical_point) ||        // Integer.compareUnsigned
                                   (cool_down > 0 && wsub(now, fail) < cool_down);
            const bool is_ready = ok != 0 && !(((pend >> j) & 1u) || unhealthy);
            if (is_ready && ++n > p.followers / 2) { out = 1; break; }
        }
    }
    return out;
}
__global__ __launch_bounds__(256) void ready_kernel(const HealthParams p, int64_t now, int32_t critical_point, int64_t cool_down, uint8_t *ready)
{
    if (p.now_mem) now = *p.now_mem;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.t.groups) return;
    ready[g] = ready_of(p, now, critical_point, cool_down, g);
}

// ---- the tail of a recorded tick (rg_tick2): timers + health, the leaders' sends, isReady and the fired tickets of every group in ONE launch ------------
// A single-round tick is launch-bound (four graph nodes: 56 us for ~10 us of work at 65 536 groups), and the three steps after the decisions have the same
// shape — one lane per group, every group — with only same-group dependencies between them: isReady reads the statistics the fold wrote and the `prepared`
// mark the send side set, all of the SAME group, i.e. by the same lane, in program order. The ticket of expire_tail comes last (the last workgroup stays).
template <int F>
__global__ __launch_bounds__(256) void tick_tail_kernel(const TickTailParams p)
{
    __shared__ uint4 stage[4][3 * 64];
    __shared__ uint32_t part[136];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const bool active = g < p.fp.tp.count;
    const int64_t d = active ? fold_group(p.fp, g) : 0;
    if (p.qp.head != nullptr) replicate_wave<F>(p.qp, stage[wave], blockIdx.x * 256u + wave * 64u, lane);
    if (p.ready != nullptr && active) p.ready[g] = ready_of(p.rp, *p.rp.now_mem, p.critical_point, p.cool_down, g);
    if (p.fp.expire) expire_tail(p.fp, d, g, true, active, part);
}
// ---- a recorded tick as ONE launch: the decisions of step32_kernel (compact rows, compact outcome rows, dense) and, by the same workgroup for its own 64
// groups, everything tick_tail_kernel does. What the tail reads was written by THIS workgroup — the outcome rows by its I/O wavefront, the table by its
// deciding wavefront —, so a workgroup barrier behind `s_waitcnt vmcnt(0)` is all the ordering it needs (the wavefronts of a workgroup share their CU's L1;
// none of these lines was read before it was written); only the fired-ticket list crosses workgroups, through expire_tail's ticket as before.
template <int F, int WAVES>
__global__ __launch_bounds__(2 * BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void tick_kernel(const StepParams p, const TickTailParams tp)
{
    __shared__ alignas(16) unsigned char smem[SplitLds<F, true, 2>::BYTES];
    static_assert(SplitLds<F, true, 2>::BYTES >= 3 * 64 * 16 + 136 * 4, "the tail's staging rows and scan words reuse the step's LDS");
    if (!narrow_body<F, false, true, 1, 1>(p, smem)) {
        if (threadIdx.x == 0) { RG_NOTE_FALLBACK(); atomicAdd(p.wide_bodies, 1ull); }
        lds_barrier();
        split_body<F, false, true, true>(p, smem);
    }
    __syncthreads();                                         // every store of this workgroup has landed; its LDS is free
    uint4 *stage = reinterpret_cast<uint4 *>(smem);
    uint32_t *part = reinterpret_cast<uint32_t *>(smem + 3 * 64 * 16);
    // Both wavefronts map lane -> group as the step did. The tail is a chain of dependent memory round trips (a row's flags, then what they point at), not
    // work: the first wavefront folds the outcome rows into timers and health while the second plans the leaders' sends — neither reads what the other
    // writes —; isReady needs both (the statistics, and the `prepared` mark of a leader that sent for the first time), so it comes after a meeting, on the
    // second wavefront, while the first is already in the expiry (whose first barrier the second joins when it is done).
    const uint32_t lane = threadIdx.x & 63u;
    const bool holds = __builtin_amdgcn_readfirstlane(threadIdx.x) < (uint32_t)BLOCK;
    const uint32_t g = blockIdx.x * BLOCK + lane;
    const bool in_table = g < tp.fp.tp.count;
    const bool active = holds && in_table;
    int64_t d = 0;
    if (holds) {
        if (active) d = fold_group(tp.fp, g);
    } else {
        if (tp.qp.head != nullptr) replicate_wave<F>(tp.qp, stage, blockIdx.x * BLOCK, lane);
    }
    __syncthreads();
    if (!holds && tp.ready != nullptr && in_table) tp.ready[g] = ready_of(tp.rp, *tp.rp.now_mem, tp.critical_point, tp.cool_down, g);
    if (tp.fp.expire) expire_tail(tp.fp, d, g, holds, active, part);
}
hipError_t launch_tick(const StepParams &p, const TickTailParams &tp, int followers, hipStream_t s)
{
    const uint32_t blocks = (p.count + BLOCK - 1) / BLOCK;
    if (blocks == 0) return hipSuccess;
    if (p.count >= (1u << 28) || p.out32 == nullptr || p.force_wide != 0) return hipErrorInvalidValue;
    const bool many = blocks > 1024u;
    const dim3 grid(blocks), wg(2 * BLOCK);
    switch (followers) {
#define RG_TICK_CASE(F_) case F_: if (many) hipLaunchKernelGGL((tick_kernel<F_, 4>), grid, wg, 0, s, p, tp); else hipLaunchKernelGGL((tick_kernel<F_, 1>), grid, wg, 0, s, p, tp); break;
    RG_TICK_CASE(1) RG_TICK_CASE(2) RG_TICK_CASE(3) RG_TICK_CASE(4) RG_TICK_CASE(5) RG_TICK_CASE(6)
#undef RG_TICK_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_tick_tail(const TickTailParams &p, int followers, hipStream_t s)
{
    const uint32_t blocks = (p.fp.tp.count + 255) / 256;
    if (blocks == 0) return hipSuccess;
    switch (followers) {
#define RG_TAIL_CASE(F_) case F_: hipLaunchKernelGGL(tick_tail_kernel<F_>, dim3(blocks), dim3(256), 0, s, p); break;
    RG_TAIL_CASE(1) RG_TAIL_CASE(2) RG_TAIL_CASE(3) RG_TAIL_CASE(4) RG_TAIL_CASE(5) RG_TAIL_CASE(6)
#undef RG_TAIL_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_health_update(const HealthParams &p, hipStream_t s)
{
    if (p.count == 0) return hipSuccess;
    hipLaunchKernelGGL(health_update_kernel, dim3((p.count + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_health_failure(const HealthParams &p, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(health_failure_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n, gid, slot, flags);
    return hipGetLastError();
}
hipError_t launch_ready(const HealthParams &p, int64_t now, int32_t cp, int64_t cd, uint8_t *ready, hipStream_t s)
{
    hipLaunchKernelGGL(ready_kernel, dim3((p.t.groups + 255) / 256), dim3(256), 0, s, p, now, cp, cd, ready);
    return hipGetLastError();
}

// Plain streaming copy, the yardstick `roofline.measured_copy_gbps` is read from: four independent 16-byte non-temporal loads per lane in
// flight, then their four non-temporal stores (neither side is read again: keep the caches out of it); the grid covers the buffer once.
constexpr int COPY_UNROLL = 4;
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    // every workgroup walks its own contiguous slice, COPY_UNROLL x 4 KiB at a time, non-temporal both ways: the best of the access shapes
    // tools/membench.hip tried on an MI355X (5.49 TB/s where the grid-stride form of rounds 1-3 gave 4.3-4.9; profiles/r03a_membench.txt)
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    size_t i = lo + threadIdx.x;
    for (; i + (COPY_UNROLL - 1) * 256 < hi; i += COPY_UNROLL * 256) {
        u32x4 v[COPY_UNROLL];
#pragma unroll
        for (int k = 0; k < COPY_UNROLL; k++) v[k] = __builtin_nontemporal_load(src + i + k * 256);
#pragma unroll
        for (int k = 0; k < COPY_UNROLL; k++) __builtin_nontemporal_store(v[k], dst + i + k * 256);
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}

// ---- compact transfer formats of the pipelined host path (rg_submit_async_packed, include/raftgpu.h) ---------------------------
// Upload: a, b, c, d and the entry terms arrive as int32 and are read as such by step32_kernel (rg_step.hpp) — no widening pass.
// Download: logfx / persist rows that the reply marks as present, packed in row order (same three passes as the expired-timer
// list: per-wavefront ballot counts, one-block prefix sum, scatter). The scatter writes straight into the caller's page-locked
// buffers — the length of the lists is not known on the host when a copy would have to be queued.
__device__ __forceinline__ bool row_has_logfx(uint32_t fa)
{
    return ((fa & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0u) | (RG_F_STATUS(fa) == (uint32_t)RG_NEED_HOST);
}
__global__ __launch_bounds__(256) void outcome_count_kernel(const rg_reply_t *__restrict__ reply, uint32_t rows, uint32_t *n_logfx, uint32_t *n_persist)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t fa = i < rows ? reply[i].flags : 0u;
    const unsigned long long ml = __ballot(row_has_logfx(fa)), mp = __ballot((fa & RG_F_PERSIST) != 0u);
    if ((threadIdx.x & 63u) == 0u && i < rows) { n_logfx[i >> 6] = (uint32_t)__popcll(ml); n_persist[i >> 6] = (uint32_t)__popcll(mp); }
}
__global__ __launch_bounds__(256) void outcome_emit_kernel(const rg_reply_t *__restrict__ reply, const I64x2 *__restrict__ logfx,
                                                           const rg_persist_t *__restrict__ persist, uint32_t rows, const uint32_t *off_logfx,
                                                           const uint32_t *off_persist, I64x2 *out_logfx, uint32_t cap_logfx,
                                                           rg_persist_t *out_persist, uint32_t cap_persist)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t fa = i < rows ? reply[i].flags : 0u;
    const bool hl = row_has_logfx(fa), hp = (fa & RG_F_PERSIST) != 0u;
    const unsigned long long ml = __ballot(hl), mp = __ballot(hp);
    const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
    if (hl) {
        const uint32_t pos = off_logfx[i >> 6] + (uint32_t)__popcll(ml & below);
        if (pos < cap_logfx) out_logfx[pos] = logfx[i];
    }
    if (hp) {
        const uint32_t pos = off_persist[i >> 6] + (uint32_t)__popcll(mp & below);
        if (pos < cap_persist) out_persist[pos] = persist[i];
    }
}

// counts: [2][waves] scratch; totals: [2] (device-visible page-locked host memory)
hipError_t launch_outcome_count(const rg_reply_t *reply, uint32_t rows, uint32_t *counts, uint32_t *totals, hipStream_t s)
{
    const uint32_t blocks = (rows + 255u) / 256u, waves = (rows + 63u) / 64u;
    hipLaunchKernelGGL(outcome_count_kernel, dim3(blocks), dim3(256), 0, s, reply, rows, counts, counts + waves);
    hipLaunchKernelGGL(timers_scan_kernel, dim3(1), dim3(1024), 0, s, counts, waves, totals);
    hipLaunchKernelGGL(timers_scan_kernel, dim3(1), dim3(1024), 0, s, counts + waves, waves, totals + 1);
    return hipGetLastError();
}
hipError_t launch_outcome_emit(const rg_reply_t *reply, const I64x2 *logfx, const rg_persist_t *persist, uint32_t rows, const uint32_t *counts,
                               I64x2 *out_logfx, uint32_t cap_logfx, rg_persist_t *out_persist, uint32_t cap_persist, hipStream_t s)
{
    const uint32_t blocks = (rows + 255u) / 256u, waves = (rows + 63u) / 64u;
    hipLaunchKernelGGL(outcome_emit_kernel, dim3(blocks), dim3(256), 0, s, reply, logfx, persist, rows, counts, counts + waves, out_logfx, cap_logfx,
                       out_persist, cap_persist);
    return hipGetLastError();
}

hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s, (const u32x4 *)src, (u32x4 *)dst, bytes / 16);
    return hipGetLastError();
}

}  // namespace rg
