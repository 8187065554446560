// rg_kernels.hip — gfx950 kernels of the batched multi-Raft decision engine.
//
// step_kernel<F, SPARSE, LANES> / step_split_kernel<F, SPARSE>: the replacement of the reference EventLoop drain
// (support/EventLoopGroup.java:32-46).  One lane = one raft group for the whole launch:
//   * group state is read ONCE (16-byte coalesced loads from the structure-of-structs table),
//     kept in VGPRs across all `rounds` of the batch, and written back once;
//   * the per-follower Leadership.State columns are staged in LDS ([follower][lane]) only for groups
//     that lead, so the runtime responder slot indexes LDS, not registers;
//   * per round every lane loads its 40-byte event as 8+16+16 B, two rounds ahead of its use (software
//     prefetch) — the event/outcome streams are what HBM sees;
//   * outcomes: the 16-byte reply is always stored; log/commit effects and the durable
//     (term, votedFor) pair are stored only for rows that have them;
//   * decision counters are per-lane tallies, reduced over the wavefront once and added to the wave's own
//     slot of a counter table at the end (no atomics).
// step_kernel: workgroup = one wavefront that does all of it (lanes never share LDS columns, no barrier anywhere).
// step_split_kernel: workgroup = a deciding and an I/O wavefront for the same 64 groups, one LDS-only barrier per
// round (see its header comment). The host picks per launch (raftgpu.cpp: step_lanes).
#include "rg_device.hpp"

namespace rg {

struct EventRow {                   // what the row index addresses: 8 + 16 + 16 bytes
    uint32_t hdr, aux;
    int64_t a, b, c, d;
};
struct EventTail {                  // what the header addresses
    int64_t hx, hy;                 // hint (only meaningful when the header carries RG_HDR_HINT_BIT)
    int64_t e0, e1, e2, e3;         // first entry terms of an AppendEntries request
};

// Stage 1 of the event pipeline: the three row-addressed loads (8 + 16 + 16 B per lane). Nothing here
// depends on loaded data, so the loads are issued two rounds ahead of their use.
__device__ __forceinline__ void load_event(const StepParams &p, size_t row, EventRow &e)
{
    const rg_ev_head_t h = p.head[row];
    const I64x2 ab = p.ab[row], cd = p.cd[row];
    e.hdr = h.hdr; e.aux = h.aux;
    e.a = ab.x; e.b = ab.y; e.c = cd.x; e.d = cd.y;
}

// Stage 2, one round later (the header has landed by now): loads whose ADDRESS comes from the header —
// the first entry terms of an AppendEntries request and the optional hint. Issued one round ahead of use.
__device__ __forceinline__ void load_event_tail(const StepParams &p, size_t row, const EventRow &e, EventTail &t)
{
    if (p.hint != nullptr && RG_HDR_HINT(e.hdr)) { const I64x2 hh = p.hint[row]; t.hx = hh.x; t.hy = hh.y; }
    // four unconditional 8-byte loads at 32-bit offsets from one uniform base: lanes without a k-th entry read the
    // word at offset 0 (always readable) instead of branching around the load. The host keeps entry_count <= 2^29,
    // so (aux + k) * 8 cannot wrap.
    const uint32_t n = RG_HDR_N(e.hdr);
    const bool have_terms = (p.entry_terms != nullptr) & (p.entry_count != 0);
    const char *base = have_terms ? reinterpret_cast<const char *>(p.entry_terms) : reinterpret_cast<const char *>(p.head);
    const bool ae = (RG_HDR_KIND(e.hdr) == RG_EV_AE_REQ) & (n > 0) & have_terms & ((uint64_t)e.aux + n <= p.entry_count);
    const uint32_t o = e.aux * 8u;
    const uint32_t o0 = ae ? o : 0u, o1 = (ae & (n > 1u)) ? o + 8u : 0u, o2 = (ae & (n > 2u)) ? o + 16u : 0u,
                   o3 = (ae & (n > 3u)) ? o + 24u : 0u;
    t.e0 = *reinterpret_cast<const int64_t *>(base + o0); t.e1 = *reinterpret_cast<const int64_t *>(base + o1);
    t.e2 = *reinterpret_cast<const int64_t *>(base + o2); t.e3 = *reinterpret_cast<const int64_t *>(base + o3);
}

// every value the 32-bit tier reads from the event is in [0, NARROW_LIMIT)
__device__ __forceinline__ bool event_narrow(int64_t a, int64_t b, int64_t c, int64_t d, int64_t e0)
{
    return ((uint64_t)a | (uint64_t)b | (uint64_t)c | (uint64_t)d | (uint64_t)e0) < NARROW_LIMIT;
}

// Group state: table -> registers (nine 16-byte coalesced loads), follower columns of a prepared leader -> LDS, and back.
__device__ __forceinline__ void load_group(const DevTable &t, uint32_t gi, Group &g)
{
    const uint32_t G = t.groups;
    const I64x2 tc = t.term_commit[gi], ep = t.epoch[gi], w = t.window[gi];
    const Ident id = t.ident[gi];
    const Elect el = t.elect[gi];
    g.term = tc.x; g.commit = tc.y; g.epoch_index = ep.x; g.epoch_term = ep.y; g.first = w.x; g.last = w.y;
    g.voted_for = id.voted_for; g.leader = id.leader; g.role_epoch = id.role_epoch;
    g.role = (int32_t)(id.meta & META_ROLE);
    g.td = (id.meta & META_TD) != 0; g.prepared = (id.meta & META_PREP) != 0;
    g.rc = (int32_t)((id.meta >> META_RC_SHIFT) & 7u);
    g.pending = (id.meta >> META_PEND_SHIFT) & 0x7Fu;
    g.elected_term = el.elected_term; g.elected_epoch = el.elected_epoch; g.votes = el.votes;
    const I64x2 r0 = t.runs[gi], r1 = t.runs[(size_t)G + gi], r2 = t.runs[(size_t)2 * G + gi], r3 = t.runs[(size_t)3 * G + gi];
    g.s0 = r0.x; g.t0 = r0.y; g.s1 = r1.x; g.t1 = r1.y; g.s2 = r2.x; g.t2 = r2.y; g.s3 = r3.x; g.t3 = r3.y;
    g.log_dirty = false; g.peers_dirty = false;
}

template <int F>
__device__ __forceinline__ void stage_peers(const DevTable &t, uint32_t gi, const Group &g, Peers<F> &pe)
{
    if (!g.prepared) return;
    const uint32_t G = t.groups;
#pragma unroll
    for (int j = 0; j < F; j++) {
        const I64x2 en = t.peer_en[(size_t)j * G + gi];
        const Match m = t.peer_m[(size_t)j * G + gi];
        pe.last_epoch[j * BLOCK] = en.x; pe.next_index[j * BLOCK] = en.y;
        pe.match_index[j * BLOCK] = m.match_index; pe.rejection[j * BLOCK] = m.rejection;
    }
}

template <int F>
__device__ __forceinline__ void store_group(const DevTable &t, uint32_t gi, uint32_t G, const Group &g, const Peers<F> &pe)
{
    t.term_commit[gi] = I64x2{g.term, g.commit};
    t.epoch[gi] = I64x2{g.epoch_index, g.epoch_term};
    t.window[gi] = I64x2{g.first, g.last};
    Ident id;
    id.voted_for = g.voted_for; id.leader = g.leader; id.role_epoch = g.role_epoch;
    id.meta = (uint32_t)g.role | (g.td ? META_TD : 0u) | (g.prepared ? META_PREP : 0u) |
              ((uint32_t)g.rc << META_RC_SHIFT) | (g.pending << META_PEND_SHIFT);
    t.ident[gi] = id;
    Elect el;
    el.elected_term = g.elected_term; el.elected_epoch = g.elected_epoch; el.votes = g.votes;
    t.elect[gi] = el;
    if (g.log_dirty) {
        t.runs[gi] = I64x2{g.s0, g.t0};
        t.runs[(size_t)G + gi] = I64x2{g.s1, g.t1};
        t.runs[(size_t)2 * G + gi] = I64x2{g.s2, g.t2};
        t.runs[(size_t)3 * G + gi] = I64x2{g.s3, g.t3};
    }
    if (g.peers_dirty) {
#pragma unroll
        for (int j = 0; j < F; j++) {
            t.peer_en[(size_t)j * G + gi] = I64x2{pe.last_epoch[j * BLOCK], pe.next_index[j * BLOCK]};
            Match m;
            m.match_index = pe.match_index[j * BLOCK]; m.rejection = pe.rejection[j * BLOCK]; m.pad = 0;
            t.peer_m[(size_t)j * G + gi] = m;
        }
    }
}

// (Fewer groups per wavefront — upper lanes masked off — was measured in round 1 at 64 / 32 / 16 / 8 lanes on 65 536 groups:
// 0.193 / 0.204 / 0.358 / 0.527 ms. A half-masked wavefront still costs both passes of a wave64 instruction; the knob is gone.)
template <int F, bool SPARSE>
__global__ __launch_bounds__(BLOCK, 1) void step_kernel(const StepParams p)
{
    __shared__ int64_t sh_epoch[F * BLOCK], sh_next[F * BLOCK], sh_match[F * BLOCK];
    __shared__ int32_t sh_rej[F * BLOCK];

    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * BLOCK + lane;
    const bool active = i < p.count;
    // Lanes past the end of the batch (tail wavefront, or the masked half of a narrow one) shadow the batch's last row:
    // they load and decide like everybody else — so the round loop has no divergent control flow around it — and only
    // their stores are switched off.
    const uint32_t ir = active ? i : p.count - 1u;
    const uint32_t gi = SPARSE ? p.gid[ir] : ir;
    const uint32_t G = p.t.groups;

    Group g;
    load_group(p.t, gi, g);
    // start the event pipeline before anything that has to wait for the state loads above
    EventRow cur{}, near{}, far{};
    EventTail cur_t{}, near_t{};
    const uint32_t last_round = p.rounds - 1u;            // the host never launches with rounds == 0
    load_event(p, ir, cur);
    load_event(p, (size_t)(last_round < 1u ? last_round : 1u) * p.count + ir, near);
    Peers<F> pe{sh_epoch + lane, sh_next + lane, sh_match + lane, sh_rej + lane};
    stage_peers<F>(p.t, gi, g, pe);

    Stepper<F> st(p, g, pe);
    st.refresh_narrow();
    const bool FAST = p.fast_paths != 0;                 // RG_FAST=0 forces every row through the general handlers (tests)
    // decision counters: per-lane 32-bit tallies (no scalar registers tied up across the loop), reduced over the
    // wavefront once at the end
    uint32_t c_rows = 0, c_replied = 0, c_conv = 0, c_commit = 0, c_assert = 0, c_need = 0, c_stale = 0, c_append = 0;
    bool blocked = false;

    // outcome of the previous round, stored one round late (see the drain below)
    rg_reply_t pend_rep{0, 0u, 0u};
    I64x2 pend_lfx{0, 0};
    rg_persist_t pend_per{0, 0, 0};
    bool pend_w_lfx = false, pend_w_per = false;

    // three-deep event pipeline: `far` = round r+2 (row loads in flight), `near` = round r+1 (header landed,
    // header-addressed loads in flight), `cur` = round r (complete). Every wait falls at the top of a
    // round, for memory operations issued a full round earlier, so their latency overlaps decision work.
    load_event_tail(p, ir, cur, cur_t);
    for (uint32_t r = 0; r < p.rounds; r++) {
        const size_t row = (size_t)r * p.count + ir;
        // Drain HERE, before issuing anything new: the vm counter retires in order and (on gfx9-class ISAs)
        // counts stores too, so (a) a wait placed lazily inside the divergent decision code would degrade to
        // vmcnt(0) and also wait for the loads issued below, and (b) draining right after the outcome stores
        // would expose the full store latency every round. Hence: loads AND the previous round's stores are
        // issued right after this point and get a whole round of decision work to complete.
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0) expcnt(7) lgkmcnt(15)
        if (r > 0) {
            if (active) p.reply[row - p.count] = pend_rep;
            if (pend_w_lfx) p.logfx[row - p.count] = pend_lfx;
            if (pend_w_per) p.persist[row - p.count] = pend_per;
        }
        // prefetch without conditions: past the last round the pipeline simply re-reads the last round's rows
        const uint32_t r1 = r + 1u < p.rounds ? r + 1u : last_round, r2 = r + 2u < p.rounds ? r + 2u : last_round;
        load_event(p, (size_t)r2 * p.count + ir, far);
        load_event_tail(p, (size_t)r1 * p.count + ir, near, near_t);

        {
            const uint32_t kind = RG_HDR_KIND(cur.hdr);
            // every lane goes through tier 1: it contains wave-uniform branches on ballots and is therefore called from converged
            // code; a lane blocked after a NEED_HOST simply asks for nothing
            const bool skip = blocked & (kind != RG_EV_NONE);
            const bool done = st.try_fast(FAST & !skip, cur.hdr, cur.aux, cur.a, cur.b, cur.c, cur.d, cur_t.e0, entries_readable(p, cur.hdr, cur.aux),
                                          entries_same_term(cur.hdr, cur_t.e0, cur_t.e1, cur_t.e2, cur_t.e3),
                                          event_narrow(cur.a, cur.b, cur.c, cur.d, cur_t.e0));
            const bool slow = !done & !skip;
            if (skip) st.fx = Fx{0u, RG_SKIPPED_AFTER_NEED_HOST, 0, 0};
            if (__builtin_amdgcn_ballot_w64(slow) != 0) {
                if (slow) st.run(cur.hdr, cur.aux, cur.a, cur.b, cur.c, cur.d, cur_t.hx, cur_t.hy, cur_t.e0, cur_t.e1, cur_t.e2, cur_t.e3);
                st.refresh_narrow();                      // the general handlers work on 64-bit values
            }
            const uint32_t status = st.fx.status, flags = st.fx.flags;
            if (status == RG_NEED_HOST) blocked = true;
            pend_rep.resp_term = (flags & RG_F_REPLIED) ? st.fx.resp_term : 0;
            pend_rep.flags = flags | ((uint32_t)g.role << RG_F_ROLE_SHIFT) | (status << RG_F_STATUS_SHIFT);
            pend_rep.role_epoch = g.role_epoch;
            pend_w_lfx = active & (((flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0) | (status == RG_NEED_HOST));
            pend_lfx = I64x2{g.commit, st.fx.log_from};
            pend_w_per = active & ((flags & RG_F_PERSIST) != 0);
            pend_per.term = g.term; pend_per.voted_for = g.voted_for; pend_per.role = g.role;

            c_rows += kind != RG_EV_NONE ? 1u : 0u;
            c_replied += (flags >> 1) & 1u;             // RG_F_REPLIED
            c_conv += (flags >> 3) & 1u;                // RG_F_ROLE_CHANGED
            c_commit += (flags >> 5) & 1u;              // RG_F_COMMIT
            c_append += (flags >> 7) & 1u;              // RG_F_LOG_APPEND
            c_assert += (status != RG_OK && status < RG_NPE_MAJOR_NULL) ? 1u : 0u;
            c_need += status == RG_NEED_HOST ? 1u : 0u;
            c_stale += status == RG_DROPPED_STALE_ROLE ? 1u : 0u;
        }
        cur = near; cur_t = near_t;
        near = far;
    }
    if (active && p.rounds > 0) {
        const size_t row = (size_t)(p.rounds - 1) * p.count + ir;
        p.reply[row] = pend_rep;
        if (pend_w_lfx) p.logfx[row] = pend_lfx;
        if (pend_w_per) p.persist[row] = pend_per;
    }

    if (active) store_group<F>(p.t, gi, G, g, pe);
    // Wavefront reduction of the tallies: butterfly over the 64 lanes, then each wave adds into its own
    // 64-byte slot of the counter table with a plain read-modify-write (8 atomics per wave onto 8 shared
    // words cost ~60 us per launch at 1024 waves). rg_counters_read sums the slots.
    uint32_t tally[RG_NUM_COUNTERS] = {c_rows, c_replied, c_conv, c_commit, c_assert, c_need, c_stale, c_append};
#pragma unroll
    for (int c = 0; c < RG_NUM_COUNTERS; c++) {
        uint32_t v = active ? tally[c] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        tally[c] = v;
    }
    if (lane < RG_NUM_COUNTERS) {
        uint32_t v = tally[0];
#pragma unroll
        for (int c = 1; c < RG_NUM_COUNTERS; c++) v = (lane == (uint32_t)c) ? tally[c] : v;
        unsigned long long *slot = p.counters + (size_t)blockIdx.x * RG_NUM_COUNTERS + lane;
        *slot += v;
    }
}

// ---- step_split_kernel: the same decisions, two instruction streams per 64 groups -----------------------------
// One wavefront per SIMD gets one issue slot every ~4 cycles and leaves half of the SIMD's VALU slots empty (DESIGN.md
// §6). This variant gives every 64 groups a workgroup of TWO wavefronts with different jobs:
//   wave 1 (I/O)     row addressing, the event loads (same three-stage prefetch as above), the outcome stores and the
//                    decision counters — everything that does not need the group's state;
//   wave 0 (decide)  group state in registers, follower state in LDS, tier 1 / tier 2 — and nothing else.
// They meet once per round at an LDS-only barrier. Events travel through a two-slot LDS ring written one round ahead,
// outcomes through a two-slot ring read one round behind, so neither wave ever waits for the other's memory traffic:
//   round r:  I/O    writes event r+1 -> ev[(r+1)&1], reads outcome r-1 <- out[(r-1)&1] and stores it, issues next loads
//             decide reads event r <- ev[r&1], decides, writes outcome r -> out[r&1]
//   barrier   (s_waitcnt lgkmcnt(0) + s_barrier: LDS traffic only — global loads/stores stay in flight across it)
enum { EV_HEAD = 0, EV_A, EV_B, EV_C, EV_D, EV_HX, EV_HY, EV_E0, EV_E1, EV_E2, EV_E3, EV_FIELDS };
enum { OUT_RESP = 0, OUT_FLAGS, OUT_COMMIT, OUT_FROM, OUT_TERM, OUT_VOTE, OUT_FIELDS };

__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0), vmcnt/expcnt untouched: my LDS writes have landed
    __builtin_amdgcn_s_barrier();
}

template <int F, bool SPARSE>
__global__ __launch_bounds__(2 * BLOCK) void step_split_kernel(const StepParams p)
{
    __shared__ int64_t sh_epoch[F * BLOCK], sh_next[F * BLOCK], sh_match[F * BLOCK];
    __shared__ int32_t sh_rej[F * BLOCK];
    __shared__ uint64_t sh_ev[2][EV_FIELDS][BLOCK];
    __shared__ uint64_t sh_out[2][OUT_FIELDS][BLOCK];

    const uint32_t lane = threadIdx.x & (BLOCK - 1);
    const bool io_wave = __builtin_amdgcn_readfirstlane(threadIdx.x) >= (uint32_t)BLOCK;       // wave-uniform
    const uint32_t i = blockIdx.x * BLOCK + lane;
    const bool active = i < p.count;
    const uint32_t ir = active ? i : p.count - 1u;       // lanes past the end shadow the last row; only their stores are off
    const uint32_t last_round = p.rounds - 1u;

    if (io_wave) {
        auto row_of = [&](uint32_t r) { return (size_t)(r < p.rounds ? r : last_round) * p.count + ir; };
        auto publish = [&](uint32_t slot, const EventRow &e, const EventTail &t) {
            const uint32_t hdr = (e.hdr & ~(HDR_SAME | HDR_ENTRIES_OK)) | (entries_readable(p, e.hdr, e.aux) ? HDR_ENTRIES_OK : 0u) |
                                 (entries_same_term(e.hdr, t.e0, t.e1, t.e2, t.e3) ? HDR_SAME : 0u);
            sh_ev[slot][EV_HEAD][lane] = (uint64_t)hdr | ((uint64_t)e.aux << 32);
            sh_ev[slot][EV_A][lane] = (uint64_t)e.a; sh_ev[slot][EV_B][lane] = (uint64_t)e.b;
            sh_ev[slot][EV_C][lane] = (uint64_t)e.c; sh_ev[slot][EV_D][lane] = (uint64_t)e.d;
            sh_ev[slot][EV_HX][lane] = (uint64_t)t.hx; sh_ev[slot][EV_HY][lane] = (uint64_t)t.hy;
            sh_ev[slot][EV_E0][lane] = (uint64_t)t.e0; sh_ev[slot][EV_E1][lane] = (uint64_t)t.e1;
            sh_ev[slot][EV_E2][lane] = (uint64_t)t.e2; sh_ev[slot][EV_E3][lane] = (uint64_t)t.e3;
        };
        uint32_t c_rows = 0, c_replied = 0, c_conv = 0, c_commit = 0, c_assert = 0, c_need = 0, c_stale = 0, c_append = 0;
        auto retire = [&](uint32_t r, uint32_t hdr) {      // outcome of round r: LDS -> global, plus the tallies
            const uint32_t slot = r & 1u;
            const size_t row = (size_t)r * p.count + ir;
            const uint64_t fe = sh_out[slot][OUT_FLAGS][lane];
            const uint32_t flags_all = (uint32_t)fe, flags = flags_all & 0xFFFFu, status = RG_F_STATUS(flags_all);
            rg_reply_t rep;
            rep.resp_term = (int64_t)sh_out[slot][OUT_RESP][lane]; rep.flags = flags_all; rep.role_epoch = (uint32_t)(fe >> 32);
            if (active) p.reply[row] = rep;
            const bool w_lfx = active & (((flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0) | (status == RG_NEED_HOST));
            if (w_lfx) p.logfx[row] = I64x2{(int64_t)sh_out[slot][OUT_COMMIT][lane], (int64_t)sh_out[slot][OUT_FROM][lane]};
            if (active & ((flags & RG_F_PERSIST) != 0)) {
                const uint64_t v = sh_out[slot][OUT_VOTE][lane];
                rg_persist_t per;
                per.term = (int64_t)sh_out[slot][OUT_TERM][lane]; per.voted_for = (int32_t)(uint32_t)v; per.role = (int32_t)(uint32_t)(v >> 32);
                p.persist[row] = per;
            }
            c_rows += RG_HDR_KIND(hdr) != RG_EV_NONE ? 1u : 0u;
            c_replied += (flags >> 1) & 1u; c_conv += (flags >> 3) & 1u; c_commit += (flags >> 5) & 1u; c_append += (flags >> 7) & 1u;
            c_assert += (status != RG_OK && status < RG_NPE_MAJOR_NULL) ? 1u : 0u;
            c_need += status == RG_NEED_HOST ? 1u : 0u;
            c_stale += status == RG_DROPPED_STALE_ROLE ? 1u : 0u;
        };

        // Five rows in flight. At the top of round r:  n1 = row r+1 and its tail (issued two rounds ago — what is published now),
        // n2 = row r+2 (its tail was issued last round), n3 = row r+3 (issued two rounds ago: its header is what this round's tail
        // loads are addressed by), n4 = row r+4 (issued last round). Every value is consumed TWO rounds after its load was issued,
        // and the vm counter retires in order, so the wait in front of publish() only covers operations older than last round's:
        // with one round of slack (the first version of this loop) the round could not be shorter than one memory round trip —
        // measured 1.4 us, i.e. 0.09 ms per 64 rounds whatever the deciding wavefront did (profiles/r02_cycle_breakdown.txt).
        EventRow n1{}, n2{}, n3{}, n4{};
        EventTail t1{}, t2{};
        uint32_t hdr_cur, hdr_prev = 0u; // headers of rounds r and r-1 (the tallies need the kind of a retired row)
        {
            EventRow first{};
            EventTail first_t{};
            load_event(p, row_of(0), first);
            load_event(p, row_of(1), n1);
            load_event(p, row_of(2), n2);
            load_event(p, row_of(3), n3);
            load_event(p, row_of(4), n4);
            load_event_tail(p, row_of(0), first, first_t);
            load_event_tail(p, row_of(1), n1, t1);
            load_event_tail(p, row_of(2), n2, t2);
            publish(0u, first, first_t);
            hdr_cur = first.hdr;
        }
        lds_barrier();                                   // event 0 is visible
        for (uint32_t r = 0; r < p.rounds; r++) {
            publish((r + 1u) & 1u, n1, t1);
            if (r > 0) retire(r - 1u, hdr_prev);
            hdr_prev = hdr_cur; hdr_cur = n1.hdr;
            n1 = n2; t1 = t2;
            n2 = n3;
            n3 = n4;
            load_event_tail(p, row_of(r + 3u), n2, t2);
            load_event(p, row_of(r + 5u), n4);
            lds_barrier();
        }
        retire(last_round, hdr_prev);

        uint32_t tally[RG_NUM_COUNTERS] = {c_rows, c_replied, c_conv, c_commit, c_assert, c_need, c_stale, c_append};
#pragma unroll
        for (int c = 0; c < RG_NUM_COUNTERS; c++) {
            uint32_t v = active ? tally[c] : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            tally[c] = v;
        }
        if (lane < RG_NUM_COUNTERS) {
            uint32_t v = tally[0];
#pragma unroll
            for (int c = 1; c < RG_NUM_COUNTERS; c++) v = (lane == (uint32_t)c) ? tally[c] : v;
            unsigned long long *slot = p.counters + (size_t)blockIdx.x * RG_NUM_COUNTERS + lane;
            *slot += v;
        }
        return;
    }

    // ---- the deciding wavefront ----------------------------------------------------------------------------------
    // This wavefront is the critical path of the workgroup and the I/O wavefronts it shares a SIMD with have ~1 000 ticks of slack per
    // round: issue priority over them. Same-box A/B at config 3: 0.1166 -> 0.1104 ms per launch (profiles/r02_cycle_breakdown.txt section 8).
    __builtin_amdgcn_s_setprio(3);
    const uint32_t gi = SPARSE ? p.gid[ir] : ir;
    const uint32_t G = p.t.groups;
    Group g;
    load_group(p.t, gi, g);
    Peers<F> pe{sh_epoch + lane, sh_next + lane, sh_match + lane, sh_rej + lane};
    stage_peers<F>(p.t, gi, g, pe);
    Stepper<F> st(p, g, pe);
    // measured (profiles/r02_*): with one deciding wavefront per SIMD the round is a dependent chain, and the 32-bit tier's
    // entry test (an LDS read, a ballot, a branch) sits at its head: 0.120 ms against 0.112 ms per launch at 65 536 groups.
    // The single-wavefront kernel (two or more deciding wavefronts per SIMD) gains from it (0.204 -> 0.198 ms at 131 072).
    st.narrow_tier = false;
    st.refresh_narrow();
    const bool FAST = p.fast_paths != 0;
    bool blocked = false;
    lds_barrier();                                       // event 0 is visible
    for (uint32_t r = 0; r < p.rounds; r++) {
        const uint32_t slot = r & 1u;
        const uint64_t head = sh_ev[slot][EV_HEAD][lane];
        const uint32_t hdr = (uint32_t)head, aux = (uint32_t)(head >> 32);
        const int64_t a = (int64_t)sh_ev[slot][EV_A][lane], b = (int64_t)sh_ev[slot][EV_B][lane],
                      c = (int64_t)sh_ev[slot][EV_C][lane], d = (int64_t)sh_ev[slot][EV_D][lane];
        const int64_t e0 = (int64_t)sh_ev[slot][EV_E0][lane];
        const bool ev_narrow = false;
        const uint32_t kind = RG_HDR_KIND(hdr);
        // tier 1 branches on wavefront ballots: every lane calls it (a lane blocked after a NEED_HOST asks for nothing)
        const bool skip = blocked & (kind != RG_EV_NONE);
        const bool done = st.try_fast(FAST & !skip, hdr, aux, a, b, c, d, e0, (hdr & HDR_ENTRIES_OK) != 0, (hdr & HDR_SAME) != 0, ev_narrow);
        const bool slow = !done & !skip;
        if (skip) st.fx = Fx{0u, RG_SKIPPED_AFTER_NEED_HOST, 0, 0};
        if (__builtin_amdgcn_ballot_w64(slow) != 0) {
            if (slow) {
                // the general handlers also want the hint and the other prefetched entry terms: read only here
                const int64_t hx = (int64_t)sh_ev[slot][EV_HX][lane], hy = (int64_t)sh_ev[slot][EV_HY][lane];
                const int64_t e1 = (int64_t)sh_ev[slot][EV_E1][lane], e2 = (int64_t)sh_ev[slot][EV_E2][lane],
                              e3 = (int64_t)sh_ev[slot][EV_E3][lane];
                st.run(hdr, aux, a, b, c, d, hx, hy, e0, e1, e2, e3);
            }
            st.refresh_narrow();                         // the general handlers work on 64-bit values
        }
        const uint32_t status = st.fx.status, flags = st.fx.flags;
        if (status == RG_NEED_HOST) blocked = true;
        const uint32_t flags_all = flags | ((uint32_t)g.role << RG_F_ROLE_SHIFT) | (status << RG_F_STATUS_SHIFT);
        sh_out[slot][OUT_RESP][lane] = (flags & RG_F_REPLIED) ? (uint64_t)st.fx.resp_term : 0ull;
        sh_out[slot][OUT_FLAGS][lane] = (uint64_t)flags_all | ((uint64_t)g.role_epoch << 32);
        sh_out[slot][OUT_COMMIT][lane] = (uint64_t)g.commit;
        sh_out[slot][OUT_FROM][lane] = (uint64_t)st.fx.log_from;
        sh_out[slot][OUT_TERM][lane] = (uint64_t)g.term;
        sh_out[slot][OUT_VOTE][lane] = (uint64_t)(uint32_t)g.voted_for | ((uint64_t)(uint32_t)g.role << 32);
        lds_barrier();
    }
    if (active) store_group<F>(p.t, gi, G, g, pe);
}

// this wavefront's earlier LDS writes are visible to its later LDS reads (other wavefronts are not involved)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}
// outputs nobody on the device reads again: write-through, do not keep them in the caches
typedef uint32_t u32x4 __attribute__((vector_size(16)));
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
template <int F>
__global__ __launch_bounds__(256) void replicate_kernel(const ReplicateParams p)
{
    __shared__ uint4 stage[4][3 * 64];                // per wavefront: 64 heads (3 x 16 B) or 64 sends (2 x 16 B)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * blockDim.x + wave * 64u;      // first row of this wavefront
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
    uint32_t pend = (id.meta >> META_PEND_SHIFT) & 0x7Fu;
    if (leader & !prepared & active) {                // Leader.prepareReplication :30-50
        const int64_t next0 = wadd(has_log ? last : ep.x, 1);
#pragma unroll
        for (int j = 0; j < F; j++) {
            p.t.peer_en[(size_t)j * G + gi] = I64x2{ep.x, next0};
            p.t.peer_m[(size_t)j * G + gi] = Match{0, 0, 0};
        }
        id.meta = (id.meta & ~(0x7Fu << META_PEND_SHIFT)) | META_PREP;
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
    uint4 *st = stage[wave];
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

__global__ __launch_bounds__(256) void timers_update_kernel(const TimerParams p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.count) return;
    const uint32_t g = p.gid ? p.gid[i] : i;
    int64_t d = p.deadline[g];
    for (uint32_t r = 0; r < p.rounds; r++) {
        const rg_reply_t rep = p.reply[(size_t)r * p.count + i];
        if (rep.flags & RG_F_RESET_TIMER)
            d = rearm(p, d, g, (int)RG_F_ROLE(rep.flags), (rep.flags & RG_F_ROLE_CHANGED) != 0, (rep.flags & RG_F_TIMER_MUTED) != 0,
                      rep.role_epoch, p.now[r]);
    }
    p.deadline[g] = d;
}

__global__ __launch_bounds__(256) void timers_arm_kernel(const TimerParams p)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.groups) return;
    if (p.deadline[g] != 0) return;
    const Ident id = p.ident[g];
    p.deadline[g] = rearm(p, 0, g, (int)(id.meta & META_ROLE), true, false, id.role_epoch, p.now[0]);
}

// Expired groups in ascending order, three passes so the list is deterministic:
//  1. per wavefront: ballot of (0 < deadline <= now), popcount -> counts[wave]
//  2. one block: exclusive prefix sum over the wavefront counts
//  3. per wavefront: every expired lane writes its gid at offset[wave] + popcount(ballot below its lane) and marks
//     the ticket fired (electionTimeout's CAS deadline -> TimerTicket.TIMEOUT)
__global__ __launch_bounds__(256) void timers_count_kernel(const int64_t *deadline, uint32_t groups, int64_t now, uint32_t *counts)
{
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

__global__ __launch_bounds__(256) void timers_emit_kernel(int64_t *deadline, const Ident *ident, uint32_t groups, int64_t now, const uint32_t *offsets,
                                                          uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity)
{
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

hipError_t launch_timers_update(const TimerParams &p, hipStream_t s)
{
    if (p.count == 0) return hipSuccess;
    hipLaunchKernelGGL(timers_update_kernel, dim3((p.count + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_timers_arm(const TimerParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(timers_arm_kernel, dim3((p.groups + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_timers_expired(int64_t *deadline, const Ident *ident, uint32_t groups, int64_t now, uint32_t *counts, uint32_t *total,
                                 uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, hipStream_t s)
{
    const uint32_t blocks = (groups + 255) / 256, waves = (groups + 63) / 64;
    hipLaunchKernelGGL(timers_count_kernel, dim3(blocks), dim3(256), 0, s, deadline, groups, now, counts);
    hipLaunchKernelGGL(timers_scan_kernel, dim3(1), dim3(1024), 0, s, counts, waves, total);
    hipLaunchKernelGGL(timers_emit_kernel, dim3(blocks), dim3(256), 0, s, deadline, ident, groups, now, counts, out_gid, out_epoch, capacity);
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
        const rg_reply_t rep = p.reply[row];
        if ((rep.flags & RG_F_ROLE_CHANGED) && RG_F_ROLE(rep.flags) == RG_LEADER) {      // new Leader: new State objects
            for (uint32_t j = 0; j < p.followers; j++) { p.ok[j * G + g] = 0; p.fail[j * G + g] = 0; p.recent[j * G + g] = 0; }
            continue;
        }
        const uint32_t hdr = p.head[row].hdr, kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), st = RG_F_STATUS(rep.flags);
        const bool ack = kind == RG_EV_AE_ACK || kind == RG_EV_IS_ACK;
        // statSuccess ran iff the callback got past the fence and the term check and the row was applied
        const bool reached = st == RG_OK || st == RG_A_MATCH_ROLLBACK || st == RG_NPE_MAJOR_NULL || st == RG_A_COMMIT_ROLLBACK;
        if (!ack || !reached || (rep.flags & RG_F_ROLE_CHANGED) || slot >= p.followers + 1 || slot == p.self) continue;
        const uint32_t j = slot < p.self ? slot : slot - 1;
        const int64_t cur = p.ok[j * G + g];
        if (p.now[r] > cur) p.ok[j * G + g] = p.now[r];                          // increaseMono
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
__global__ __launch_bounds__(256) void ready_kernel(const HealthParams p, int64_t now, int32_t critical_point, int64_t cool_down, uint8_t *ready)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.t.groups) return;
    const Ident id = p.t.ident[g];
    uint8_t out = 0;
    if ((id.meta & META_ROLE) == RG_LEADER && (id.meta & META_PREP)) {
        const uint32_t pend = (id.meta >> META_PEND_SHIFT) & 0x7Fu;
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
    ready[g] = out;
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
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (COPY_UNROLL - 1) * stride < n; i += COPY_UNROLL * stride) {
        u32x4 v[COPY_UNROLL];
#pragma unroll
        for (int k = 0; k < COPY_UNROLL; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < COPY_UNROLL; k++) __builtin_nontemporal_store(v[k], dst + i + k * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// ---- compact transfer formats of the pipelined host path (rg_submit_async_packed, include/raftgpu.h) ---------------------------
// Upload: a, b, c, d and the entry terms arrive as int32 and are widened into the 64-bit columns the step kernels read — an extra
// HBM pass of 56 B per row (~0.06 ms for 4.2 M rows) that saves 16 + 4n bytes per row on a link forty times slower than HBM.
struct alignas(16) Quad32 { int32_t a, b, c, d; };
__global__ __launch_bounds__(256) void widen_events_kernel(const Quad32 *__restrict__ q, I64x2 *__restrict__ ab, I64x2 *__restrict__ cd, uint32_t rows)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= rows) return;
    const Quad32 v = q[i];
    ab[i] = I64x2{(int64_t)v.a, (int64_t)v.b};
    cd[i] = I64x2{(int64_t)v.c, (int64_t)v.d};
}
__global__ __launch_bounds__(256) void widen_terms_kernel(const int32_t *__restrict__ src, int64_t *__restrict__ dst, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = (int64_t)src[i];
}

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

hipError_t launch_widen(const void *abcd32, I64x2 *ab, I64x2 *cd, uint32_t rows, const int32_t *terms32, int64_t *terms, uint64_t nterms, hipStream_t s)
{
    if (rows) hipLaunchKernelGGL(widen_events_kernel, dim3((rows + 255u) / 256u), dim3(256), 0, s, (const Quad32 *)abcd32, ab, cd, rows);
    if (nterms) hipLaunchKernelGGL(widen_terms_kernel, dim3((uint32_t)((nterms + 255u) / 256u)), dim3(256), 0, s, terms32, terms, nterms);
    return hipGetLastError();
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

template <int F>
static hipError_t launch_single(const StepParams &p, bool sparse, hipStream_t s)
{
    const uint32_t blocks = (p.count + BLOCK - 1) / BLOCK;
    if (blocks == 0) return hipSuccess;
    if (sparse) hipLaunchKernelGGL((step_kernel<F, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
    else        hipLaunchKernelGGL((step_kernel<F, false>), dim3(blocks), dim3(BLOCK), 0, s, p);
    return hipGetLastError();
}

template <int F>
static hipError_t launch_split(const StepParams &p, bool sparse, hipStream_t s)
{
    const uint32_t blocks = (p.count + BLOCK - 1) / BLOCK;
    if (blocks == 0) return hipSuccess;
    if (sparse) hipLaunchKernelGGL((step_split_kernel<F, true>), dim3(blocks), dim3(2 * BLOCK), 0, s, p);
    else        hipLaunchKernelGGL((step_split_kernel<F, false>), dim3(blocks), dim3(2 * BLOCK), 0, s, p);
    return hipGetLastError();
}

template <int F>
static hipError_t launch_f(const StepParams &p, bool sparse, int lanes, hipStream_t s)
{
    switch (lanes) {
    case 0:  return launch_split<F>(p, sparse, s);      // two wavefronts (decide + I/O) per 64 groups
    case 64: return launch_single<F>(p, sparse, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(const StepParams &p, int followers, bool sparse, int lanes, hipStream_t s)
{
    switch (followers) {
    case 1: return launch_f<1>(p, sparse, lanes, s);
    case 2: return launch_f<2>(p, sparse, lanes, s);
    case 3: return launch_f<3>(p, sparse, lanes, s);
    case 4: return launch_f<4>(p, sparse, lanes, s);
    case 5: return launch_f<5>(p, sparse, lanes, s);
    case 6: return launch_f<6>(p, sparse, lanes, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s, (const u32x4 *)src, (u32x4 *)dst, bytes / 16);
    return hipGetLastError();
}

}  // namespace rg
