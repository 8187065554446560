// rg_step.hpp — the step kernels: the replacement of the reference EventLoop drain (support/EventLoopGroup.java:32-46).
// Included by rg_kernels.hip only.
//
// One lane = one raft group for the whole launch:
//   * group state is read ONCE (16-byte coalesced loads from the structure-of-structs table), kept in VGPRs across all
//     `rounds` of the batch, and written back once;
//   * the per-follower Leadership.State of groups that lead is staged in LDS, so the runtime responder slot indexes LDS,
//     not registers;
//   * per round every lane loads its event row several rounds ahead of its use (software prefetch) — the event / outcome
//     streams are what HBM sees, and they are read and written with non-temporal accesses (nobody reads them twice);
//   * outcomes: the 16-byte reply is always stored; log/commit effects and the durable (term, votedFor) pair only for rows
//     that have them;
//   * decision counters are per-lane tallies, reduced over the wavefront once and added to the workgroup's own slot of a
//     counter table at the end (no atomics).
// Three kernels share the decision code of rg_device.hpp (tier 1 + Stepper):
//   step_kernel<F, SPARSE>        wide rows (rg_batch_t), workgroup = ONE wavefront that does all of it; chosen when a launch
//                                 has more than one wavefront of groups per SIMD
//   step_split_kernel<F, SPARSE>  wide rows, workgroup = a deciding and an I/O wavefront for the same 64 groups, one LDS-only
//                                 barrier per round (split_body)
//   step32_kernel<F, SPARSE>      compact rows (rg_batch32_t), the same two wavefronts; while every value of the workgroup's 64
//                                 groups and of their rows is below 2^30 the deciding wavefront works on a 32-bit image
//                                 (narrow_body); the first value outside that domain sends THE WORKGROUP back to round 0 in
//                                 64-bit arithmetic (split_body on the compact rows) — nothing was written to the table yet and
//                                 outcome rows are simply written again, so the results are the 64-bit ones, bit for bit.
#pragma once
#include "rg_device.hpp"
#include "rg_tier1n.hpp"

namespace rg {

typedef uint32_t u32x2 __attribute__((vector_size(8)));
typedef uint32_t u32x4 __attribute__((vector_size(16)));

// event and outcome rows are touched once: keep them out of the caches
template <class T> __device__ __forceinline__ T nt_load16(const T *p)
{
    static_assert(sizeof(T) == 16, "16-byte row");
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    T r; __builtin_memcpy(&r, &v, 16); return r;
}
template <class T> __device__ __forceinline__ T nt_load8(const T *p)
{
    static_assert(sizeof(T) == 8, "8-byte row");
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p));
    T r; __builtin_memcpy(&r, &v, 8); return r;
}
template <class T> __device__ __forceinline__ void nt_store16(T *p, const T &x)
{
    static_assert(sizeof(T) == 16, "16-byte row");
    u32x4 v; __builtin_memcpy(&v, &x, 16);
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}

// Global-memory pointers made from integers (the I/O wavefront of the 32-bit body keeps its five column bases as scalars of their own, see
// there): the address space must be spelled out, or every access through them becomes a FLAT instruction — which also counts on lgkmcnt and
// would make the LDS hand-over barrier wait for the global prefetch (tests/test_kernel_static_cpu.py holds the kernels to "no FLAT").
#ifndef RG_GLOBAL_AS                // (the host emulation of tests/devemu defines these away)
#define RG_GLOBAL_AS __attribute__((address_space(1)))
#define RG_OWN_SGPRS(v) asm volatile("; %0 in scalar registers of its own" : "+s"(v))
#define RG_FRESH_VGPR(v) asm volatile("; %0 recomputed from here on" : "+v"(v))   // what is derived from v after this point is not the value derived before it
#endif
// row `index` of the T-array at `base`; with a scalar base and an index known to be below 2^28 this is global_load / global_store's own
// addressing form (saddr + 32-bit voffset): no vector address arithmetic
template <class T> __device__ __forceinline__ T nt_load_at(uint64_t base, uint32_t index)
{
    typedef uint32_t vec_t __attribute__((vector_size(sizeof(T))));
    const vec_t v = __builtin_nontemporal_load(reinterpret_cast<const RG_GLOBAL_AS vec_t *>(base) + index);
    T r; __builtin_memcpy(&r, &v, sizeof(T)); return r;
}
template <class T> __device__ __forceinline__ void nt_store_at(uint64_t base, uint32_t index, const T &x)
{
    typedef uint32_t vec_t __attribute__((vector_size(sizeof(T))));
    vec_t v; __builtin_memcpy(&v, &x, sizeof(T));
    __builtin_nontemporal_store(v, reinterpret_cast<RG_GLOBAL_AS vec_t *>(base) + index);
}

// -DRG_PROBE (experiment build, tools/probe.py): s_memtime deltas per section of a round, summed per wavefront and reported through the
// workgroup's counter slot INSTEAD of the decision counters (words 0-3: I/O wavefront, 4-7: deciding wavefront). Nothing in the default build.
#ifdef RG_PROBE
#define RG_PROBE_BEGIN() uint64_t pb_t_ = __builtin_amdgcn_s_memtime(); uint32_t pb_[4] = {0u, 0u, 0u, 0u}
#define RG_PROBE_MARK(i) do { __builtin_amdgcn_s_waitcnt(0xC07F); const uint64_t n_ = __builtin_amdgcn_s_memtime(); pb_[i] += (uint32_t)(n_ - pb_t_); pb_t_ = n_; } while (0)
#define RG_PROBE_FLUSH(base) do { if ((threadIdx.x & 63u) == 0u) for (int q_ = 0; q_ < 4; q_++) p.counters[(size_t)blockIdx.x * RG_NUM_COUNTERS + (base) + q_] += pb_[q_]; } while (0)
#else
#define RG_PROBE_BEGIN() ((void)0)
#define RG_PROBE_MARK(i) ((void)0)
#define RG_PROBE_FLUSH(base) ((void)0)
#endif

// -DRG_PROBE_HWID (experiment build, tools/placement.py): where the dispatcher put the two wavefronts of every workgroup and when each started
// and ended (HW_REG_HW_ID, HW_REG_XCC_ID, s_memrealtime), written raw into the workgroup's counter slot INSTEAD of the decision counters:
// words 0/1 = hw id | xcc id << 32 of the deciding / the I/O wavefront, 2/3 = start / end of the deciding one, 4/5 = of the I/O one.
#ifdef RG_PROBE_HWID
#define RG_HWID_BEGIN(w) do { if ((threadIdx.x & 63u) == 0u) { unsigned long long *s_ = p.counters + (size_t)blockIdx.x * RG_NUM_COUNTERS; \
    s_[w] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
    s_[2 + 2 * (w)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define RG_HWID_END(w) do { if ((threadIdx.x & 63u) == 0u) p.counters[(size_t)blockIdx.x * RG_NUM_COUNTERS + 3 + 2 * (w)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RG_HWID_BEGIN(w) ((void)0)
#define RG_HWID_END(w) ((void)0)
#endif

#ifndef RG_NOTE_SLOW                // the host emulation counts the rows and the wave-rounds that leave tier 1 (tools/tier1_coverage.py); nothing on the GPU
#define RG_NOTE_SLOW(rows, any) ((void)0)
#endif

constexpr uint32_t KIND_OUT_OF_DOMAIN = 15u;      // LDS copy of a compact row whose fields leave [0, EV_LIMIT): no event kind has this code
constexpr uint32_t HDR_SAME_IN = 1u << 9;         // compact rows, LDS / register copy: RG_HDR_SAME_TERM of the wire header (wide rows keep the hint bit here)

struct EventRow {                   // wide: what the row index addresses, 8 + 16 + 16 bytes; compact: 8 + 16 bytes widened
    uint32_t hdr, aux;
    int64_t a, b, c, d;
};
struct EventTail {                  // wide rows: what the header addresses
    int64_t hx, hy;                 // hint (only meaningful when the header carries RG_HDR_HINT_BIT)
    int64_t e0, e1, e2, e3;         // first entry terms of an AppendEntries request
};

// Stage 1 of the event pipeline: the row-addressed loads. Nothing here depends on loaded data, so the loads are issued
// rounds ahead of their use.
template <bool EV32>
__device__ __forceinline__ void load_event(const StepParams &p, size_t row, EventRow &e)
{
    const rg_ev_head_t h = nt_load8(p.head + row);
    e.hdr = h.hdr; e.aux = h.aux;
    if constexpr (EV32) {
        const I32x4 q = nt_load16(p.abcd32 + row);
        e.a = q.x; e.b = q.y; e.c = q.z; e.d = q.w;
    } else {
        const I64x2 ab = nt_load16(p.ab + row), cd = nt_load16(p.cd + row);
        e.a = ab.x; e.b = ab.y; e.c = cd.x; e.d = cd.y;
    }
}

// Stage 2 (wide rows), one round later (the header has landed by now): loads whose ADDRESS comes from the header —
// the first entry terms of an AppendEntries request and the optional hint. Compact rows have neither: the term shared by
// the entries travels in the row, anything else is read by the general handlers on demand.
template <bool EV32>
__device__ __forceinline__ void load_event_tail(const StepParams &p, size_t row, const EventRow &e, EventTail &t)
{
    if constexpr (EV32) {
        t.hx = 0; t.hy = 0;
        t.e0 = (int64_t)(int32_t)e.aux; t.e1 = t.e0; t.e2 = t.e0; t.e3 = t.e0;
    } else {
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
}

// The state-independent facts of a row, folded into header bits 10, 11 (rg_device.hpp: HDR_AE_OK, HDR_PEER_OK). Compact rows also get
// their RG_HDR_SAME_TERM bit moved to bit 9 and, when a field lies outside the 32-bit tier's domain, the kind KIND_OUT_OF_DOMAIN.
template <bool EV32>
__device__ __forceinline__ uint32_t decorate(const StepParams &p, const EventRow &e, const EventTail &t, bool mark_out_of_domain)
{
    const uint32_t hdr = e.hdr, kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr);
    const uint32_t P = (uint32_t)p.cluster, self = (uint32_t)p.self;
    const bool peer_ok = (slot < P) & (slot != self);
    bool one_term;                   // no entries, or entries that tier 1 knows to be readable and of one term
    uint32_t keep;
    if constexpr (EV32) {
        const bool same = (hdr & RG_HDR_SAME_TERM) != 0;
        one_term = (n == 0) | same;
        keep = (hdr & ~(7u << 9)) | (same ? HDR_SAME_IN : 0u);
    } else {
        const bool readable = (n <= 4u) & (p.entry_terms != nullptr) & ((uint64_t)e.aux + n <= p.entry_count);
        const bool same = ((n < 2u) | (t.e1 == t.e0)) & ((n < 3u) | (t.e2 == t.e0)) & ((n < 4u) | (t.e3 == t.e0));
        one_term = (n == 0) | (readable & same);
        keep = hdr & ~(3u << 10);
    }
    const bool ae_ok = (kind == RG_EV_AE_REQ) & (slot < P) & (e.c != 0) & (n <= RG_MAX_AE_ENTRIES) & one_term;
    uint32_t out = keep | (ae_ok ? HDR_AE_OK : 0u) | (peer_ok ? HDR_PEER_OK : 0u);
    if constexpr (EV32) {
        if (mark_out_of_domain) {
            const uint32_t w = (uint32_t)(int32_t)e.a | (uint32_t)(int32_t)e.b | (uint32_t)(int32_t)e.c | (uint32_t)(int32_t)e.d |
                               (((kind == RG_EV_AE_REQ) & ((hdr & RG_HDR_SAME_TERM) != 0)) ? e.aux : 0u);
            out = (w >= EV_LIMIT) ? (out | KIND_OUT_OF_DOMAIN) : out;
        }
    }
    return out;
}

// how the general handlers read the entry terms of this row
template <bool EV32>
__device__ __forceinline__ Entries entries_of(const StepParams &p, uint32_t hdr_in, uint32_t aux, int64_t e0, int64_t e1, int64_t e2, int64_t e3)
{
    Entries en;
    if constexpr (EV32) {
        en.t64 = nullptr;
        en.t32 = p.entry_terms32 ? p.entry_terms32 + aux : nullptr;
        en.same = ((hdr_in & HDR_SAME_IN) != 0) & (RG_HDR_KIND(hdr_in) == RG_EV_AE_REQ);
        en.prefetched = false;
        en.e0 = (int64_t)(int32_t)aux; en.e1 = en.e0; en.e2 = en.e0; en.e3 = en.e0;
    } else {
        en.t64 = p.entry_terms ? p.entry_terms + aux : nullptr;
        en.t32 = nullptr;
        en.same = false;
        en.prefetched = true;
        en.e0 = e0; en.e1 = e1; en.e2 = e2; en.e3 = e3;
    }
    return en;
}
// RG_BAD_EVENT check of an AppendEntries row that carries entries: are they readable at all
__device__ __forceinline__ bool entries_readable(const StepParams &p, const Entries &en, uint32_t aux, uint32_t n)
{
    return en.same | (((en.t64 != nullptr) | (en.t32 != nullptr)) & ((uint64_t)aux + n <= p.entry_count));
}

// Group state: table -> registers (nine 16-byte coalesced loads), follower state of a prepared leader -> LDS, and back.
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
    g.pending = (id.meta >> META_PEND_SHIFT) & META_PEND_MASK;
    g.elected_term = el.elected_term; g.elected_epoch = el.elected_epoch; g.votes = el.votes;
    const I64x2 r0 = t.runs[gi], r1 = t.runs[(size_t)G + gi], r2 = t.runs[(size_t)2 * G + gi], r3 = t.runs[(size_t)3 * G + gi];
    g.s0 = r0.x; g.t0 = r0.y; g.s1 = r1.x; g.t1 = r1.y; g.s2 = r2.x; g.t2 = r2.y; g.s3 = r3.x; g.t3 = r3.y;
    g.log_dirty = false; g.peers_dirty = false;
    g.refresh_tail();
}

template <int F>
__device__ __forceinline__ void stage_peers(const DevTable &t, uint32_t gi, const Group &g, PeersWide<F> &pe)
{
    if (!g.prepared) return;
    const uint32_t G = t.groups;
#pragma unroll
    for (int j = 0; j < F; j++) {
        const I64x2 en = t.peer_en[(size_t)j * G + gi];
        const Match m = t.peer_m[(size_t)j * G + gi];
        pe.set_last_epoch(j, en.x); pe.set_next_index(j, en.y);
        pe.set_match_index(j, m.match_index); pe.set_rejection(j, m.rejection);
    }
}
// the same into the 32-bit records, relative to the group's index base (pe.base); false when a value has no image in [0, EV_LIMIT)
template <int F>
__device__ __forceinline__ bool stage_peers(const DevTable &t, uint32_t gi, bool prepared, PeersNarrow<F> &pe)
{
    if (!prepared) return true;
    const uint32_t G = t.groups;
    const int64_t base = pe.base;
    uint64_t w = 0;
#pragma unroll
    for (int j = 0; j < F; j++) {
        const I64x2 en = t.peer_en[(size_t)j * G + gi];
        const Match m = t.peer_m[(size_t)j * G + gi];
        const int64_t ep = to_rel(en.x, base), nx = to_rel(en.y, base), mt = to_rel(m.match_index, base);
        w |= (uint64_t)ep | (uint64_t)nx | (uint64_t)mt;
        pe.rec[j * BLOCK] = I32x4{(int32_t)ep, (int32_t)nx, (int32_t)mt, m.rejection};
        *pe.mslot(j) = (int32_t)mt;
    }
    return w < (uint64_t)EV_LIMIT;
}

template <class PE>
__device__ __forceinline__ void store_group(const DevTable &t, uint32_t gi, const Group &g, const PE &pe, int followers)
{
    const uint32_t G = t.groups;
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
        for (int j = 0; j < followers; j++) {
            t.peer_en[(size_t)j * G + gi] = I64x2{pe.last_epoch(j), pe.next_index(j)};
            Match m;
            m.match_index = pe.match_index(j); m.rejection = pe.rejection(j); m.pad = 0;
            t.peer_m[(size_t)j * G + gi] = m;
        }
    }
}

// the eight decision counters of include/raftgpu.h, kept per lane by whoever sees the outcome rows
struct Tally {
    uint32_t rows = 0, replied = 0, conv = 0, commit = 0, asserts = 0, need = 0, stale = 0, append = 0;
    uint32_t packed = 0;             // add_packed(): REPLIED, ROLE_CHANGED, COMMIT, LOG_APPEND as four byte counters
    __device__ __forceinline__ void add(uint32_t kind, uint32_t flags, uint32_t status)
    {
        rows += kind != RG_EV_NONE ? 1u : 0u;
        replied += (flags >> 1) & 1u;               // RG_F_REPLIED
        conv += (flags >> 3) & 1u;                  // RG_F_ROLE_CHANGED
        commit += (flags >> 5) & 1u;                // RG_F_COMMIT
        append += (flags >> 7) & 1u;                // RG_F_LOG_APPEND
        add_status(status);
    }
    __device__ __forceinline__ void add_status(uint32_t status)
    {
        asserts += (status != RG_OK && status < RG_NPE_MAJOR_NULL) ? 1u : 0u;
        need += status == RG_NEED_HOST ? 1u : 0u;
        stale += status == RG_DROPPED_STALE_ROLE ? 1u : 0u;
    }
    // The same with the four flag counters in one register (the 32-bit body's I/O wavefront, whose instructions count): flag bits 1, 3, 5, 7
    // land in bits 0, 8, 16, 24 of one product — ((flags >> 1) & 0x55) * (1 + 2^6 + 2^12 + 2^18): bit i goes to i, i + 6, i + 12, i + 18, only
    // i + 3i reaches a multiple of 8, and no position collects more than two terms, so no carry reaches one either. Call spill() at least
    // every 255 rows.
    __device__ __forceinline__ void add_packed(uint32_t kind, uint32_t flags, uint32_t status)
    {
        rows += kind != RG_EV_NONE ? 1u : 0u;
        packed += ((((flags >> 1) & 0x55u) * 0x41041u) & 0x01010101u);
        add_status(status);
    }
    // The three status tallies the same way, the classification read from a 64-entry LDS table (status_entry) instead of being computed with five
    // compares per row: asserts | need << 8 | stale << 16.
    uint32_t packed_status = 0;
    static __device__ __forceinline__ uint32_t status_entry(uint32_t status)
    {
        return ((status != RG_OK && status < RG_NPE_MAJOR_NULL) ? 1u : 0u) | (status == RG_NEED_HOST ? 1u << 8 : 0u) | (status == RG_DROPPED_STALE_ROLE ? 1u << 16 : 0u);
    }
    __device__ __forceinline__ void add_packed_lut(uint32_t kind, uint32_t flags, uint32_t status, const uint32_t *lut)
    {
        rows += kind != RG_EV_NONE ? 1u : 0u;
        packed += ((((flags >> 1) & 0x55u) * 0x41041u) & 0x01010101u);
        packed_status += lut[status & 63u];              // (status codes end at 35)
    }
    __device__ __forceinline__ void spill()
    {
        replied += packed & 0xFFu; conv += (packed >> 8) & 0xFFu; commit += (packed >> 16) & 0xFFu; append += packed >> 24;
        asserts += packed_status & 0xFFu; need += (packed_status >> 8) & 0xFFu; stale += (packed_status >> 16) & 0xFFu;
        packed = 0u; packed_status = 0u;
    }
    // Wavefront reduction: butterfly over the 64 lanes, then the wave adds into its workgroup's own 64-byte slot of the counter
    // table with a plain read-modify-write (8 atomics per wave onto 8 shared words cost ~60 us per launch at 1024 waves).
    __device__ __forceinline__ void flush(const StepParams &p, uint32_t lane, bool active) const
    {
        uint32_t tally[RG_NUM_COUNTERS] = {rows, replied, conv, commit, asserts, need, stale, append};
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
};

// ---- step_kernel: one wavefront per 64 groups ----------------------------------------------------------------------------
template <int F, bool SPARSE>
__global__ __launch_bounds__(BLOCK, 1) void step_kernel(const StepParams p)
{
    __shared__ int64_t sh_epoch[F * BLOCK], sh_next[F * BLOCK], sh_match[F * BLOCK];
    __shared__ int32_t sh_rej[F * BLOCK];

    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * BLOCK + lane;
    const bool active = i < p.count;
    // Lanes past the end of the batch shadow the batch's last row: they load and decide like everybody else — so the round
    // loop has no divergent control flow around it — and only their stores are switched off.
    const uint32_t ir = active ? i : p.count - 1u;
    const uint32_t gi = SPARSE ? p.gid[ir] : ir;

    Group g;
    load_group(p.t, gi, g);
    // start the event pipeline before anything that has to wait for the state loads above
    EventRow cur{}, near{}, far{};
    EventTail cur_t{}, near_t{};
    const uint32_t last_round = p.rounds - 1u;            // the host never launches with rounds == 0
    load_event<false>(p, ir, cur);
    load_event<false>(p, (size_t)(last_round < 1u ? last_round : 1u) * p.count + ir, near);
    PeersWide<F> pe;
    pe.e = sh_epoch + lane; pe.n = sh_next + lane; pe.m = sh_match + lane; pe.r = sh_rej + lane; pe.overflow = false;
    stage_peers<F>(p.t, gi, g, pe);

    Stepper<F, PeersWide<F>> st(p, g, pe);
    const bool FAST = p.fast_paths != 0;                 // RG_FAST=0 forces every row through the general handlers (tests)
    Tally tally;
    bool blocked = false;

    // outcome of the previous round, stored one round late (see the drain below)
    rg_reply_t pend_rep{0, 0u, 0u};
    I64x2 pend_lfx{0, 0};
    rg_persist_t pend_per{0, 0, 0};
    bool pend_w_lfx = false, pend_w_per = false;

    // three-deep event pipeline: `far` = round r+2 (row loads in flight), `near` = round r+1 (header landed,
    // header-addressed loads in flight), `cur` = round r (complete). Every wait falls at the top of a
    // round, for memory operations issued a full round earlier, so their latency overlaps decision work.
    load_event_tail<false>(p, ir, cur, cur_t);
    for (uint32_t r = 0; r < p.rounds; r++) {
        const size_t row = (size_t)r * p.count + ir;
        // Drain HERE, before issuing anything new: the vm counter retires in order and (on gfx9-class ISAs)
        // counts stores too, so (a) a wait placed lazily inside the divergent decision code would degrade to
        // vmcnt(0) and also wait for the loads issued below, and (b) draining right after the outcome stores
        // would expose the full store latency every round. Hence: loads AND the previous round's stores are
        // issued right after this point and get a whole round of decision work to complete.
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0) expcnt(7) lgkmcnt(15)
        if (r > 0) {
            if (active) nt_store16(p.reply + (row - p.count), pend_rep);
            if (pend_w_lfx) nt_store16(p.logfx + (row - p.count), pend_lfx);
            if (pend_w_per) nt_store16(p.persist + (row - p.count), pend_per);
        }
        // prefetch without conditions: past the last round the pipeline simply re-reads the last round's rows
        const uint32_t r1 = r + 1u < p.rounds ? r + 1u : last_round, r2 = r + 2u < p.rounds ? r + 2u : last_round;
        load_event<false>(p, (size_t)r2 * p.count + ir, far);
        load_event_tail<false>(p, (size_t)r1 * p.count + ir, near, near_t);

        {
            const uint32_t kind = RG_HDR_KIND(cur.hdr);
            // every lane goes through tier 1: it contains wave-uniform branches on ballots and is therefore called from converged
            // code; a lane blocked after a NEED_HOST simply asks for nothing
            const bool skip = blocked & (kind != RG_EV_NONE);
            const uint32_t hdr = decorate<false>(p, cur, cur_t, false);
            const bool done = tier1<F, int64_t, PeersWide<F>>(p, g, pe, st.fx, FAST & !skip, hdr, cur.aux, cur.a, cur.b, cur.c, cur.d, cur_t.e0);
            bool slow = !done & !skip;
            if (skip) st.fx = Fx{0u, RG_SKIPPED_AFTER_NEED_HOST, 0, 0};
            if (__builtin_amdgcn_ballot_w64(slow) != 0) {
                if (slow) slow = !tier15w<F, int64_t, PeersWide<F>>(p, g, pe, st.fx, FAST, hdr, cur.aux, (RG_HDR_HINT(cur.hdr) != 0) & (p.hint != nullptr), cur.a, cur.b, cur.c, cur.d, cur_t.e0);
                if (slow) {
                    const Entries en = entries_of<false>(p, cur.hdr, cur.aux, cur_t.e0, cur_t.e1, cur_t.e2, cur_t.e3);
                    st.run(cur.hdr, cur.aux, cur.a, cur.b, cur.c, cur.d, (RG_HDR_HINT(cur.hdr) != 0) & (p.hint != nullptr), cur_t.hx, cur_t.hy, en,
                           entries_readable(p, en, cur.aux, RG_HDR_N(cur.hdr)));
                }
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
            tally.add(kind, flags, status);
        }
        cur = near; cur_t = near_t;
        near = far;
    }
    if (active) {
        const size_t row = (size_t)(p.rounds - 1) * p.count + ir;
        nt_store16(p.reply + row, pend_rep);
        if (pend_w_lfx) nt_store16(p.logfx + row, pend_lfx);
        if (pend_w_per) nt_store16(p.persist + row, pend_per);
    }

    if (active) store_group(p.t, gi, g, pe, F);
    tally.flush(p, lane, active);
}

// ---- two instruction streams per 64 groups ------------------------------------------------------------------------------------
// One wavefront per SIMD gets one issue slot every ~4 cycles and leaves half of the SIMD's VALU slots empty (DESIGN.md
// §6). These bodies give every 64 groups a workgroup of TWO wavefronts with different jobs:
//   wave 1 (I/O)     row addressing, the event loads, the state-independent facts of a row (decorate), the outcome stores and the
//                    decision counters — everything that does not need the group's state;
//   wave 0 (decide)  group state in registers, follower state in LDS, tier 1 / tier 2 — and nothing else; it runs at raised issue
//                    priority (s_setprio 3): it is the critical path, its I/O neighbours on the SIMD have slack
//                    (same-box A/B at config 3: 0.1167 -> 0.1102 ms per launch, profiles/r03a_prio_ab.jsonl).
// They meet once per round at an LDS-only barrier. Events travel through a two-slot LDS ring written one round ahead,
// outcomes through a two-slot ring read one round behind, so neither wave ever waits for the other's memory traffic:
//   round r:  I/O    writes event r+1 -> ev[(r+1)&1], reads outcome r-1 <- out[(r-1)&1] and stores it, issues next loads
//             decide reads event r <- ev[r&1], decides, writes outcome r -> out[r&1]
//   barrier   (s_waitcnt lgkmcnt(0) + s_barrier: LDS traffic only — global loads/stores stay in flight across it)
enum { EV_HEAD = 0, EV_A, EV_B, EV_C, EV_D, EV_E0, EV_HX, EV_HY, EV_E1, EV_E2, EV_E3, EV_FIELDS };
enum { OUT_RESP = 0, OUT_FLAGS, OUT_COMMIT, OUT_FROM, OUT_TERM, OUT_VOTE, OUT_FIELDS };

__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0), vmcnt/expcnt untouched: my LDS writes have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");              // (compiler only: no LDS access of this wavefront moves across the hand-over)
}

// LDS of a two-wavefront workgroup. The 64-bit body and the 32-bit body never run at the same time: one buffer, two layouts. On compact rows
// (EV32) the 64-bit body's event ring carries five fields instead of eleven (no hint, the entry term is `aux`): 18 KB per workgroup at F = 4 (19 KB with the 32-bit body's tables),
// eight workgroups per CU instead of six.
template <int F, bool EV32, int NO = 2>
struct SplitLds {
    // 64-bit body
    static constexpr int EVF = EV32 ? (int)EV_D + 1 : (int)EV_FIELDS;
    static constexpr size_t W_EPOCH = 0, W_NEXT = W_EPOCH + F * BLOCK * 8, W_MATCH = W_NEXT + F * BLOCK * 8, W_REJ = W_MATCH + F * BLOCK * 8,
                            W_EV = W_REJ + F * BLOCK * 4, W_OUT = W_EV + 2 * EVF * BLOCK * 8, W_END = W_OUT + 2 * OUT_FIELDS * BLOCK * 8;
    // 32-bit body: follower records, their matchIndex row, a four-slot event ring (written two rounds ahead), a two-slot outcome ring, the I/O
    // wavefront's tables: class word by (kind, slot), flags by predicate word (19 KB in all: eight workgroups per CU still fit)
    static constexpr int MV = (F + 3) / 4;
    static constexpr int NEV = 4;
    static constexpr size_t N_REC = 0, N_MV = N_REC + F * BLOCK * 16, N_EVH = N_MV + MV * BLOCK * 16, N_EVQ = N_EVH + NEV * BLOCK * 16,
                            N_O0 = N_EVQ + NEV * BLOCK * 16, N_O1 = N_O0 + NO * BLOCK * 16, N_BAIL = N_O1 + NO * BLOCK * 16,      // (NO outcome slots: 2, or 4 where two rounds share a hand-over)
                            N_LUTC = N_BAIL + 16, N_LUTM = N_LUTC + 256 * 4, N_LUTE = N_LUTM + 128 * 4,      // the I/O wavefront's tables (below)
                            N_BASE = N_LUTE + 128 * 2, N_LUTS = N_BASE + BLOCK * 8,                           // the groups' index bases (the deciding wavefront's; round 5)
                            N_END = N_LUTS + 64 * 4;                                                          // the I/O wavefront's status -> tally table
    static constexpr size_t BYTES = EV32 ? (W_END > N_END ? W_END : N_END) : W_END;
};

// The 64-bit body of a two-wavefront workgroup, on wide (EV32 = false) or compact (EV32 = true) rows.
// OUT32 (compact rows only): the outcome goes out as rg_out32_t / rg_persist32_t rows (rg_submit32c); an event whose values do not fit int32 is
// flagged RG_F_WIDE_VALUES and its full rows go to the overflow columns p.reply / p.logfx / p.persist, when the caller gave them.
template <int F, bool SPARSE, bool EV32, bool OUT32 = false>
__device__ __forceinline__ void split_body(const StepParams &p, unsigned char *smem)
{
    static_assert(EV32 || !OUT32, "compact outcome rows belong to the compact-row kernels");
    typedef SplitLds<F, EV32> L;
    int64_t *sh_epoch = reinterpret_cast<int64_t *>(smem + L::W_EPOCH), *sh_next = reinterpret_cast<int64_t *>(smem + L::W_NEXT),
            *sh_match = reinterpret_cast<int64_t *>(smem + L::W_MATCH);
    int32_t *sh_rej = reinterpret_cast<int32_t *>(smem + L::W_REJ);
    uint64_t (*sh_ev)[L::EVF][BLOCK] = reinterpret_cast<uint64_t (*)[L::EVF][BLOCK]>(smem + L::W_EV);
    uint64_t (*sh_out)[OUT_FIELDS][BLOCK] = reinterpret_cast<uint64_t (*)[OUT_FIELDS][BLOCK]>(smem + L::W_OUT);

    const uint32_t lane = threadIdx.x & (BLOCK - 1);
    const bool io_wave = __builtin_amdgcn_readfirstlane(threadIdx.x) >= (uint32_t)BLOCK;       // wave-uniform
    const uint32_t i = blockIdx.x * BLOCK + lane;
    const bool active = i < p.count;
    const uint32_t ir = active ? i : p.count - 1u;       // lanes past the end shadow the last row; only their stores are off
    const uint32_t last_round = p.rounds - 1u;

    if (io_wave) {
        auto row_of = [&](uint32_t r) { return (size_t)(r < p.rounds ? r : last_round) * p.count + ir; };
        // compact rows (and compact outcome rows) carry log indices relative to the group's index base (rg_device.hpp: to_rel): this body decides on
        // absolute values, so the row's index fields are taken off the base as they are handed over, and put back on it in the rows of OUT32
        int64_t io_base = 0;
        if constexpr (EV32) io_base = (p.has_bases != 0) ? p.t.ibase[SPARSE ? p.gid[ir] : ir] : 0;
        // (round 6: a wavefront whose 64 bases are all zero — every table that never had one — skips the conversions at both ends behind ONE wave-uniform
        //  flag, as the 32-bit body does: they were part of what this body lost in round 5, 0.1006 -> 0.1140 ms at config 3)
        const bool io_rel = EV32 && __builtin_amdgcn_ballot_w64(io_base != 0) != 0;
        auto publish = [&](uint32_t slot, const EventRow &e, const EventTail &t) {
            sh_ev[slot][EV_HEAD][lane] = (uint64_t)decorate<EV32>(p, e, t, false) | ((uint64_t)e.aux << 32);
            if (io_rel) {
                const uint32_t ix = index_fields(RG_HDR_KIND(e.hdr));
                sh_ev[slot][EV_A][lane] = (uint64_t)((ix & 1u) ? to_abs(e.a, io_base) : e.a); sh_ev[slot][EV_B][lane] = (uint64_t)((ix & 2u) ? to_abs(e.b, io_base) : e.b);
                sh_ev[slot][EV_C][lane] = (uint64_t)((ix & 4u) ? to_abs(e.c, io_base) : e.c); sh_ev[slot][EV_D][lane] = (uint64_t)((ix & 8u) ? to_abs(e.d, io_base) : e.d);
            } else {
                sh_ev[slot][EV_A][lane] = (uint64_t)e.a; sh_ev[slot][EV_B][lane] = (uint64_t)e.b; sh_ev[slot][EV_C][lane] = (uint64_t)e.c; sh_ev[slot][EV_D][lane] = (uint64_t)e.d;
            }
            if constexpr (!EV32) {
                sh_ev[slot][EV_E0][lane] = (uint64_t)t.e0;
                sh_ev[slot][EV_HX][lane] = (uint64_t)t.hx; sh_ev[slot][EV_HY][lane] = (uint64_t)t.hy;
                sh_ev[slot][EV_E1][lane] = (uint64_t)t.e1;
                sh_ev[slot][EV_E2][lane] = (uint64_t)t.e2; sh_ev[slot][EV_E3][lane] = (uint64_t)t.e3;
            }
        };
        Tally tally;
        auto retire = [&](uint32_t r, uint32_t hdr) {      // outcome of round r: LDS -> global, plus the tallies
            const uint32_t slot = r & 1u;
            const size_t row = (size_t)r * p.count + ir;
            const uint64_t fe = sh_out[slot][OUT_FLAGS][lane];
            const uint32_t flags_all = (uint32_t)fe, flags = flags_all & 0xFFFFu, status = RG_F_STATUS(flags_all);
            rg_reply_t rep;
            rep.resp_term = (int64_t)sh_out[slot][OUT_RESP][lane]; rep.flags = flags_all; rep.role_epoch = (uint32_t)(fe >> 32);
            const bool w_lfx = active & (((flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0) | (status == RG_NEED_HOST));
            const bool w_per = active & ((flags & RG_F_PERSIST) != 0);
            const I64x2 lfx{(int64_t)sh_out[slot][OUT_COMMIT][lane], (int64_t)sh_out[slot][OUT_FROM][lane]};
            const uint64_t v = sh_out[slot][OUT_VOTE][lane];
            rg_persist_t per;
            per.term = (int64_t)sh_out[slot][OUT_TERM][lane]; per.voted_for = (int32_t)(uint32_t)v; per.role = (int32_t)(uint32_t)(v >> 32);
            if constexpr (OUT32) {
                const int64_t commit_r = io_rel ? to_rel(lfx.x, io_base) : lfx.x, from_r = io_rel ? to_rel(lfx.y, io_base) : lfx.y;
                const uint64_t bits = (uint64_t)rep.resp_term | (uint64_t)commit_r | (w_lfx ? (uint64_t)from_r : 0ull) | (w_per ? (uint64_t)per.term : 0ull);
                const bool wide = bits >= (1ull << 31);
                if (active) {
                    nt_store16(p.out32 + row, I32x4{(int32_t)rep.resp_term, (int32_t)(flags_all | (wide ? RG_F_WIDE_VALUES : 0u)), (int32_t)commit_r, (int32_t)from_r});
                    if (w_per) nt_store16(p.persist32 + row, I32x4{(int32_t)per.term, per.voted_for, (int32_t)rep.role_epoch, per.role});
                    if (wide & (p.reply != nullptr)) {
                        nt_store16(p.reply + row, rep);
                        if (w_lfx) nt_store16(p.logfx + row, lfx);
                        if (w_per) nt_store16(p.persist + row, per);
                    }
                }
            } else {
                if (active) nt_store16(p.reply + row, rep);
                if (w_lfx) nt_store16(p.logfx + row, lfx);
                if (w_per) nt_store16(p.persist + row, per);
            }
            tally.add(RG_HDR_KIND(hdr), flags, status);
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
            load_event<EV32>(p, row_of(0), first);
            load_event<EV32>(p, row_of(1), n1);
            load_event<EV32>(p, row_of(2), n2);
            load_event<EV32>(p, row_of(3), n3);
            load_event<EV32>(p, row_of(4), n4);
            load_event_tail<EV32>(p, row_of(0), first, first_t);
            load_event_tail<EV32>(p, row_of(1), n1, t1);
            load_event_tail<EV32>(p, row_of(2), n2, t2);
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
            load_event_tail<EV32>(p, row_of(r + 3u), n2, t2);
            load_event<EV32>(p, row_of(r + 5u), n4);
            lds_barrier();
        }
        retire(last_round, hdr_prev);
        tally.flush(p, lane, active);
        return;
    }

    // ---- the deciding wavefront ----------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const uint32_t gi = SPARSE ? p.gid[ir] : ir;
    Group g;
    load_group(p.t, gi, g);
    PeersWide<F> pe;
    pe.e = sh_epoch + lane; pe.n = sh_next + lane; pe.m = sh_match + lane; pe.r = sh_rej + lane; pe.overflow = false;
    stage_peers<F>(p.t, gi, g, pe);
    Stepper<F, PeersWide<F>> st(p, g, pe);
    const bool FAST = p.fast_paths != 0;
    bool blocked = false;
    lds_barrier();                                       // event 0 is visible
    for (uint32_t r = 0; r < p.rounds; r++) {
        const uint32_t slot = r & 1u;
        const uint64_t head = sh_ev[slot][EV_HEAD][lane];
        const uint32_t hdr = (uint32_t)head, aux = (uint32_t)(head >> 32);
        const int64_t a = (int64_t)sh_ev[slot][EV_A][lane], b = (int64_t)sh_ev[slot][EV_B][lane],
                      c = (int64_t)sh_ev[slot][EV_C][lane], d = (int64_t)sh_ev[slot][EV_D][lane];
        int64_t e0 = (int64_t)(int32_t)aux;              // compact rows: the term shared by the carried entries travels in aux
        if constexpr (!EV32) e0 = (int64_t)sh_ev[slot][EV_E0][lane];
        const uint32_t kind = RG_HDR_KIND(hdr);
        // tier 1 branches on wavefront ballots: every lane calls it (a lane blocked after a NEED_HOST asks for nothing)
        const bool skip = blocked & (kind != RG_EV_NONE);
        const bool done = tier1<F, int64_t, PeersWide<F>>(p, g, pe, st.fx, FAST & !skip, hdr, aux, a, b, c, d, e0);
        bool slow = !done & !skip;
        if (skip) st.fx = Fx{0u, RG_SKIPPED_AFTER_NEED_HOST, 0, 0};
        if (__builtin_amdgcn_ballot_w64(slow) != 0) {
            if (slow) slow = !tier15w<F, int64_t, PeersWide<F>>(p, g, pe, st.fx, FAST, hdr, aux, EV32 ? false : ((RG_HDR_HINT(hdr) != 0) & (p.hint != nullptr)), a, b, c, d, e0);
            if (slow) {
                // the general handlers also want the hint and the other prefetched entry terms: read only here
                int64_t hx = 0, hy = 0, e1 = e0, e2 = e0, e3 = e0;
                if constexpr (!EV32) {
                    hx = (int64_t)sh_ev[slot][EV_HX][lane]; hy = (int64_t)sh_ev[slot][EV_HY][lane];
                    e1 = (int64_t)sh_ev[slot][EV_E1][lane]; e2 = (int64_t)sh_ev[slot][EV_E2][lane]; e3 = (int64_t)sh_ev[slot][EV_E3][lane];
                }
                const Entries en = entries_of<EV32>(p, hdr, aux, e0, e1, e2, e3);
                const bool hinted = EV32 ? false : ((RG_HDR_HINT(hdr) != 0) & (p.hint != nullptr));
                st.run(hdr, aux, a, b, c, d, hinted, hx, hy, en, entries_readable(p, en, aux, RG_HDR_N(hdr)));
            }
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
    // (the table's column addresses are formed again here: kept from load_group across the round loop they cost the 128-VGPR variant of
    // step32_kernel four registers' worth of scratch spills)
    uint32_t gi_out = gi;
    RG_FRESH_VGPR(gi_out);
    if (active) store_group(p.t, gi_out, g, pe, F);
}

template <int F, bool SPARSE>
__global__ __launch_bounds__(2 * BLOCK) void step_split_kernel(const StepParams p)
{
    __shared__ alignas(16) unsigned char smem[SplitLds<F, false>::BYTES];
    split_body<F, SPARSE, false>(p, smem);
}

// ---- the 32-bit body on compact rows ------------------------------------------------------------------------------------------
// Same protocol as split_body, with everything that crosses LDS half as wide and one round more of slack:
//   event   r+2 : {class word, aux, n, header} and {a, b, c, d} as two 16-byte LDS stores into a FOUR-slot ring, written two rounds ahead (the
//                 I/O wavefront is never the one that is waited for). (Round 3 had the deciding wavefront read event r+1 while it decided
//                 event r, into a second register set, the round loop written out twice: worth 2 % then, nothing since tier 1 went to
//                 sign words — the two copies of the loop disagree about the registers of the group image, 11 copies per round —
//                 profiles/r04g_io_trims_and_prefetch_ab.jsonl. Gone.)
//   outcome r-1 : {resp_term, predicate word, role_epoch, commit} and {log_from, term, votedFor, role} as two 16-byte rows (two-slot ring)
// and no header-addressed loads at all: the term shared by the carried entries is IN the row (RG_HDR_SAME_TERM), so the I/O wavefront's
// stream is two loads per row, issued six rounds ahead of the decision, four rows in registers (the loop is unrolled by four: no copies).
// The deciding wavefront keeps a GroupN (rg_tier1n.hpp); a row that tier 1 does not decide is handed to the general handlers on a widened copy
// of the image and the result is narrowed back.
// Returns false when the workgroup left the 32-bit domain (a group or a Leadership.State value at or above 2^30 at load, a row field
// outside [0, 2^30), a state value the general handlers pushed to STATE_LIMIT): nothing of this body's work counts then.
struct Row32 { U32x2 h; I32x4 q; };

// What the I/O wavefront makes of a compact row for the deciding wavefront of the 32-bit body: {class word, aux, n, header'}. The class word
// (rg_tier1n.hpp: CW_*) holds every fact about the row that does not depend on the group's state, one bit each, plus the follower index of an
// ack's responder. Most of it depends on (kind, slot) only — cluster size and own slot are launch constants — and comes from a 256-entry
// table in LDS that the I/O wavefront fills before the first round (class_entry); what depends on the row's other fields is cleared per row
// (class_word). header' is the header as loaded, with KIND_OUT_OF_DOMAIN when a field leaves [0, EV_LIMIT) — `aux` too where it is a role
// epoch (acks, vote replies, timeouts) or the entries' term: such a row has no class and sends the workgroup to the 64-bit body.
constexpr int CW_AUXC = 18;         // table only: aux of this kind is a role epoch (it must be in the domain)
__device__ __forceinline__ uint32_t class_entry(const StepParams &p, uint32_t idx)
{
    const uint32_t kind = idx & 15u, slot = idx >> 4;
    const uint32_t P = (uint32_t)p.cluster, self = (uint32_t)p.self;
    const bool peer_ok = (slot < P) & (slot != self);
    const bool ack = (kind == RG_EV_AE_ACK) & peer_ok, vreq = (kind == RG_EV_RV_REQ) | (kind == RG_EV_PV_REQ),
               vrep = (kind == RG_EV_RV_REPLY) | (kind == RG_EV_PV_REPLY);
    const uint32_t j = ack ? (slot < self ? slot : slot - 1u) : 0u;
    const uint32_t cls = (ack ? 1u << CW_ACK : 0u) | (((kind == RG_EV_AE_REQ) & (slot < P)) ? 1u << CW_AE : 0u) | ((kind == RG_EV_CLIENT_APPEND) ? 1u << CW_CLIENT : 0u) |
                         (((ack | (kind == RG_EV_IS_ACK)) & peer_ok) ? 1u << CW_ACKANY : 0u) |
                         ((kind - (uint32_t)RG_EV_RV_REQ <= (uint32_t)(RG_EV_TIMEOUT - RG_EV_RV_REQ)) ? 1u << CW_ELK : 0u) |
                         ((vrep & peer_ok) ? 1u << CW_VR : 0u) | ((kind == RG_EV_PV_REPLY) ? 1u << CW_PV : 0u) | ((kind == RG_EV_TIMEOUT) ? 1u << CW_TO : 0u) |
                         ((vreq & (slot < P)) ? 1u << CW_VQ : 0u) | ((kind == RG_EV_PV_REQ) ? 1u << CW_PVQ : 0u) | ((kind == RG_EV_NONE) ? 1u << CW_NONE : 0u) |
                         ((((0x1CCu >> kind) & 1u) != 0) ? 1u << CW_AUXC : 0u);              // kinds 2, 3, 6, 7, 8
    return cls | (j << 10) | (slot << 5) | (31u - j);
}
// Per row, in sign words (rg_tier1n.hpp): AE stays only with prevLogTerm != 0 and no entries, or at most RG_MAX_AE_ENTRIES of one term;
// CLIENT only with n >= 1; nothing stays when a field is out of the domain.
__device__ __forceinline__ I32x4 class_word(const uint32_t *lutc, const Row32 &x)
{
    const uint32_t hdr = x.h.x, aux = x.h.y;
    const uint32_t e = lutc[hdr & 0xFFu];
    const int32_t n = (int32_t)(hdr >> 12);
    const sw same = (int32_t)(hdr << (31 - 10));                                           // RG_HDR_SAME_TERM
    const int32_t n_lim = (same >> 31) & (int32_t)RG_MAX_AE_ENTRIES;
    const sw ae_bad = ~s_pos(x.q.z) | s_lt(n_lim, n), cli_bad = s_lt(n, 1);
    const uint32_t auxm = (uint32_t)(((int32_t)(e << (31 - CW_AUXC)) | ((int32_t)(e << (31 - CW_AE)) & same)) >> 31);    // aux counts: a role epoch, or the entries' term
    const uint32_t w = (uint32_t)x.q.x | (uint32_t)x.q.y | (uint32_t)x.q.z | (uint32_t)x.q.w | (aux & auxm);
    const uint32_t ood = (uint32_t)((int32_t)(w | (w << 1)) >> 31);                         // a field >= 2^30
    const uint32_t off = (((uint32_t)ae_bad & 0x80000000u) >> (31 - CW_AE)) | (((uint32_t)cli_bad & 0x80000000u) >> (31 - CW_CLIENT)) | (1u << CW_AUXC) |
                         (ood & 0xFFF80000u);
    const uint32_t cw = (e & ~off) | ((hdr & 0x100u) << (CW_FLAG - 8));
    return I32x4{(int32_t)cw, (int32_t)aux, n, (int32_t)(hdr | (ood & KIND_OUT_OF_DOMAIN))};
}
// the flags | status << 16 of a predicate word, from two 128-entry tables (main block: bits 6..0, election block: bits 13..7)
__device__ __forceinline__ uint32_t expand_by_table(const uint32_t *lutm, const uint16_t *lute, uint32_t w)
{
    const uint32_t fast = lutm[w & 127u] | (uint32_t)lute[(w >> 7) & 127u];
    return ((int32_t)w < 0) ? (w & 0x00FFFFFFu) : fast;
}

// IOW = 2 (experiment, -DRG_IOW2: launches of at most one workgroup per pair of SIMDs): the I/O work of a workgroup on TWO wavefronts — wave 1 fetches
// and publishes events, wave 2 retires outcomes and keeps the tallies — so that neither is ever what the deciding wavefront waits for at the
// barrier; all three meet at the same s_barrier every round.
// RPB (round 6): ROUNDS PER HAND-OVER. The two wavefronts drain their LDS queues and meet once every RPB rounds instead of once per round: with RPB = 2 the
// deciding wavefront decides rounds 2i and 2i + 1 back to back while the I/O wavefront publishes events 2i + 2, 2i + 3 and retires outcomes 2i - 2, 2i - 1;
// the event ring keeps its four slots, the outcome ring gets four (2 * RPB). Half the s_waitcnt lgkmcnt(0) + s_barrier pairs, no instruction added per round.
template <int F, bool SPARSE, bool OUT32, int IOW = 1, int RPB = 1>
__device__ __forceinline__ bool narrow_body(const StepParams &p, unsigned char *smem)
{
    static_assert(RPB == 1 || (RPB == 2 && IOW == 1), "two rounds per hand-over: the two-wavefront workgroup only");
    constexpr int NO = 2 * RPB;
    typedef SplitLds<F, true, NO> L;
    I32x4 *sh_rec = reinterpret_cast<I32x4 *>(smem + L::N_REC);
    int32_t *sh_mv = reinterpret_cast<int32_t *>(smem + L::N_MV);
    I32x4 (*sh_evh)[BLOCK] = reinterpret_cast<I32x4 (*)[BLOCK]>(smem + L::N_EVH);
    I32x4 (*sh_evq)[BLOCK] = reinterpret_cast<I32x4 (*)[BLOCK]>(smem + L::N_EVQ);
    I32x4 (*sh_o0)[BLOCK] = reinterpret_cast<I32x4 (*)[BLOCK]>(smem + L::N_O0);
    I32x4 (*sh_o1)[BLOCK] = reinterpret_cast<I32x4 (*)[BLOCK]>(smem + L::N_O1);
    // The bail mark: 0, 1 (the state load left the domain) or r + 2 (round r did). Written by the deciding wavefront before a barrier, read by
    // the I/O wavefront after it — a plain LDS word (a `volatile` generic pointer would become a FLAT load whose vmcnt(0) drains the I/O
    // wavefront's whole prefetch every round). The deciding wavefront may already be one round further and have written a LATER round's
    // mark when the I/O wavefront looks: only a mark that is due makes it leave, so both always pass the same number of barriers.
    uint32_t *sh_bail = reinterpret_cast<uint32_t *>(smem + L::N_BAIL);
    uint32_t *sh_lutc = reinterpret_cast<uint32_t *>(smem + L::N_LUTC), *sh_lutm = reinterpret_cast<uint32_t *>(smem + L::N_LUTM);
    uint16_t *sh_lute = reinterpret_cast<uint16_t *>(smem + L::N_LUTE);
    uint32_t *sh_luts = reinterpret_cast<uint32_t *>(smem + L::N_LUTS);

    const uint32_t lane = threadIdx.x & (BLOCK - 1);
    const bool io_wave = __builtin_amdgcn_readfirstlane(threadIdx.x) >= (uint32_t)BLOCK;       // wave-uniform
    const uint32_t i = blockIdx.x * BLOCK + lane;
    const bool active = i < p.count;
    const uint32_t ir = active ? i : p.count - 1u;
    const uint32_t last_round = p.rounds - 1u;

    if (io_wave) {
        RG_HWID_BEGIN(1);
        // (wave-uniform roles: with one I/O wavefront it has both)
        const bool do_pub = IOW == 1 || __builtin_amdgcn_readfirstlane(threadIdx.x) < 2u * BLOCK;
        const bool do_ret = IOW == 1 || !do_pub;
        // Addresses. The five columns this loop touches are [round][row] arrays: a row's address is a SCALAR (column base + round * count * size,
        // 64-bit, computed on the scalar unit) plus the lane's own 32-bit byte offset, which never changes — the form global_load / global_store
        // take directly (saddr + voffset), no vector address arithmetic in the loop. The bases are kernel arguments; each gets scalar registers
        // of its own here: left inside the 16-dword argument tuple they are spilled to VGPR lanes as a whole and re-read piecewise (v_readlane)
        // at every use, 30 vector instructions per round (round 3 measured that as free — it was, while this wavefront was not what a round
        // waits for and the SIMD it shares with a deciding wavefront had issue slots to spare; neither holds any more: profiles/r04e_probe.txt).
        // (launch_compact refuses batches of 2^28 rows per round or more: the lane offset must fit 32 bits)
        // (OUT32: the always-written column is rg_out32_t, the conditional one rg_persist32_t; there is no third)
        uint64_t b_head = reinterpret_cast<uint64_t>(p.head), b_q = reinterpret_cast<uint64_t>(p.abcd32),
                 b_reply = OUT32 ? reinterpret_cast<uint64_t>(p.out32) : reinterpret_cast<uint64_t>(p.reply),
                 b_logfx = reinterpret_cast<uint64_t>(p.logfx),
                 b_persist = OUT32 ? reinterpret_cast<uint64_t>(p.persist32) : reinterpret_cast<uint64_t>(p.persist);
        RG_OWN_SGPRS(b_head); RG_OWN_SGPRS(b_q); RG_OWN_SGPRS(b_reply); RG_OWN_SGPRS(b_logfx); RG_OWN_SGPRS(b_persist);
        const uint64_t round_rows = p.count;
        const uint32_t irm = ir & 0x0FFFFFFFu;           // (spelled out for the instruction selector: ir < count < 2^28)
        // the group's index base: only the wide effect rows need it (compact outcome rows carry indices relative to it, like the event rows)
        int64_t io_base = 0;
        if constexpr (!OUT32) io_base = (p.has_bases != 0) ? p.t.ibase[SPARSE ? p.gid[ir] : ir] : 0;
        auto fetch = [&](uint32_t r, Row32 &x) {
            const uint64_t rb = (uint64_t)(r < p.rounds ? r : last_round) * round_rows;
            x.h = nt_load_at<U32x2>(b_head + rb * 8u, irm);
            x.q = nt_load_at<I32x4>(b_q + rb * 16u, irm);
        };
        // the tables (only this wavefront reads them: no barrier needed, its own LDS operations are ordered)
#pragma unroll
        for (int k = 0; k < 4; k++) sh_lutc[lane + 64u * k] = class_entry(p, lane + 64u * k);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t i7 = lane + 64u * k;
            sh_lutm[i7] = expand_predicates(i7);                                              // (no election bit set)
            sh_lute[i7] = (uint16_t)(expand_predicates((i7 << 7) | 0x42u) & 0xFFFFu);       // (main bits of "no main class": fa_n and fc_n set)
        }
        sh_luts[lane] = Tally::status_entry(lane);
        __builtin_amdgcn_wave_barrier();                 // (every lane reads entries other lanes wrote: ordered on the hardware, a meeting point for the host emulation's lane threads)
        auto publish = [&](uint32_t slot, const Row32 &x) {
            sh_evh[slot][lane] = class_word(sh_lutc, x);
            sh_evq[slot][lane] = x.q;
        };
        Tally tally;
        auto retire = [&](uint32_t r, uint32_t hdr) {
            const uint32_t slot = r & (uint32_t)(NO - 1);
            const uint64_t rb16 = (uint64_t)r * round_rows * 16u;
            const I32x4 o0 = sh_o0[slot][lane], o1 = sh_o1[slot][lane];
            // the deciding wavefront hands over truth values, not flags (rg_tier1n.hpp): the flags, the role field and the "valid iff REPLIED" rule are made here
            const uint32_t flags_all = expand_by_table(sh_lutm, sh_lute, (uint32_t)o0.y) | ((uint32_t)o1.w << RG_F_ROLE_SHIFT), flags = flags_all & 0xFFFFu, status = RG_F_STATUS(flags_all);
            // (lanes past the end of the batch shadow its last row in everything — same group, same events, same outcome: they store it again,
            // to the same address; an exec mask around three stores costs more than the duplicates of one workgroup)
            if constexpr (OUT32) {
                // ONE row that every lane stores: {RaftResponse.term, flags, commitIndex after the row, log_from} — every value of this body fits,
                // so RG_F_WIDE_VALUES never appears here; log_from is only meaningful under its flags (rg_out32_t), no select spent on it
                nt_store_at(b_reply + rb16, irm, I32x4{(flags & RG_F_REPLIED) ? o0.x : 0, (int32_t)flags_all, o0.w, o1.x});
                if ((flags & RG_F_PERSIST) != 0) nt_store_at(b_persist + rb16, irm, I32x4{o1.y, o1.z, o0.z, o1.w});
            } else {
                rg_reply_t rep;
                // (every term / index of this body is in [0, 2^31): zero-extended below, no v_ashr)
                rep.resp_term = (flags & RG_F_REPLIED) ? (int64_t)(uint32_t)o0.x : 0; rep.flags = flags_all; rep.role_epoch = (uint32_t)o0.z;
                nt_store_at(b_reply + rb16, irm, rep);
                const bool w_lfx = ((flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0) | (status == RG_NEED_HOST);
                // (rg_logfx_t speaks absolute indices: the image's values go back on the group's base)
                if (w_lfx) nt_store_at(b_logfx + rb16, irm, I64x2{to_abs((int64_t)(uint32_t)o0.w, io_base), to_abs((int64_t)(uint32_t)o1.x, io_base)});
                if ((flags & RG_F_PERSIST) != 0) {
                    rg_persist_t per;
                    per.term = (int64_t)(uint32_t)o1.y; per.voted_for = o1.z; per.role = o1.w;
                    nt_store_at(b_persist + rb16, irm, per);
                }
            }
            tally.add_packed_lut(RG_HDR_KIND(hdr), flags, status, sh_luts);
        };
        // Rows r+2 .. r+5 are in registers at the top of round r, row k in buf[k & 3]; the loop is unrolled by four so that the indices are
        // compile-time constants. Round r publishes row r+2 and re-fills its registers with row r+6.
        Row32 buf[4];
        uint32_t hdr_m1 = 0u, hdr_0, hdr_1;              // headers of rounds r-1, r, r+1 (the tallies need the kind of a retired row)
        if (do_pub) {
            Row32 e0, e1;
            fetch(0, e0); fetch(1, e1); fetch(2, buf[2]); fetch(3, buf[3]); fetch(4, buf[0]); fetch(5, buf[1]);
            publish(0u, e0); publish(1u, e1);
            hdr_0 = e0.h.x; hdr_1 = e1.h.x;
        } else {
            hdr_0 = hdr_1 = 0u;
            for (int k = 0; k < 4; k++) { buf[k].h = U32x2{0u, 0u}; buf[k].q = I32x4{0, 0, 0, 0}; }
        }
        lds_barrier();                                   // events 0, 1 and the mark of the state load are visible
        { const uint32_t seen0 = *sh_bail; if (__builtin_amdgcn_readfirstlane(seen0) == 1u) return false; }
        // The mark is READ right after a barrier and LOOKED AT just before the next one: the read's latency is spent on this round's own work,
        // and a workgroup that is leaving has published an event and stored an outcome row too many: the outcome row is rewritten by the 64-bit
        // body, the event lands in LDS that body is about to reuse — hence the barrier between the two bodies in step32_kernel.
        bool bailed = false;
        uint32_t seen = 0u;                              // the mark as it stood after the previous barrier
        RG_PROBE_BEGIN();
        if constexpr (RPB == 2) {
            // Two rounds per hand-over. Step k of the unrolled loop (round r = r0 + k) publishes event r + 2, retires outcome r - 2 — written in the PREVIOUS
            // pair of rounds, visible since the last barrier — and re-fills its registers with row r + 6; the wavefronts meet after every odd round (and
            // after an odd last one). hk[i]: header of the newest published row with index = i mod 4 (the tallies need the kind of a retired row): before step
            // k overwrites hk[(k + 2) & 3] with row r + 2's, it holds row r - 2's.
            uint32_t hk[4] = {hdr_0, hdr_1, 0u, 0u};
            for (uint32_t r0 = 0; r0 < p.rounds && !bailed; r0 += 4u) {
                if ((r0 & 127u) == 124u) tally.spill();
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t r = r0 + (uint32_t)k;
                    if (r >= p.rounds) break;
                    Row32 &x = buf[(k + 2) & 3];
                    publish((r + 2u) & 3u, x);
                    RG_PROBE_MARK(0);
                    if (r >= 2u) retire(r - 2u, hk[(k + 2) & 3]);
                    hk[(k + 2) & 3] = x.h.x;
                    fetch(r + 6u, x);
                    RG_PROBE_MARK(1);
                    if ((k & 1) != 0 || r == last_round) {
                        const uint32_t mark = __builtin_amdgcn_readfirstlane(seen), pair = r >> 1;      // pair `pair - 1` (or an earlier one) left the domain: mark <= pair + 1
                        if ((mark != 0u) & (mark <= pair + 1u)) { bailed = true; break; }
                        lds_barrier();
                        RG_PROBE_MARK(2);
                        seen = *sh_bail;
                    }
                }
            }
            if (bailed) return false;
            { const uint32_t mark = __builtin_amdgcn_readfirstlane(seen); if (mark != 0u) return false; }      // the last pair did
            for (uint32_t q = p.rounds >= 2u ? p.rounds - 2u : 0u; q < p.rounds; q++) {      // the last pair's outcomes
                const uint32_t i = q & 3u;
                retire(q, i == 0u ? hk[0] : (i == 1u ? hk[1] : (i == 2u ? hk[2] : hk[3])));
            }
            tally.spill();
#if defined(RG_PROBE)
            RG_PROBE_FLUSH(0);
#elif defined(RG_PROBE_HWID)
            RG_HWID_END(1);
#else
            tally.flush(p, lane, active);
#endif
            return true;
        }
        for (uint32_t r0 = 0; r0 < p.rounds && !bailed; r0 += 4u) {
            if ((r0 & 127u) == 124u) tally.spill();       // (the byte counters of add_packed)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t r = r0 + (uint32_t)k;
                if (r >= p.rounds) break;
                Row32 &x = buf[(k + 2) & 3];
                if (do_pub) publish((r + 2u) & 3u, x);
                RG_PROBE_MARK(0);
                if constexpr (IOW == 1) {
                    if (r > 0) retire(r - 1u, hdr_m1);
                    hdr_m1 = hdr_0; hdr_0 = hdr_1; hdr_1 = x.h.x;
                } else {
                    // (the retiring wavefront never saw the row: its kind is still in the event ring — slot (r - 1) & 3 is rewritten in round r + 1)
                    if (do_ret && r > 0) retire(r - 1u, (uint32_t)sh_evh[(r - 1u) & 3u][lane].w);
                }
                if (do_pub) fetch(r + 6u, x);
                RG_PROBE_MARK(1);
                const uint32_t mark = __builtin_amdgcn_readfirstlane(seen);      // round r - 1 (or earlier) left the domain: mark <= r + 1
                if ((mark != 0u) & (mark <= r + 1u)) { bailed = true; break; }
                lds_barrier();
                RG_PROBE_MARK(2);
                seen = *sh_bail;
            }
        }
        if (bailed) return false;
        { const uint32_t mark = __builtin_amdgcn_readfirstlane(seen); if (mark != 0u) return false; }      // the last round did
        if constexpr (IOW == 1) retire(last_round, hdr_m1);
        else if (do_ret) retire(last_round, (uint32_t)sh_evh[last_round & 3u][lane].w);
        tally.spill();
#if defined(RG_PROBE)
        if (do_pub) RG_PROBE_FLUSH(0);
#elif defined(RG_PROBE_HWID)
        if (do_pub) RG_HWID_END(1);
#else
        if (do_ret) tally.flush(p, lane, active);
#endif
        return true;
    }

    // ---- the deciding wavefront ----------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    RG_HWID_BEGIN(0);
    const uint32_t gi = SPARSE ? p.gid[ir] : ir;
    GroupN g;
    PeersNarrow<F> pe;
    // The group's index base (0 unless the host set one): the image below is relative to it (to_rel). A table without bases — every wavefront whose 64
    // bases are all 0 — skips the conversions at both ends of the launch behind ONE wave-uniform flag; where there are bases they wait in LDS for the
    // general handlers and the write-back (kept in two VGPRs across the round loop they cost it 1.8 %, re-read from the table at the end of the launch a
    // serialised memory round trip: 3.5 % — same-box A/Bs, profiles/r05e, r05f).
    int64_t *sh_base = reinterpret_cast<int64_t *>(smem + L::N_BASE) + lane;
    pe.rec = sh_rec + lane; pe.mv = sh_mv + lane * 4; pe.overflow = false; pe.base = 0;
    const bool FAST = p.fast_paths != 0;
    bool in_domain, any_base;
    {
        const int64_t base = (p.has_bases != 0) ? p.t.ibase[gi] : 0;       // (a table that never had a base: not even the load)
        any_base = __builtin_amdgcn_ballot_w64(base != 0) != 0;
        *sh_base = base;
        pe.base = base;
        Group g64;
        load_group(p.t, gi, g64);
        if (any_base) g64 = group_to_rel(g64, base);
        // (role epochs grow by at most two per round: a launch of fewer than 2^24 rounds cannot take one out of s_ne()'s domain;
        //  with a base the epoch must lie above it: the one index tier 1 computes from — prepareReplication's epoch.index + 1 — is then never "0 + 1")
        in_domain = fits32(g64, EV_LIMIT) & small_fields_fit(g64) & (p.force_wide == 0) & (p.rounds < (1u << 24)) & ((base == 0) | (g64.epoch_index > 0)) & (base >= 0);
        narrow_into(g, g64);
        g.nallow = FAST ? 0 : -1;
        g.recache();
        in_domain = in_domain & stage_peers<F>(p.t, gi, g64.prepared, pe);
    }
    bool bailed = __builtin_amdgcn_ballot_w64(!in_domain) != 0;
    if (lane == 0) *sh_bail = bailed ? 1u : 0u;
    bool blocked = false;
    lds_barrier();
    if (bailed) return false;
    RG_PROBE_BEGIN();
#ifdef RG_EVENT_PREFETCH
    // (experiment, round 6) The event of round r + 1 is in the ring since the barrier of round r - 1: it is READ at the end of round r, ahead of the
    // drain-and-meet that closes the round, so that its LDS latency is spent inside that wait instead of at the top of the next round.
    I32x4 h_next = sh_evh[0][lane], q_next = sh_evq[0][lane];
#endif
    auto round = [&](const uint32_t r) {
#ifdef RG_EVENT_PREFETCH
        const I32x4 h = h_next, q = q_next;
#else
        const I32x4 h = sh_evh[r & 3u][lane], q = sh_evq[r & 3u][lane];
#endif
        RG_PROBE_MARK(0);
        OutN out;
        const sw done = tier1n<F>(p, g, pe, out, h.x, (uint32_t)h.y, h.z, q.x, q.y, q.z, q.w);
        const bool open = done >= 0;                     // not decided by tier 1: the general handlers — or nothing, for a group blocked after a NEED_HOST
        RG_PROBE_MARK(1);
        // The outcome rows go to LDS BEFORE the branch on "somebody needs the general handlers": the ballot's scalar result is not there yet when the
        // branch is issued (the VALU -> SALU round trip, profiles/r04b_issue_bench.txt), and these stores are independent work to spend that wait on.
        // A round that does visit the general handlers writes the rows again afterwards.
        const uint64_t b_open = __builtin_amdgcn_ballot_w64(open);
        const uint32_t slot = r & (uint32_t)(NO - 1);
        sh_o0[slot][lane] = I32x4{out.resp, (int32_t)out.pw, (int32_t)g.role_epoch, g.commit};
        sh_o1[slot][lane] = I32x4{out.log_from, g.term, g.voted_for, g.role};
        if (b_open != 0) {
            // (the header as loaded, KIND_OUT_OF_DOMAIN apart; the general handlers expect the same-term mark where decorate<true> puts it)
            const uint32_t hdr = ((uint32_t)h.w & ~(7u << 9)) | ((((uint32_t)h.w & RG_HDR_SAME_TERM) != 0) ? HDR_SAME_IN : 0u), aux = (uint32_t)h.y, kind = RG_HDR_KIND(hdr);
            const bool skip = open & blocked & (kind != RG_EV_NONE);
            // tier 1.5 (rg_tier1n.hpp): four row classes an election or a cache miss leaves behind, decided on the 32-bit image without widening it
            bool park;
            const bool slow = open & !skip & !tier15<F>(p, g, pe, out, park, open & !skip, h.x, h.y, h.z, q.x, q.y, q.z, q.w);
            if (park) { blocked = true; g.nallow = -1; g.recache(); }
            RG_NOTE_SLOW(slow, lane == 0);                  // (the host emulation counts the rows and wave-rounds that reach the general handlers)
            bool bail = slow & (kind == KIND_OUT_OF_DOMAIN);
            if (slow & !bail) {
                // the general handlers decide on ABSOLUTE values: the image and the row's index fields are taken off the base, the results put back on it
                const int64_t base = any_base ? *sh_base : 0;
                pe.base = base;
                Group g64 = group_to_abs(widen(g), base);
                Stepper<F, PeersNarrow<F>> st(p, g64, pe);
                const Entries en = entries_of<true>(p, hdr, aux, 0, 0, 0, 0);
                const uint32_t ix = index_fields(kind);
                st.run(hdr, aux, (ix & 1u) ? to_abs(q.x, base) : (int64_t)q.x, (ix & 2u) ? to_abs(q.y, base) : (int64_t)q.y,
                       (ix & 4u) ? to_abs(q.z, base) : (int64_t)q.z, (ix & 8u) ? to_abs(q.w, base) : (int64_t)q.w, false, 0, 0, en, entries_readable(p, en, aux, RG_HDR_N(hdr)));
                g64 = group_to_rel(g64, base);
                st.fx.log_from = to_rel(st.fx.log_from, base);
                bail = !fits32(g64, STATE_LIMIT) | !small_fields_fit(g64) | pe.overflow | ((base != 0) & (g64.epoch_index <= 0)) |
                       (((uint64_t)st.fx.resp_term | (uint64_t)st.fx.log_from) >= (uint64_t)STATE_LIMIT);
                narrow_into(g, g64);
                if (st.fx.status == RG_NEED_HOST) blocked = true;
                g.nallow = (FAST & !blocked) ? 0 : -1;
                g.recache();
                out.pw = PW_SLOW | st.fx.flags | (st.fx.status << RG_F_STATUS_SHIFT);
                out.resp = (int32_t)st.fx.resp_term; out.log_from = (int32_t)st.fx.log_from;
            }
            if (skip) { out.pw = PW_SLOW | ((uint32_t)RG_SKIPPED_AFTER_NEED_HOST << RG_F_STATUS_SHIFT); out.resp = 0; out.log_from = 0; }
            if (__builtin_amdgcn_ballot_w64(bail) != 0) {
                if (lane == 0) *sh_bail = r / (uint32_t)RPB + 2u;      // (the hand-over this round belongs to, + 2)
                bailed = true;
            }
            sh_o0[slot][lane] = I32x4{out.resp, (int32_t)out.pw, (int32_t)g.role_epoch, g.commit};
            sh_o1[slot][lane] = I32x4{out.log_from, g.term, g.voted_for, g.role};
        } else {
            RG_NOTE_SLOW(false, lane == 0);
        }
        RG_PROBE_MARK(2);
#ifdef RG_EVENT_PREFETCH
        h_next = sh_evh[(r + 1u) & 3u][lane]; q_next = sh_evq[(r + 1u) & 3u][lane];
#endif
        if constexpr (RPB == 1) {
            lds_barrier();
            RG_PROBE_MARK(3);
        }
    };
    if constexpr (RPB == 2) {
        for (uint32_t r = 0; r < p.rounds; r++) {        // (ONE copy of the round's code — the general handlers are inlined in it —, the round counter a scalar)
            round(r);
            if (bailed) break;
            if ((r & 1u) != 0u || r == last_round) {     // the hand-over: after every odd round and after an odd last one (two scalar tests)
                lds_barrier();
                RG_PROBE_MARK(3);
            }
        }
        if (bailed) lds_barrier();                       // (the hand-over of the pair that left the domain: the I/O wavefront counts it too)
    } else {
#ifdef RG_EVENT_PREFETCH
    { uint32_t r = 0; do { round(r); r++; } while (!bailed & (r < p.rounds)); }      // (rounds >= 1: the host never launches an empty batch)
#else
    for (uint32_t r = 0; r < p.rounds; r++) {
        round(r);
        if (bailed) break;
    }
#endif
    }
    if (bailed) return false;
    RG_PROBE_FLUSH(4);
    RG_HWID_END(0);
    if (active) {
        Group g64 = widen(g);
        pe.base = 0;
        if (any_base) {
            const int64_t base = *sh_base;
            pe.base = base;
            g64 = group_to_abs(g64, base);
        }
        uint32_t gi_out = gi;
        RG_FRESH_VGPR(gi_out);
        store_group(p.t, gi_out, g64, pe, F);       // (pe's scalar accessors return absolute values)
    }
    return true;
}

#ifndef RG_NOTE_FALLBACK            // the host emulation (tests/devemu) counts the workgroups that take the 64-bit body; nothing on the GPU
#define RG_NOTE_FALLBACK() ((void)0)
#endif

// Two register budgets. WAVES = 1: whatever the allocator wants (launches of at most one workgroup per pair of SIMDs, 65 536 rows). WAVES = 4:
// at most 128 VGPRs, so that eight workgroups (19 KB of LDS each) are resident per CU — a launch of 131 072 rows then runs in ONE pass with two
// deciding wavefronts per SIMD instead of a pass and a third (round 3's same-box A/B at config 4's shard: 0.1994 -> 0.1307 ms per launch,
// profiles/r03d_w4_ab.jsonl). Since round 4 the allocator asks for 113 / 112 VGPRs: both variants fit either budget, the two are kept for the
// bound itself (a change that needs more registers shows up as spills in the second, not as a launch that takes two passes).
#ifndef RG_ROUNDS_PER_HANDOVER     // 2 = experiment build (round 6): measured SLOWER than a hand-over per round — config 3 0.0659 -> 0.0683 ms, config 2 0.0390 -> 0.0417,
#define RG_ROUNDS_PER_HANDOVER 1   // same box, same session (profiles/r06g_two_rounds_per_handover_ab.jsonl); bit-exact either way (499 GPU tests, the reference's digest)
#endif
template <int F, bool SPARSE, int WAVES, bool OUT32, int IOW = 1>
__global__ __launch_bounds__((1 + IOW) * BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void step32_kernel(const StepParams p)
{
    // (the experiment: two rounds per hand-over where four workgroups share a CU — WAVES = 1: launches of up to 65 536 rows, 23 KB of LDS; one where eight do)
    constexpr int RPB = (WAVES == 1 && IOW == 1) ? RG_ROUNDS_PER_HANDOVER : 1;
    __shared__ alignas(16) unsigned char smem[SplitLds<F, true, 2 * RPB>::BYTES];
    if (narrow_body<F, SPARSE, OUT32, IOW, RPB>(p, smem)) return;
    if (threadIdx.x == 0) { RG_NOTE_FALLBACK(); atomicAdd(p.wide_bodies, 1ull); }      // (a workgroup that left the 32-bit domain: rare by design, counted so a host can see it)
    if constexpr (IOW == 2) {
        // the 64-bit body knows two wavefronts: the third only keeps the barriers' count (one to get here, one per staging, one per round)
        if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 2u * BLOCK) {
            for (uint32_t k = 0; k < p.rounds + 2u; k++) lds_barrier();
            return;
        }
    }
    // Both wavefronts left the 32-bit body after the same number of barriers, but not at the same moment: the I/O wavefront looks at the bail
    // mark one round late (narrow_body) and has by then published one more event into N_EVH / N_EVQ — memory the 64-bit body's follower
    // records (W_MATCH, W_REJ) overlap. Meet once more before anybody stages anything (never executed on the 32-bit path).
    lds_barrier();
    // start over in 64-bit arithmetic
    split_body<F, SPARSE, true, OUT32>(p, smem);
}

// RG_FORCE_WIDE=1 (differential tests; bench.py's int64-body pass): the 64-bit body on compact rows from the start, as a kernel of its own name —
// a profile of a run that has both lists them apart (the 32-bit body's launches are what `roofline` describes)
template <int F, bool SPARSE, int WAVES, bool OUT32>
__global__ __launch_bounds__(2 * BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void step32_wide_kernel(const StepParams p)
{
    __shared__ alignas(16) unsigned char smem[SplitLds<F, true>::BYTES];
    if (threadIdx.x == 0) { RG_NOTE_FALLBACK(); atomicAdd(p.wide_bodies, 1ull); }
    split_body<F, SPARSE, true, OUT32>(p, smem);
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
static hipError_t launch_compact(const StepParams &p, bool sparse, hipStream_t s)
{
    const uint32_t blocks = (p.count + BLOCK - 1) / BLOCK;
    if (blocks == 0) return hipSuccess;
    if (p.count >= (1u << 28)) return hipErrorInvalidValue;       // the I/O wavefront addresses a row as scalar base + 32-bit lane offset
    const bool many = blocks > 1024u;                    // more than one workgroup per pair of SIMDs on a 256-CU part
    const dim3 grid(blocks), wg(2 * BLOCK);
    if (p.out32 != nullptr) {                            // compact outcome rows (rg_submit32c): dense batches only
        if (sparse) return hipErrorInvalidValue;
        if (p.force_wide != 0) {
            if (many) hipLaunchKernelGGL((step32_wide_kernel<F, false, 4, true>), grid, wg, 0, s, p);
            else      hipLaunchKernelGGL((step32_wide_kernel<F, false, 1, true>), grid, wg, 0, s, p);
        } else {
            if (many) hipLaunchKernelGGL((step32_kernel<F, false, 4, true>), grid, wg, 0, s, p);
#ifdef RG_IOW2
            else      hipLaunchKernelGGL((step32_kernel<F, false, 1, true, 2>), grid, dim3(3 * BLOCK), 0, s, p);
#else
            else      hipLaunchKernelGGL((step32_kernel<F, false, 1, true>), grid, wg, 0, s, p);
#endif
        }
        return hipGetLastError();
    }
    if (p.force_wide != 0) {
        if (sparse) {
            if (many) hipLaunchKernelGGL((step32_wide_kernel<F, true, 4, false>), grid, wg, 0, s, p);
            else      hipLaunchKernelGGL((step32_wide_kernel<F, true, 1, false>), grid, wg, 0, s, p);
        } else {
            if (many) hipLaunchKernelGGL((step32_wide_kernel<F, false, 4, false>), grid, wg, 0, s, p);
            else      hipLaunchKernelGGL((step32_wide_kernel<F, false, 1, false>), grid, wg, 0, s, p);
        }
        return hipGetLastError();
    }
    if (sparse) {
        if (many) hipLaunchKernelGGL((step32_kernel<F, true, 4, false>), grid, wg, 0, s, p);
        else      hipLaunchKernelGGL((step32_kernel<F, true, 1, false>), grid, wg, 0, s, p);
    } else {
        if (many) hipLaunchKernelGGL((step32_kernel<F, false, 4, false>), grid, wg, 0, s, p);
        else      hipLaunchKernelGGL((step32_kernel<F, false, 1, false>), grid, wg, 0, s, p);
    }
    return hipGetLastError();
}

// shape: 0 = step_split_kernel, 64 = step_kernel (wide rows); 32 = step32_kernel (compact rows: p.abcd32 set)
template <int F>
static hipError_t launch_f(const StepParams &p, bool sparse, int shape, hipStream_t s)
{
    switch (shape) {
    case 0:  return launch_split<F>(p, sparse, s);
    case 64: return launch_single<F>(p, sparse, s);
    case 32: return launch_compact<F>(p, sparse, s);
    default: return hipErrorInvalidValue;
    }
}

// clusters of 8 .. 15 nodes (ABI 5): the wide-row kernels only — the decision code is generic in F (the quorum select is an insertion network, the
// follower records live in LDS), the compact-row kernels' LDS budget and class word are not
template <int F>
static hipError_t launch_big(const StepParams &p, bool sparse, int shape, hipStream_t s)
{
    switch (shape) {
    case 0:  return launch_split<F>(p, sparse, s);
    case 64: return launch_single<F>(p, sparse, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(const StepParams &p, int followers, bool sparse, int shape, hipStream_t s)
{
    switch (followers) {
#ifndef RG_BUILD_ONLY_F4            // analysis builds (tools/spine.sh): one cluster size, seconds instead of a minute
    case 1: return launch_f<1>(p, sparse, shape, s);
    case 2: return launch_f<2>(p, sparse, shape, s);
    case 3: return launch_f<3>(p, sparse, shape, s);
    case 5: return launch_f<5>(p, sparse, shape, s);
    case 6: return launch_f<6>(p, sparse, shape, s);
    case 7: return launch_big<7>(p, sparse, shape, s);
    case 8: return launch_big<8>(p, sparse, shape, s);
    case 9: return launch_big<9>(p, sparse, shape, s);
    case 10: return launch_big<10>(p, sparse, shape, s);
    case 11: return launch_big<11>(p, sparse, shape, s);
    case 12: return launch_big<12>(p, sparse, shape, s);
    case 13: return launch_big<13>(p, sparse, shape, s);
    case 14: return launch_big<14>(p, sparse, shape, s);
#endif
    case 4: return launch_f<4>(p, sparse, shape, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rg
