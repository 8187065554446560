// rg_kernels.hip — gfx950 kernels of the batched multi-Raft decision engine.
//
// step_kernel<F, SPARSE>: the replacement of the reference EventLoop drain
// (support/EventLoopGroup.java:32-46).  One lane = one raft group for the whole launch:
//   * group state is read ONCE (16-byte coalesced loads from the structure-of-structs table),
//     kept in VGPRs across all `rounds` of the batch, and written back once;
//   * the per-follower Leadership.State columns are staged in LDS ([follower][lane]) only for groups
//     that lead, so the runtime responder slot indexes LDS, not registers;
//   * per round every lane loads its 40-byte event as 8+16+16 B, with the next round's event
//     already in flight (software prefetch) — the event/outcome streams are what HBM sees;
//   * outcomes: the 16-byte reply is always stored; log/commit effects and the durable
//     (term, votedFor) pair are stored only for rows that have them;
//   * decision counters are wave-level: ballot + popcount per round into scalar registers, one
//     atomic per wave per counter at the end.
// Workgroup = one wavefront (64 lanes): lanes never share LDS columns, so no barrier exists anywhere.
#include "rg_device.hpp"

namespace rg {

struct Event {
    uint32_t hdr, aux;
    int64_t a, b, c, d, hx, hy;
};

__device__ __forceinline__ void load_event(const StepParams &p, size_t row, Event &e)
{
    const rg_ev_head_t h = p.head[row];
    const I64x2 ab = p.ab[row], cd = p.cd[row];
    e.hdr = h.hdr; e.aux = h.aux;
    e.a = ab.x; e.b = ab.y; e.c = cd.x; e.d = cd.y;
    e.hx = 0; e.hy = 0;
    if (p.hint != nullptr && RG_HDR_HINT(h.hdr)) { const I64x2 hh = p.hint[row]; e.hx = hh.x; e.hy = hh.y; }
}

template <int F, bool SPARSE>
__global__ __launch_bounds__(BLOCK, 1) void step_kernel(const StepParams p)
{
    __shared__ int64_t sh_epoch[F * BLOCK], sh_next[F * BLOCK], sh_match[F * BLOCK];
    __shared__ int32_t sh_rej[F * BLOCK];

    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * BLOCK + lane;
    const bool active = i < p.count;
    const uint32_t gi = active ? (SPARSE ? p.gid[i] : i) : 0u;
    const uint32_t G = p.t.groups;

    Group g;
    {
        const I64x2 tc = p.t.term_commit[gi], ep = p.t.epoch[gi], w = p.t.window[gi];
        const Ident id = p.t.ident[gi];
        const Elect el = p.t.elect[gi];
        g.term = tc.x; g.commit = tc.y; g.epoch_index = ep.x; g.epoch_term = ep.y; g.first = w.x; g.last = w.y;
        g.voted_for = id.voted_for; g.leader = id.leader; g.role_epoch = id.role_epoch;
        g.role = (int32_t)(id.meta & META_ROLE);
        g.td = (id.meta & META_TD) != 0; g.prepared = (id.meta & META_PREP) != 0;
        g.rc = (int32_t)((id.meta >> META_RC_SHIFT) & 7u);
        g.pending = (id.meta >> META_PEND_SHIFT) & 0x7Fu;
        g.elected_term = el.elected_term; g.elected_epoch = el.elected_epoch; g.votes = el.votes;
        {
            const I64x2 r0 = p.t.runs[gi], r1 = p.t.runs[(size_t)G + gi], r2 = p.t.runs[(size_t)2 * G + gi],
                        r3 = p.t.runs[(size_t)3 * G + gi];
            g.s0 = r0.x; g.t0 = r0.y; g.s1 = r1.x; g.t1 = r1.y; g.s2 = r2.x; g.t2 = r2.y; g.s3 = r3.x; g.t3 = r3.y;
        }
        g.log_dirty = false; g.peers_dirty = false;
    }
    Peers<F> pe{sh_epoch + lane, sh_next + lane, sh_match + lane, sh_rej + lane};
    if (active && g.prepared) {
#pragma unroll
        for (int j = 0; j < F; j++) {
            const I64x2 en = p.t.peer_en[(size_t)j * G + gi];
            const Match m = p.t.peer_m[(size_t)j * G + gi];
            pe.last_epoch[j * BLOCK] = en.x; pe.next_index[j * BLOCK] = en.y;
            pe.match_index[j * BLOCK] = m.match_index; pe.rejection[j * BLOCK] = m.rejection;
        }
    }

    Stepper<F> st(p, g, pe);
    unsigned long long cnt[RG_NUM_COUNTERS];
#pragma unroll
    for (int c = 0; c < RG_NUM_COUNTERS; c++) cnt[c] = 0ull;
    bool blocked = false;

    Event cur, nxt;
    if (active) load_event(p, i, cur);
    for (uint32_t r = 0; r < p.rounds; r++) {
        const size_t row = (size_t)r * p.count + i;
        if (active && r + 1 < p.rounds) load_event(p, row + p.count, nxt);

        uint32_t flags = 0, status = RG_OK, kind = 0;
        if (active) {
            kind = RG_HDR_KIND(cur.hdr);
            if (blocked && kind != RG_EV_NONE) {
                st.fx = Fx{0u, RG_SKIPPED_AFTER_NEED_HOST, 0, 0};
            } else {
                st.run(cur.hdr, cur.aux, cur.a, cur.b, cur.c, cur.d, cur.hx, cur.hy);
            }
            status = st.fx.status;
            if (status == RG_NEED_HOST) blocked = true;
            flags = st.fx.flags;
            rg_reply_t rep;
            rep.resp_term = (flags & RG_F_REPLIED) ? st.fx.resp_term : 0;
            rep.flags = flags | ((uint32_t)g.role << RG_F_ROLE_SHIFT) | (status << RG_F_STATUS_SHIFT);
            rep.role_epoch = g.role_epoch;
            p.reply[row] = rep;
            if ((flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) || status == RG_NEED_HOST)
                p.logfx[row] = I64x2{g.commit, st.fx.log_from};
            if (flags & RG_F_PERSIST) {
                rg_persist_t per;
                per.term = g.term; per.voted_for = g.voted_for; per.role = g.role;
                p.persist[row] = per;
            }
        }
        // wave-level tallies: one ballot + popcount per counter per round
        const bool is_assert = status != RG_OK && status < RG_NPE_MAJOR_NULL;
        cnt[0] += __popcll(__ballot(kind != RG_EV_NONE));
        cnt[1] += __popcll(__ballot((flags & RG_F_REPLIED) != 0));
        cnt[2] += __popcll(__ballot((flags & RG_F_ROLE_CHANGED) != 0));
        cnt[3] += __popcll(__ballot((flags & RG_F_COMMIT) != 0));
        cnt[4] += __popcll(__ballot(is_assert));
        cnt[5] += __popcll(__ballot(status == RG_NEED_HOST));
        cnt[6] += __popcll(__ballot(status == RG_DROPPED_STALE_ROLE));
        cnt[7] += __popcll(__ballot((flags & RG_F_LOG_APPEND) != 0));
        cur = nxt;
    }

    if (active) {
        p.t.term_commit[gi] = I64x2{g.term, g.commit};
        p.t.epoch[gi] = I64x2{g.epoch_index, g.epoch_term};
        p.t.window[gi] = I64x2{g.first, g.last};
        Ident id;
        id.voted_for = g.voted_for; id.leader = g.leader; id.role_epoch = g.role_epoch;
        id.meta = (uint32_t)g.role | (g.td ? META_TD : 0u) | (g.prepared ? META_PREP : 0u) |
                  ((uint32_t)g.rc << META_RC_SHIFT) | (g.pending << META_PEND_SHIFT);
        p.t.ident[gi] = id;
        Elect el;
        el.elected_term = g.elected_term; el.elected_epoch = g.elected_epoch; el.votes = g.votes;
        p.t.elect[gi] = el;
        if (g.log_dirty) {
            p.t.runs[gi] = I64x2{g.s0, g.t0};
            p.t.runs[(size_t)G + gi] = I64x2{g.s1, g.t1};
            p.t.runs[(size_t)2 * G + gi] = I64x2{g.s2, g.t2};
            p.t.runs[(size_t)3 * G + gi] = I64x2{g.s3, g.t3};
        }
        if (g.peers_dirty) {
#pragma unroll
            for (int j = 0; j < F; j++) {
                p.t.peer_en[(size_t)j * G + gi] = I64x2{pe.last_epoch[j * BLOCK], pe.next_index[j * BLOCK]};
                Match m;
                m.match_index = pe.match_index[j * BLOCK]; m.rejection = pe.rejection[j * BLOCK]; m.pad = 0;
                p.t.peer_m[(size_t)j * G + gi] = m;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < RG_NUM_COUNTERS; c++)
            if (cnt[c]) atomicAdd(&p.counters[c], cnt[c]);
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

template <int F>
static hipError_t launch_f(const StepParams &p, bool sparse, hipStream_t s)
{
    const uint32_t blocks = (p.count + BLOCK - 1) / BLOCK;
    if (blocks == 0) return hipSuccess;
    if (sparse) hipLaunchKernelGGL((step_kernel<F, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
    else        hipLaunchKernelGGL((step_kernel<F, false>), dim3(blocks), dim3(BLOCK), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_step(const StepParams &p, int followers, bool sparse, hipStream_t s)
{
    switch (followers) {
    case 1: return launch_f<1>(p, sparse, s);
    case 2: return launch_f<2>(p, sparse, s);
    case 3: return launch_f<3>(p, sparse, s);
    case 4: return launch_f<4>(p, sparse, s);
    case 5: return launch_f<5>(p, sparse, s);
    case 6: return launch_f<6>(p, sparse, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst, bytes / 16);
    return hipGetLastError();
}

}  // namespace rg
