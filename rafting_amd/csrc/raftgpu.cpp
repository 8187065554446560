// raftgpu.cpp — host side of libraftgpu.so: the C-ABI of include/raftgpu.h over the gfx950 kernels.
//
// Owns, per table: the HBM structure-of-structs state (rg_device.hpp DevTable), one HIP stream, grow-only
// staging buffers for RG_MEM_HOST submissions, device decision counters and a pool of HIP event pairs for
// kernel timing.  There is no CPU fallback: every entry point that needs the GPU fails with a HIP error
// string when no device is present.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rg_device.hpp"

namespace rg {
hipError_t launch_step(const StepParams &p, int followers, bool sparse, int shape, hipStream_t s);
hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t s);
hipError_t launch_replicate(const ReplicateParams &p, int followers, hipStream_t s);
hipError_t launch_health_update(const HealthParams &p, hipStream_t s);
hipError_t launch_health_failure(const HealthParams &p, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, hipStream_t s);
hipError_t launch_ready(const HealthParams &p, int64_t now, int32_t cp, int64_t cd, uint8_t *ready, hipStream_t s);
hipError_t launch_timers_update(const TimerParams &p, hipStream_t s);
hipError_t launch_tick_fold(const TickFoldParams &p, hipStream_t s);
hipError_t launch_tick_tail(const TickTailParams &p, int followers, hipStream_t s);
hipError_t launch_tick(const StepParams &p, const TickTailParams &tp, int followers, hipStream_t s);
hipError_t launch_timers_arm(const TimerParams &p, hipStream_t s);
hipError_t launch_timers_expired(int64_t *deadline, const Ident *ident, uint32_t groups, int64_t now, const int64_t *now_mem, uint32_t *counts, uint32_t *total,
                                 uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, hipStream_t s);
hipError_t launch_outcome_count(const rg_reply_t *reply, uint32_t rows, uint32_t *counts, uint32_t *totals, hipStream_t s);
hipError_t launch_outcome_emit(const rg_reply_t *reply, const I64x2 *logfx, const rg_persist_t *persist, uint32_t rows, const uint32_t *counts,
                               I64x2 *out_logfx, uint32_t cap_logfx, rg_persist_t *out_persist, uint32_t cap_persist, hipStream_t s);
}  // namespace rg

using rg::DevTable;
using rg::I64x2;

struct Staging {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct PipeSlot {                               // one batch in flight on the pipelined host-memory path (rg_submit_async)
    Staging gid, head, ab, cd, hint, terms, reply, logfx, persist;
    Staging abcd32, terms32, counts;            // rg_submit_async_packed: narrow uploads, per-wavefront list counts
    hipEvent_t up = nullptr, done = nullptr, down = nullptr;
};

struct rg_table {
    int device = 0;
    uint32_t G = 0, P = 0, self = 0, F = 0;
    int pre_vote = 0;
    hipStream_t stream = nullptr;
    DevTable dt{};
    unsigned long long *counters = nullptr;     // [counter_slots][RG_NUM_COUNTERS], one slot per wave of a dense launch
    unsigned long long *wide_bodies = nullptr;  // one word (rg_wide_body_workgroups)
    size_t counter_slots = 0;
    int fast_paths = 1;                         // RG_FAST=0: general handlers only (differential tests)
    int force_wide = 0;                         // RG_FORCE_WIDE=1: the compact-format kernel skips its 32-bit body (differential tests)
    int require_fence = 0;                      // rg_table_option(RG_OPT_REQUIRE_FENCED_TIMEOUTS)
    int has_bases = 0;                          // rg_index_base_set has stored a non-zero base at some time
    Staging st_abcd32, st_terms32, st_out32, st_persist32;
    int lanes = -1;                             // -1: pick per launch; 0: split kernel; 64: single-wavefront kernel (RG_SPLIT env forces one)
    uint32_t simds = 1024;                      // SIMDs of the device (CUs x 4)
    Staging st_gid, st_head, st_ab, st_cd, st_hint, st_terms, st_reply, st_logfx, st_persist, st_hb, st_fl, st_sh, st_ss;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    uint64_t t_launches = 0;
    double t_total_ms = 0.0;
    hipEvent_t region0 = nullptr, region1 = nullptr;
    int64_t *timer_deadline = nullptr;          // [G] N4
    uint32_t *timer_epoch = nullptr;            // [G] role epoch after the last batch the timers saw (rg_timers_update32 chains compact rows from it)
    unsigned long long *tick_masks = nullptr;   // [ceil(G / 256) * 4 >= workgroups of any tick kernel] status words of the expiry's look-back (rg_kernels.hip: expire_tail)
    uint32_t *tick_ticket = nullptr;            // [1] their generation
    uint64_t config_gen = 0;                    // bumped by whatever changes what a recorded launch has baked in (options, the first index base)
    std::vector<struct rg_tick *> ticks;        // live recordings: invalidated when the table goes
    std::vector<struct rg_tick2 *> ticks2;
    uint32_t *timer_counts = nullptr;           // [waves + 1]: per-wave counts / offsets, last = total
    int64_t election_ms = 900, heartbeat_ms = 300;   // raft1.xml:10-13
    uint64_t timer_seed = 0;
    Staging st_tgid, st_hgid, st_hslot, st_hflag, st_ready;
    int64_t *health_ok = nullptr, *health_fail = nullptr;   // [F][G] N4b
    int32_t *health_recent = nullptr;
    hipStream_t s_in = nullptr, s_out = nullptr; // copy-in / copy-out streams of the pipelined path
    PipeSlot pipe[RG_PIPELINE_DEPTH];
    uint64_t pipe_head = 0, pipe_tail = 0;       // batches submitted / waited for
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(rg_table *t, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (t) t->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(t, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail((t), -2, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

static int wait_oldest(rg_table *t)
{
    if (t->pipe_tail == t->pipe_head) return 1;
    HIP_TRY(t, hipEventSynchronize(t->pipe[t->pipe_tail % RG_PIPELINE_DEPTH].down));
    t->pipe_tail += 1;
    return 0;
}

// every entry point starts here: select the device and (unless the caller IS the pipeline) let batches in flight land first
static int bind(rg_table *t, bool drain = true)
{
    HIP_TRY(t, hipSetDevice(t->device));
    while (drain && t->pipe_tail != t->pipe_head)
        if (int rc = wait_oldest(t); rc < 0) return rc;
    return 0;
}

static int reserve(rg_table *t, Staging &s, size_t bytes)
{
    if (bytes <= s.cap) return 0;
    if (s.ptr) HIP_TRY(t, hipFree(s.ptr));
    s.ptr = nullptr; s.cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIP_TRY(t, hipMalloc(&s.ptr, cap));
    s.cap = cap;
    return 0;
}

static int drain_timing(rg_table *t)
{
    if (t->ev_used == 0) return 0;
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    for (size_t i = 0; i < t->ev_used; i++) {
        float ms = 0.f;
        HIP_TRY(t, hipEventElapsedTime(&ms, t->ev_pool[i].first, t->ev_pool[i].second));
        t->t_total_ms += ms;
        t->t_launches += 1;
    }
    t->ev_used = 0;
    return 0;
}

extern "C" {

int rg_abi_version(void) { return RG_ABI_VERSION; }

const char *rg_last_error(const rg_table_t *t) { return t ? t->err.c_str() : g_create_err.c_str(); }
uint32_t rg_table_groups(const rg_table_t *t) { return t ? t->G : 0; }
uint32_t rg_table_cluster(const rg_table_t *t) { return t ? t->P : 0; }

int rg_table_option(rg_table_t *t, int option, int value)
{
    if (!t) return -1;
    switch (option) {
    case RG_OPT_REQUIRE_FENCED_TIMEOUTS:
        if (t->require_fence != (value != 0)) t->config_gen += 1;       // (a recorded tick has the old value baked in: rg_tick_launch refuses it)
        t->require_fence = value != 0;
        return 0;
    default: return fail(t, -1, "rg_table_option: unknown option %d", option);
    }
}

static void tick_release(struct rg_tick *k);
static void tick2_release(struct rg_tick2 *k);

int rg_table_destroy(rg_table_t *t)
{
    if (!t) return 0;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    // recordings that outlive their table keep no pointer into it: their launch / wait fail with -1, their destroy only frees the handle
    for (struct rg_tick *k : t->ticks) tick_release(k);
    for (struct rg_tick2 *k : t->ticks2) tick2_release(k);
    t->ticks.clear(); t->ticks2.clear();
    void *cols[] = {t->dt.term_commit, t->dt.epoch, t->dt.window, t->dt.ident, t->dt.elect, t->dt.runs,
                    t->dt.peer_en, t->dt.peer_m, t->dt.ibase, t->wide_bodies, t->counters, t->st_gid.ptr, t->st_head.ptr, t->st_ab.ptr,
                    t->st_cd.ptr, t->st_hint.ptr, t->st_terms.ptr, t->st_reply.ptr, t->st_logfx.ptr,
                    t->st_persist.ptr, t->st_hb.ptr, t->st_fl.ptr, t->st_sh.ptr, t->st_ss.ptr, t->timer_deadline, t->timer_epoch, t->tick_masks, t->tick_ticket,
                    t->timer_counts, t->st_tgid.ptr, t->st_hgid.ptr, t->st_hslot.ptr, t->st_hflag.ptr, t->st_ready.ptr, t->health_ok,
                    t->health_fail, t->health_recent, t->st_abcd32.ptr, t->st_terms32.ptr, t->st_out32.ptr, t->st_persist32.ptr};
    for (void *c : cols) if (c) (void)hipFree(c);
    for (PipeSlot &sl : t->pipe) {
        if (sl.down) (void)hipEventSynchronize(sl.down);
        for (Staging *st : {&sl.gid, &sl.head, &sl.ab, &sl.cd, &sl.hint, &sl.terms, &sl.reply, &sl.logfx, &sl.persist, &sl.abcd32, &sl.terms32, &sl.counts})
            if (st->ptr) (void)hipFree(st->ptr);
        for (hipEvent_t ev : {sl.up, sl.done, sl.down}) if (ev) (void)hipEventDestroy(ev);
    }
    if (t->s_in) (void)hipStreamDestroy(t->s_in);
    if (t->s_out) (void)hipStreamDestroy(t->s_out);
    for (auto &e : t->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (t->region0) { (void)hipEventDestroy(t->region0); (void)hipEventDestroy(t->region1); }
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return 0;
}

int rg_table_create(int device, uint32_t groups, uint32_t cluster, uint32_t self_slot, int pre_vote, rg_table_t **out)
{
    if (!out) return fail(nullptr, -1, "rg_table_create: out is NULL");
    *out = nullptr;
    if (groups == 0) return fail(nullptr, -1, "rg_table_create: groups must be > 0");
    if (cluster < RG_MIN_CLUSTER || cluster > RG_MAX_CLUSTER)
        return fail(nullptr, -1, "rg_table_create: cluster size %u outside [%d, %d]", cluster, RG_MIN_CLUSTER, RG_MAX_CLUSTER);
    if (self_slot >= cluster) return fail(nullptr, -1, "rg_table_create: self_slot %u >= cluster %u", self_slot, cluster);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, -2, "rg_table_create: no HIP device (%s) — libraftgpu has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, -1, "rg_table_create: device %d of %d", device, ndev);

    rg_table *t = new rg_table();
    t->device = device; t->G = groups; t->P = cluster; t->self = self_slot; t->F = cluster - 1;
    t->pre_vote = pre_vote != 0;
#define CREATE_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e2_ = (expr);                                                           \
        if (e2_ != hipSuccess) {                                                           \
            fail(nullptr, -2, "%s failed: %s", #expr, hipGetErrorString(e2_));             \
            rg_table_destroy(t);                                                           \
            return -2;                                                                     \
        }                                                                                  \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    const size_t G = groups, F = t->F;
    CREATE_TRY(hipMalloc((void **)&t->dt.term_commit, G * sizeof(I64x2)));
    CREATE_TRY(hipMalloc((void **)&t->dt.epoch, G * sizeof(I64x2)));
    CREATE_TRY(hipMalloc((void **)&t->dt.window, G * sizeof(I64x2)));
    CREATE_TRY(hipMalloc((void **)&t->dt.ident, G * sizeof(rg::Ident)));
    CREATE_TRY(hipMalloc((void **)&t->dt.elect, G * sizeof(rg::Elect)));
    CREATE_TRY(hipMalloc((void **)&t->dt.runs, G * rg::K * sizeof(I64x2)));
    CREATE_TRY(hipMalloc((void **)&t->dt.peer_en, G * F * sizeof(I64x2)));
    CREATE_TRY(hipMalloc((void **)&t->dt.peer_m, G * F * sizeof(rg::Match)));
    CREATE_TRY(hipMalloc((void **)&t->dt.ibase, G * sizeof(int64_t)));
    CREATE_TRY(hipMalloc((void **)&t->wide_bodies, sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(t->wide_bodies, 0, sizeof(unsigned long long), t->stream));
    {
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, device));
        t->simds = (uint32_t)prop.multiProcessorCount * 4u;
    }
    t->lanes = -1;
    if (const char *e = getenv("RG_SPLIT")) t->lanes = atoi(e) != 0 ? 0 : 64;   // force either kernel
    if (const char *e = getenv("RG_FAST")) t->fast_paths = atoi(e) != 0;
    if (const char *e = getenv("RG_FORCE_WIDE")) t->force_wide = atoi(e) != 0;
    t->counter_slots = (G + 63) / 64 + 1;       // one slot per workgroup of a dense launch
    CREATE_TRY(hipMalloc((void **)&t->counters, t->counter_slots * RG_NUM_COUNTERS * sizeof(unsigned long long)));
    t->dt.groups = groups;
    CREATE_TRY(hipMalloc((void **)&t->timer_deadline, G * sizeof(int64_t)));
    // timers_count_kernel runs whole 256-lane workgroups: every one of their wavefronts stores a count, also those past the
    // last group, so the array covers the grid (found by the host emulation: ceil(G/64) + 1 entries were 2 short at G = 300)
    CREATE_TRY(hipMalloc((void **)&t->timer_counts, ((G + 255) / 256 * 4 + 1) * sizeof(uint32_t)));
    CREATE_TRY(hipMemsetAsync(t->timer_deadline, 0, G * sizeof(int64_t), t->stream));
    CREATE_TRY(hipMalloc((void **)&t->tick_masks, ((G + 255) / 256 * 4) * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(t->tick_masks, 0, ((G + 255) / 256 * 4) * sizeof(unsigned long long), t->stream));      // (status words of the expiry's look-back: state 0 = none)
    CREATE_TRY(hipMalloc((void **)&t->tick_ticket, sizeof(uint32_t)));
    CREATE_TRY(hipMemsetAsync(t->tick_ticket, 0, sizeof(uint32_t), t->stream));
    CREATE_TRY(hipMalloc((void **)&t->timer_epoch, G * sizeof(uint32_t)));
    CREATE_TRY(hipMemsetAsync(t->timer_epoch, 0, G * sizeof(uint32_t), t->stream));
    CREATE_TRY(hipMalloc((void **)&t->health_ok, G * F * sizeof(int64_t)));
    CREATE_TRY(hipMalloc((void **)&t->health_fail, G * F * sizeof(int64_t)));
    CREATE_TRY(hipMalloc((void **)&t->health_recent, G * F * sizeof(int32_t)));
    CREATE_TRY(hipMemsetAsync(t->health_ok, 0, G * F * sizeof(int64_t), t->stream));
    CREATE_TRY(hipMemsetAsync(t->health_fail, 0, G * F * sizeof(int64_t), t->stream));
    CREATE_TRY(hipMemsetAsync(t->health_recent, 0, G * F * sizeof(int32_t), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.term_commit, 0, G * sizeof(I64x2), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.epoch, 0, G * sizeof(I64x2), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.window, 0, G * sizeof(I64x2), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.runs, 0, G * rg::K * sizeof(I64x2), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.peer_en, 0, G * F * sizeof(I64x2), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.peer_m, 0, G * F * sizeof(rg::Match), t->stream));
    CREATE_TRY(hipMemsetAsync(t->dt.ibase, 0, G * sizeof(int64_t), t->stream));
    CREATE_TRY(hipMemsetAsync(t->counters, 0, t->counter_slots * RG_NUM_COUNTERS * sizeof(unsigned long long), t->stream));
    {   // fresh groups = what RaftContext.initialize leaves: Follower, term 0, no vote, empty log
        std::vector<rg::Ident> id(G, rg::Ident{RG_NO_NODE, RG_NO_NODE, 1u, 0u});
        std::vector<rg::Elect> el(G, rg::Elect{0, 0u, 1});
        CREATE_TRY(hipMemcpyAsync(t->dt.ident, id.data(), G * sizeof(rg::Ident), hipMemcpyHostToDevice, t->stream));
        CREATE_TRY(hipMemcpyAsync(t->dt.elect, el.data(), G * sizeof(rg::Elect), hipMemcpyHostToDevice, t->stream));
        CREATE_TRY(hipStreamSynchronize(t->stream));
    }
#undef CREATE_TRY
    *out = t;
    return 0;
}

/* ---- state ------------------------------------------------------------------------------------ */

// 0 = step_split_kernel, otherwise the raft groups per wavefront of step_kernel
static int step_lanes(const rg_table *t, uint32_t count)
{
    if (t->lanes >= 0) return t->lanes;
    const uint32_t wavefronts = (count + 63u) / 64u;
    return wavefronts <= t->simds ? 0 : 64;
}

static bool state_complete(const rg_group_state_t *s)
{
    return s->current_term && s->voted_for && s->role && s->current_leader && s->timeout_detected && s->repl_prepared &&
           s->role_epoch && s->votes && s->elected_epoch && s->elected_term && s->commit_index && s->epoch_index &&
           s->epoch_term && s->first_index && s->last_index && s->run_count && s->run_offset && s->run_start && s->run_term &&
           s->peer_last_epoch && s->peer_next_index && s->peer_match_index && s->peer_rejection && s->peer_pending;
}

int rg_load_state(rg_table_t *t, uint32_t first, uint32_t count, const rg_group_state_t *s)
{
    if (!t) return -1;
    if (!s || !state_complete(s)) return fail(t, -1, "rg_load_state: src or one of its columns is NULL");
    if ((uint64_t)first + count > t->G) return fail(t, -1, "rg_load_state: range [%u, %u) exceeds %u groups", first, first + count, t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    const size_t n = count, F = t->F, G = t->G;
    std::vector<I64x2> tc(n), ep(n), win(n), runs(n * rg::K), en(n * F);
    std::vector<rg::Ident> id(n);
    std::vector<rg::Elect> el(n);
    std::vector<rg::Match> pm(n * F);
    for (size_t i = 0; i < n; i++) {
        const int32_t role = s->role[i];
        if (role < RG_FOLLOWER || role > RG_LEADER) return fail(t, -1, "rg_load_state: group %zu role %d", first + i, role);
        if (s->voted_for[i] < RG_NO_NODE || s->voted_for[i] >= (int32_t)t->P || s->current_leader[i] < RG_NO_NODE ||
            s->current_leader[i] >= (int32_t)t->P)
            return fail(t, -1, "rg_load_state: group %zu node slot out of range", first + i);
        const uint32_t total = s->run_count[i], off = s->run_offset[i];
        const uint32_t rc = total > (uint32_t)rg::K ? (uint32_t)rg::K : total;
        int64_t fi = 0, la = 0;
        if (total) {
            fi = s->first_index[i]; la = s->last_index[i];
            if (s->run_start[off] != fi || la < s->run_start[off + total - 1])
                return fail(t, -1, "rg_load_state: group %zu runs do not span [first,last]", first + i);
            if (fi != s->epoch_index[i] && fi != s->epoch_index[i] + 1)
                return fail(t, -1, "rg_load_state: group %zu first index %lld is neither epoch.index nor epoch.index+1",
                            first + i, (long long)fi);
            for (uint32_t k = 1; k < total; k++)
                if (s->run_start[off + k] <= s->run_start[off + k - 1] || s->run_term[off + k] == s->run_term[off + k - 1])
                    return fail(t, -1, "rg_load_state: group %zu runs are not maximal ascending runs", first + i);
        }
        tc[i] = I64x2{s->current_term[i], s->commit_index[i]};
        ep[i] = I64x2{s->epoch_index[i], s->epoch_term[i]};
        win[i] = I64x2{fi, la};
        uint32_t pend = 0;
        for (size_t j = 0; j < F; j++) {
            en[j * n + i] = I64x2{s->peer_last_epoch[i * F + j], s->peer_next_index[i * F + j]};
            pm[j * n + i] = rg::Match{s->peer_match_index[i * F + j], s->peer_rejection[i * F + j], 0};
            if (s->peer_pending[i * F + j]) pend |= 1u << j;
        }
        for (uint32_t k = 0; k < (uint32_t)rg::K; k++) {
            const bool have = k < rc;
            runs[(size_t)k * n + i] = have ? I64x2{s->run_start[off + total - rc + k], s->run_term[off + total - rc + k]} : I64x2{0, 0};
        }
        const uint32_t meta = (uint32_t)role | (s->timeout_detected[i] ? rg::META_TD : 0u) |
                              (s->repl_prepared[i] ? rg::META_PREP : 0u) | (rc << rg::META_RC_SHIFT) |
                              (pend << rg::META_PEND_SHIFT);
        id[i] = rg::Ident{s->voted_for[i], s->current_leader[i], s->role_epoch[i], meta};
        el[i] = rg::Elect{s->elected_term[i], s->elected_epoch[i], s->votes[i]};
    }
    hipStream_t st = t->stream;
    HIP_TRY(t, hipMemcpyAsync(t->dt.term_commit + first, tc.data(), n * sizeof(I64x2), hipMemcpyHostToDevice, st));
    HIP_TRY(t, hipMemcpyAsync(t->dt.epoch + first, ep.data(), n * sizeof(I64x2), hipMemcpyHostToDevice, st));
    HIP_TRY(t, hipMemcpyAsync(t->dt.window + first, win.data(), n * sizeof(I64x2), hipMemcpyHostToDevice, st));
    HIP_TRY(t, hipMemcpyAsync(t->dt.ident + first, id.data(), n * sizeof(rg::Ident), hipMemcpyHostToDevice, st));
    HIP_TRY(t, hipMemcpyAsync(t->dt.elect + first, el.data(), n * sizeof(rg::Elect), hipMemcpyHostToDevice, st));
    for (size_t k = 0; k < (size_t)rg::K; k++)
        HIP_TRY(t, hipMemcpyAsync(t->dt.runs + k * G + first, runs.data() + k * n, n * sizeof(I64x2), hipMemcpyHostToDevice, st));
    for (size_t j = 0; j < F; j++) {
        HIP_TRY(t, hipMemcpyAsync(t->dt.peer_en + j * G + first, en.data() + j * n, n * sizeof(I64x2), hipMemcpyHostToDevice, st));
        HIP_TRY(t, hipMemcpyAsync(t->dt.peer_m + j * G + first, pm.data() + j * n, n * sizeof(rg::Match), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(t, hipMemsetAsync(t->timer_deadline + first, 0, n * sizeof(int64_t), st));   // loaded groups hold no timer ticket yet
    HIP_TRY(t, hipMemcpyAsync(t->timer_epoch + first, s->role_epoch, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    for (size_t j = 0; j < F; j++) {                                                       // ... and fresh health statistics
        HIP_TRY(t, hipMemsetAsync(t->health_ok + j * G + first, 0, n * sizeof(int64_t), st));
        HIP_TRY(t, hipMemsetAsync(t->health_fail + j * G + first, 0, n * sizeof(int64_t), st));
        HIP_TRY(t, hipMemsetAsync(t->health_recent + j * G + first, 0, n * sizeof(int32_t), st));
    }
    HIP_TRY(t, hipStreamSynchronize(st));
    return 0;
}

int rg_read_state(rg_table_t *t, uint32_t first, uint32_t count, rg_group_state_t *d)
{
    if (!t) return -1;
    if (!d || !state_complete(d)) return fail(t, -1, "rg_read_state: dst or one of its columns is NULL");
    if ((uint64_t)first + count > t->G) return fail(t, -1, "rg_read_state: range exceeds %u groups", t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    const size_t n = count, F = t->F, G = t->G;
    std::vector<I64x2> tc(n), ep(n), win(n), runs(n * rg::K), en(n * F);
    std::vector<rg::Ident> id(n);
    std::vector<rg::Elect> el(n);
    std::vector<rg::Match> pm(n * F);
    hipStream_t st = t->stream;
    HIP_TRY(t, hipMemcpyAsync(tc.data(), t->dt.term_commit + first, n * sizeof(I64x2), hipMemcpyDeviceToHost, st));
    HIP_TRY(t, hipMemcpyAsync(ep.data(), t->dt.epoch + first, n * sizeof(I64x2), hipMemcpyDeviceToHost, st));
    HIP_TRY(t, hipMemcpyAsync(win.data(), t->dt.window + first, n * sizeof(I64x2), hipMemcpyDeviceToHost, st));
    HIP_TRY(t, hipMemcpyAsync(id.data(), t->dt.ident + first, n * sizeof(rg::Ident), hipMemcpyDeviceToHost, st));
    HIP_TRY(t, hipMemcpyAsync(el.data(), t->dt.elect + first, n * sizeof(rg::Elect), hipMemcpyDeviceToHost, st));
    for (size_t k = 0; k < (size_t)rg::K; k++)
        HIP_TRY(t, hipMemcpyAsync(runs.data() + k * n, t->dt.runs + k * G + first, n * sizeof(I64x2), hipMemcpyDeviceToHost, st));
    for (size_t j = 0; j < F; j++) {
        HIP_TRY(t, hipMemcpyAsync(en.data() + j * n, t->dt.peer_en + j * G + first, n * sizeof(I64x2), hipMemcpyDeviceToHost, st));
        HIP_TRY(t, hipMemcpyAsync(pm.data() + j * n, t->dt.peer_m + j * G + first, n * sizeof(rg::Match), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(t, hipStreamSynchronize(st));
    for (size_t i = 0; i < n; i++) {
        const uint32_t meta = id[i].meta;
        const uint32_t rc = (meta >> rg::META_RC_SHIFT) & 7u;
        d->current_term[i] = tc[i].x; d->commit_index[i] = tc[i].y;
        d->epoch_index[i] = ep[i].x; d->epoch_term[i] = ep[i].y;
        d->first_index[i] = rc ? win[i].x : 0; d->last_index[i] = rc ? win[i].y : 0;
        d->voted_for[i] = id[i].voted_for; d->current_leader[i] = id[i].leader; d->role_epoch[i] = id[i].role_epoch;
        d->role[i] = (int32_t)(meta & rg::META_ROLE);
        d->timeout_detected[i] = (meta & rg::META_TD) ? 1 : 0;
        d->repl_prepared[i] = (meta & rg::META_PREP) ? 1 : 0;
        d->elected_term[i] = el[i].elected_term; d->elected_epoch[i] = el[i].elected_epoch; d->votes[i] = el[i].votes;
        d->run_count[i] = rc;
        d->run_offset[i] = (uint32_t)(i * rg::K);
        for (uint32_t k = 0; k < (uint32_t)rg::K; k++) {
            d->run_start[i * rg::K + k] = k < rc ? runs[(size_t)k * n + i].x : 0;
            d->run_term[i * rg::K + k] = k < rc ? runs[(size_t)k * n + i].y : 0;
        }
        const uint32_t pend = (meta >> rg::META_PEND_SHIFT) & rg::META_PEND_MASK;
        for (size_t j = 0; j < F; j++) {
            d->peer_last_epoch[i * F + j] = en[j * n + i].x;
            d->peer_next_index[i * F + j] = en[j * n + i].y;
            d->peer_match_index[i * F + j] = pm[j * n + i].match_index;
            d->peer_rejection[i * F + j] = pm[j * n + i].rejection;
            d->peer_pending[i * F + j] = (pend >> j) & 1u;
        }
    }
    return 0;
}

/* ---- the hot path ----------------------------------------------------------------------------- */

static int launch(rg_table *t, const rg::StepParams &p, bool sparse)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (t->timing) {
        if (t->ev_used == t->ev_pool.size()) {
            if (t->ev_pool.size() >= 4096) {
                if (drain_timing(t)) return -2;
            } else {
                hipEvent_t a, b;
                HIP_TRY(t, hipEventCreate(&a));
                HIP_TRY(t, hipEventCreate(&b));
                t->ev_pool.emplace_back(a, b);
            }
        }
        e0 = t->ev_pool[t->ev_used].first; e1 = t->ev_pool[t->ev_used].second;
        HIP_TRY(t, hipEventRecord(e0, t->stream));
    }
    // Up to one wavefront of groups per SIMD, a lone deciding wavefront leaves half of its SIMD's issue slots empty:
    // give it an I/O partner (step_split_kernel). With more groups the SIMDs are shared by several deciding
    // wavefronts anyway and the single-wavefront kernel is the faster one.
    HIP_TRY(t, rg::launch_step(p, (int)t->F, sparse, p.abcd32 ? 32 : step_lanes(t, p.count), t->stream));
    if (t->timing) {
        HIP_TRY(t, hipEventRecord(e1, t->stream));
        t->ev_used += 1;
    }
    return 0;
}

// the compact formats belong to clusters of up to RG_MAX_COMPACT_CLUSTER nodes (include/raftgpu.h)
static int compact_ok(rg_table *t, const char *who)
{
    if (t->P > RG_MAX_COMPACT_CLUSTER)
        return fail(t, -1, "%s: a cluster of %u nodes is decided on wide rows (rg_submit, rg_submit_async); the compact formats end at %d nodes", who, t->P, RG_MAX_COMPACT_CLUSTER);
    return 0;
}

static int check_batch(rg_table *t, const rg_batch_t *in, const rg_outcome_t *out, bool host_memory)
{
    if (!in || !out) return fail(t, -1, "rg_submit: NULL batch or outcome");
    if (!in->head || !in->ab || !in->cd || !out->reply || !out->logfx || !out->persist)
        return fail(t, -1, "rg_submit: head/ab/cd/reply/logfx/persist are required");
    if (in->rounds == 0) return fail(t, -1, "rg_submit: rounds must be >= 1");
    const bool sparse = in->gid != nullptr;
    if (sparse) {
        if (in->rounds != 1) return fail(t, -1, "rg_submit: sparse batches carry exactly one round");
        if (in->count > t->G) return fail(t, -1, "rg_submit: %u rows for %u groups", in->count, t->G);
    } else if (in->count != t->G) {
        return fail(t, -1, "rg_submit: dense batch has %u rows per round, table has %u groups", in->count, t->G);
    }
    if (in->entry_count && !in->entry_terms) return fail(t, -1, "rg_submit: entry_count without entry_terms");
    if (in->entry_count > (1ull << 29)) return fail(t, -1, "rg_submit: at most 2^29 entry terms per batch (%llu given)", (unsigned long long)in->entry_count);
    if (sparse && host_memory) {
        for (uint32_t i = 0; i < in->count; i++) {
            if (in->gid[i] >= t->G) return fail(t, -1, "rg_submit: gid[%u]=%u out of range", i, in->gid[i]);
            if (i && in->gid[i] <= in->gid[i - 1]) return fail(t, -1, "rg_submit: gid must be strictly ascending (row %u)", i);
        }
    }
    return 0;
}

static rg::StepParams step_params(rg_table *t, const rg_batch_t *in)
{
    rg::StepParams p{};
    p.t = t->dt;
    p.rounds = in->rounds; p.count = in->count;
    p.entry_count = in->entry_count;
    p.counters = t->counters;
    p.wide_bodies = t->wide_bodies;
    p.self = (int32_t)t->self; p.cluster = (int32_t)t->P; p.majority = (int32_t)(t->P / 2 + 1); p.pre_vote = t->pre_vote;
    p.fast_paths = t->fast_paths;
    p.force_wide = t->force_wide;
    p.require_fence = t->require_fence;
    p.has_bases = t->has_bases;
    return p;
}

static int pipeline_ready(rg_table *t);

/* the pipelined host-memory path: see include/raftgpu.h */
int rg_submit_async(rg_table_t *t, const rg_batch_t *in, const rg_outcome_t *out)
{
    if (!t) return -1;
    if (int rc = check_batch(t, in, out, true)) return rc;
    if (in->count == 0) return 0;
    if (bind(t, false)) return -2;
    if (int rc = pipeline_ready(t)) return rc;
    PipeSlot &sl = t->pipe[t->pipe_head % RG_PIPELINE_DEPTH];
    const bool sparse = in->gid != nullptr;
    const size_t rows = (size_t)in->rounds * in->count;
    rg::StepParams p = step_params(t, in);
    if (reserve(t, sl.head, rows * sizeof(rg_ev_head_t)) || reserve(t, sl.ab, rows * sizeof(I64x2)) || reserve(t, sl.cd, rows * sizeof(I64x2)) ||
        reserve(t, sl.reply, rows * sizeof(rg_reply_t)) || reserve(t, sl.logfx, rows * sizeof(I64x2)) || reserve(t, sl.persist, rows * sizeof(rg_persist_t)))
        return -2;
    hipStream_t si = t->s_in, so = t->s_out;
    HIP_TRY(t, hipMemcpyAsync(sl.head.ptr, in->head, rows * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, si));
    HIP_TRY(t, hipMemcpyAsync(sl.ab.ptr, in->ab, rows * sizeof(I64x2), hipMemcpyHostToDevice, si));
    HIP_TRY(t, hipMemcpyAsync(sl.cd.ptr, in->cd, rows * sizeof(I64x2), hipMemcpyHostToDevice, si));
    p.head = (const rg_ev_head_t *)sl.head.ptr; p.ab = (const I64x2 *)sl.ab.ptr; p.cd = (const I64x2 *)sl.cd.ptr;
    if (sparse) {
        if (reserve(t, sl.gid, in->count * sizeof(uint32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(sl.gid.ptr, in->gid, in->count * sizeof(uint32_t), hipMemcpyHostToDevice, si));
        p.gid = (const uint32_t *)sl.gid.ptr;
    }
    if (in->hint) {
        if (reserve(t, sl.hint, rows * sizeof(I64x2))) return -2;
        HIP_TRY(t, hipMemcpyAsync(sl.hint.ptr, in->hint, rows * sizeof(I64x2), hipMemcpyHostToDevice, si));
        p.hint = (const I64x2 *)sl.hint.ptr;
    }
    if (in->entry_count) {
        if (reserve(t, sl.terms, in->entry_count * sizeof(int64_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(sl.terms.ptr, in->entry_terms, in->entry_count * sizeof(int64_t), hipMemcpyHostToDevice, si));
        p.entry_terms = (const int64_t *)sl.terms.ptr;
    }
    HIP_TRY(t, hipEventRecord(sl.up, si));
    HIP_TRY(t, hipStreamWaitEvent(t->stream, sl.up, 0));                  // the kernel of this batch needs its upload ...
    p.reply = (rg_reply_t *)sl.reply.ptr; p.logfx = (I64x2 *)sl.logfx.ptr; p.persist = (rg_persist_t *)sl.persist.ptr;
    if (int rc = launch(t, p, sparse)) return rc;                          // ... and, being on the table's stream, runs after the batch before it
    HIP_TRY(t, hipEventRecord(sl.done, t->stream));
    HIP_TRY(t, hipStreamWaitEvent(so, sl.done, 0));
    HIP_TRY(t, hipMemcpyAsync(out->reply, sl.reply.ptr, rows * sizeof(rg_reply_t), hipMemcpyDeviceToHost, so));
    HIP_TRY(t, hipMemcpyAsync(out->logfx, sl.logfx.ptr, rows * sizeof(I64x2), hipMemcpyDeviceToHost, so));
    HIP_TRY(t, hipMemcpyAsync(out->persist, sl.persist.ptr, rows * sizeof(rg_persist_t), hipMemcpyDeviceToHost, so));
    HIP_TRY(t, hipEventRecord(sl.down, so));
    t->pipe_head += 1;
    return 0;
}

static int pipeline_ready(rg_table *t)
{
    if (!t->s_in) {
        HIP_TRY(t, hipStreamCreateWithFlags(&t->s_in, hipStreamNonBlocking));
        HIP_TRY(t, hipStreamCreateWithFlags(&t->s_out, hipStreamNonBlocking));
        for (PipeSlot &sl : t->pipe) {
            HIP_TRY(t, hipEventCreate(&sl.up)); HIP_TRY(t, hipEventCreate(&sl.done)); HIP_TRY(t, hipEventCreate(&sl.down));
        }
    }
    if (t->pipe_head - t->pipe_tail == RG_PIPELINE_DEPTH)
        if (int rc = wait_oldest(t); rc < 0) return rc;
    return 0;
}

/* the pipelined host-memory path with compact transfer formats: see include/raftgpu.h */
int rg_submit_async_packed(rg_table_t *t, const rg_batch32_t *in, const rg_outcome_packed_t *out)
{
    if (!t) return -1;
    if (int rc = compact_ok(t, "rg_submit_async_packed")) return rc;
    if (!in || !out) return fail(t, -1, "rg_submit_async_packed: null batch or outcome");
    if (!in->head || !in->abcd || !out->reply || !out->counts || (out->logfx_cap && !out->logfx) || (out->persist_cap && !out->persist))
        return fail(t, -1, "rg_submit_async_packed: head, abcd, reply, counts and every list with a capacity are required");
    if (in->entry_count && !in->entry_terms) return fail(t, -1, "rg_submit_async_packed: entry_count %llu without entry_terms", (unsigned long long)in->entry_count);
    rg_batch_t wide{};                              // the same shape rules as every other submission (rounds, count, gid list, entry bound)
    wide.rounds = in->rounds; wide.count = in->count; wide.gid = in->gid; wide.head = in->head;
    wide.ab = wide.cd = reinterpret_cast<const rg_ev_pair_t *>(in->abcd);      // presence only: check_batch does not read event fields
    wide.entry_terms = reinterpret_cast<const int64_t *>(in->entry_terms); wide.entry_count = in->entry_count;
    rg_logfx_t dummy_l; rg_persist_t dummy_p;
    const rg_outcome_t shape{out->reply, &dummy_l, &dummy_p};
    if (int rc = check_batch(t, &wide, &shape, true)) return rc;
    if (in->count == 0) { out->counts[0] = out->counts[1] = 0; return 0; }
    const size_t rows64 = (size_t)in->rounds * in->count;
    if (rows64 >= (1ull << 31)) return fail(t, -1, "rg_submit_async_packed: %zu rows in one batch (limit 2^31 - 1)", rows64);
    if (bind(t, false)) return -2;
    // the lists are written by the device into the caller's memory: it has to be page-locked and mapped
    void *d_logfx = nullptr, *d_persist = nullptr, *d_counts = nullptr;
    if (hipHostGetDevicePointer(&d_counts, out->counts, 0) != hipSuccess ||
        (out->logfx_cap && hipHostGetDevicePointer(&d_logfx, out->logfx, 0) != hipSuccess) ||
        (out->persist_cap && hipHostGetDevicePointer(&d_persist, out->persist, 0) != hipSuccess)) {
        (void)hipGetLastError();
        return fail(t, -1, "rg_submit_async_packed: counts / logfx / persist must be page-locked memory from rg_host_alloc");
    }
    if (int rc = pipeline_ready(t)) return rc;
    PipeSlot &sl = t->pipe[t->pipe_head % RG_PIPELINE_DEPTH];
    const bool sparse = in->gid != nullptr;
    const uint32_t rows = (uint32_t)rows64, waves = (rows + 63u) / 64u;
    rg::StepParams p = step_params(t, &wide);
    if (reserve(t, sl.head, rows64 * sizeof(rg_ev_head_t)) || reserve(t, sl.abcd32, rows64 * sizeof(rg_ev_quad32_t)) ||
        reserve(t, sl.reply, rows64 * sizeof(rg_reply_t)) || reserve(t, sl.logfx, rows64 * sizeof(I64x2)) ||
        reserve(t, sl.persist, rows64 * sizeof(rg_persist_t)) || reserve(t, sl.counts, (size_t)2 * waves * sizeof(uint32_t)))
        return -2;
    hipStream_t si = t->s_in, so = t->s_out;
    HIP_TRY(t, hipMemcpyAsync(sl.head.ptr, in->head, rows64 * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, si));
    HIP_TRY(t, hipMemcpyAsync(sl.abcd32.ptr, in->abcd, rows64 * sizeof(rg_ev_quad32_t), hipMemcpyHostToDevice, si));
    if (sparse) {
        if (reserve(t, sl.gid, in->count * sizeof(uint32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(sl.gid.ptr, in->gid, in->count * sizeof(uint32_t), hipMemcpyHostToDevice, si));
        p.gid = (const uint32_t *)sl.gid.ptr;
    }
    if (in->entry_count) {
        if (reserve(t, sl.terms32, in->entry_count * sizeof(int32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(sl.terms32.ptr, in->entry_terms, in->entry_count * sizeof(int32_t), hipMemcpyHostToDevice, si));
    }
    HIP_TRY(t, hipEventRecord(sl.up, si));
    HIP_TRY(t, hipStreamWaitEvent(t->stream, sl.up, 0));
    // the compact-format kernel reads the rows as they arrived: no widening pass
    p.head = (const rg_ev_head_t *)sl.head.ptr; p.abcd32 = (const rg::I32x4 *)sl.abcd32.ptr;
    p.hint = nullptr;
    p.entry_terms32 = in->entry_count ? (const int32_t *)sl.terms32.ptr : nullptr;
    p.reply = (rg_reply_t *)sl.reply.ptr; p.logfx = (I64x2 *)sl.logfx.ptr; p.persist = (rg_persist_t *)sl.persist.ptr;
    if (int rc = launch(t, p, sparse)) return rc;
    HIP_TRY(t, rg::launch_outcome_count(p.reply, rows, (uint32_t *)sl.counts.ptr, (uint32_t *)d_counts, t->stream));
    HIP_TRY(t, hipEventRecord(sl.done, t->stream));
    HIP_TRY(t, hipStreamWaitEvent(so, sl.done, 0));
    // the scatter is link-bound (it writes host memory): on the copy-out stream, so the next batch's kernels do not queue behind it
    HIP_TRY(t, rg::launch_outcome_emit(p.reply, p.logfx, p.persist, rows, (const uint32_t *)sl.counts.ptr, (I64x2 *)d_logfx, out->logfx_cap,
                                       (rg_persist_t *)d_persist, out->persist_cap, so));
    HIP_TRY(t, hipMemcpyAsync(out->reply, sl.reply.ptr, rows64 * sizeof(rg_reply_t), hipMemcpyDeviceToHost, so));
    HIP_TRY(t, hipEventRecord(sl.down, so));
    t->pipe_head += 1;
    return 0;
}

/* the once-per-tick path: see include/raftgpu.h */
struct rg_tick {
    rg_table *t = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    Staging head, abcd, gid, terms, reply, logfx, persist, counts;
    uint64_t config_gen = 0;         // the table's at creation: has_bases, require_fence, fast_paths, force_wide are baked into the recorded kernel node
    bool in_flight = false;
};

// everything a recording holds on the device; afterwards it knows no table (called with the table's device current and its stream idle)
static void tick_release(struct rg_tick *k)
{
    if (k->exec) (void)hipGraphExecDestroy(k->exec);
    if (k->graph) (void)hipGraphDestroy(k->graph);
    k->exec = nullptr; k->graph = nullptr;
    for (Staging *st : {&k->head, &k->abcd, &k->gid, &k->terms, &k->reply, &k->logfx, &k->persist, &k->counts}) {
        if (st->ptr) (void)hipFree(st->ptr);
        st->ptr = nullptr; st->cap = 0;
    }
    k->t = nullptr;
}

int rg_tick_destroy(rg_tick_t *k)
{
    if (!k) return 0;
    if (rg_table *t = k->t) {
        (void)hipSetDevice(t->device);
        if (t->stream) (void)hipStreamSynchronize(t->stream);
        t->ticks.erase(std::remove(t->ticks.begin(), t->ticks.end(), k), t->ticks.end());
        tick_release(k);
    }
    delete k;
    return 0;
}

// a buffer a recorded copy replays from / into must be page-locked: a pageable one is either rejected late by the capture with an opaque error, or
// captured with undefined semantics
static bool page_locked(const void *p)
{
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

int rg_tick_create(rg_table_t *t, const rg_batch32_t *in, const rg_outcome_packed_t *out, rg_tick_t **tick)
{
    if (!t) return -1;
    if (int rc = compact_ok(t, "rg_tick_create")) return rc;
    if (!tick) return fail(t, -1, "rg_tick_create: tick is NULL");
    *tick = nullptr;
    if (!in || !out) return fail(t, -1, "rg_tick_create: null batch or outcome");
    if (!in->head || !in->abcd || !out->reply || !out->counts || (out->logfx_cap && !out->logfx) || (out->persist_cap && !out->persist))
        return fail(t, -1, "rg_tick_create: head, abcd, reply, counts and every list with a capacity are required");
    if (in->entry_count && !in->entry_terms) return fail(t, -1, "rg_tick_create: entry_count %llu without entry_terms", (unsigned long long)in->entry_count);
    rg_batch_t wide{};
    wide.rounds = in->rounds; wide.count = in->count; wide.gid = nullptr; wide.head = in->head;      // (a sparse gid list is read at every launch, not now)
    wide.ab = wide.cd = reinterpret_cast<const rg_ev_pair_t *>(in->abcd);
    wide.entry_terms = reinterpret_cast<const int64_t *>(in->entry_terms); wide.entry_count = in->entry_count;
    rg_logfx_t dummy_l; rg_persist_t dummy_p;
    const rg_outcome_t shape{out->reply, &dummy_l, &dummy_p};
    if (in->gid) {
        if (in->rounds != 1) return fail(t, -1, "rg_tick_create: sparse batches carry exactly one round");
        if (in->count > t->G || in->count == 0) return fail(t, -1, "rg_tick_create: %u rows for %u groups", in->count, t->G);
    } else if (int rc = check_batch(t, &wide, &shape, true)) return rc;
    if (in->rounds == 0 || in->count == 0) return fail(t, -1, "rg_tick_create: an empty shape");
    const size_t rows64 = (size_t)in->rounds * in->count;
    if (rows64 >= (1ull << 31)) return fail(t, -1, "rg_tick_create: %zu rows in one batch (limit 2^31 - 1)", rows64);
    if (bind(t)) return -2;
    void *d_logfx = nullptr, *d_persist = nullptr, *d_counts = nullptr;
    if (hipHostGetDevicePointer(&d_counts, out->counts, 0) != hipSuccess ||
        (out->logfx_cap && hipHostGetDevicePointer(&d_logfx, out->logfx, 0) != hipSuccess) ||
        (out->persist_cap && hipHostGetDevicePointer(&d_persist, out->persist, 0) != hipSuccess)) {
        (void)hipGetLastError();
        return fail(t, -1, "rg_tick_create: counts / logfx / persist must be page-locked memory from rg_host_alloc");
    }
    if (!page_locked(in->head) || !page_locked(in->abcd) || !page_locked(out->reply) || (in->gid && !page_locked(in->gid)) ||
        (in->entry_count && !page_locked(in->entry_terms)))
        return fail(t, -1, "rg_tick_create: head / abcd / gid / entry_terms / reply must be page-locked memory from rg_host_alloc (the recorded copies replay from them)");
    rg_tick *k = new rg_tick();
    k->t = t;
    k->config_gen = t->config_gen;
    const uint32_t rows = (uint32_t)rows64, waves = (rows + 63u) / 64u;
    const bool sparse = in->gid != nullptr;
    if (reserve(t, k->head, rows64 * sizeof(rg_ev_head_t)) || reserve(t, k->abcd, rows64 * sizeof(rg_ev_quad32_t)) || reserve(t, k->reply, rows64 * sizeof(rg_reply_t)) ||
        reserve(t, k->logfx, rows64 * sizeof(I64x2)) || reserve(t, k->persist, rows64 * sizeof(rg_persist_t)) || reserve(t, k->counts, (size_t)2 * waves * sizeof(uint32_t)) ||
        (sparse && reserve(t, k->gid, in->count * sizeof(uint32_t))) || (in->entry_count && reserve(t, k->terms, in->entry_count * sizeof(int32_t)))) {
        rg_tick_destroy(k);
        return -2;
    }
    rg::StepParams p = step_params(t, &wide);
    p.head = (const rg_ev_head_t *)k->head.ptr; p.abcd32 = (const rg::I32x4 *)k->abcd.ptr;
    p.gid = sparse ? (const uint32_t *)k->gid.ptr : nullptr;
    p.entry_terms32 = in->entry_count ? (const int32_t *)k->terms.ptr : nullptr;
    p.reply = (rg_reply_t *)k->reply.ptr; p.logfx = (I64x2 *)k->logfx.ptr; p.persist = (rg_persist_t *)k->persist.ptr;
    hipStream_t s = t->stream;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) e = hipMemcpyAsync(k->head.ptr, in->head, rows64 * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(k->abcd.ptr, in->abcd, rows64 * sizeof(rg_ev_quad32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && sparse) e = hipMemcpyAsync(k->gid.ptr, in->gid, in->count * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && in->entry_count) e = hipMemcpyAsync(k->terms.ptr, in->entry_terms, in->entry_count * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = rg::launch_step(p, (int)t->F, sparse, 32, s);
    if (e == hipSuccess) e = rg::launch_outcome_count(p.reply, rows, (uint32_t *)k->counts.ptr, (uint32_t *)d_counts, s);
    if (e == hipSuccess) e = rg::launch_outcome_emit(p.reply, p.logfx, p.persist, rows, (const uint32_t *)k->counts.ptr, (I64x2 *)d_logfx, out->logfx_cap,
                                                     (rg_persist_t *)d_persist, out->persist_cap, s);
    if (e == hipSuccess) e = hipMemcpyAsync(out->reply, k->reply.ptr, rows64 * sizeof(rg_reply_t), hipMemcpyDeviceToHost, s);
    hipGraph_t g = nullptr;
    const hipError_t e2 = hipStreamEndCapture(s, &g);           // (always: an open capture would poison the stream)
    k->graph = g;
    if (e == hipSuccess) e = e2;
    if (e == hipSuccess) e = hipGraphInstantiate(&k->exec, k->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rg_tick_destroy(k);
        return fail(t, -2, "rg_tick_create: %s", hipGetErrorString(e));
    }
    t->ticks.push_back(k);
    *tick = k;
    return 0;
}

int rg_tick_launch(rg_tick_t *k)
{
    if (!k || !k->t) return -1;
    rg_table *t = k->t;
    if (k->config_gen != t->config_gen)
        return fail(t, -1, "rg_tick_launch: the table's options or index bases changed after rg_tick_create (they are part of the recording): create the tick again");
    if (bind(t)) return -2;
    if (k->in_flight) { HIP_TRY(t, hipStreamSynchronize(t->stream)); k->in_flight = false; }
    HIP_TRY(t, hipGraphLaunch(k->exec, t->stream));
    k->in_flight = true;
    return 0;
}

int rg_tick_wait(rg_tick_t *k)
{
    if (!k || !k->t) return -1;
    rg_table *t = k->t;
    HIP_TRY(t, hipSetDevice(t->device));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    k->in_flight = false;
    return 0;
}

static rg::TimerParams timer_params(rg_table *t);
static rg::HealthParams health_params(rg_table *t);

/* the device-resident tick: see include/raftgpu.h */
struct rg_tick2 {
    rg_table *t = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t config_gen = 0;
    bool in_flight = false;
};

static void tick2_release(struct rg_tick2 *k)
{
    if (k->exec) (void)hipGraphExecDestroy(k->exec);
    if (k->graph) (void)hipGraphDestroy(k->graph);
    k->exec = nullptr; k->graph = nullptr;
    k->t = nullptr;
}

int rg_tick2_destroy(rg_tick2_t *k)
{
    if (!k) return 0;
    if (rg_table *t = k->t) {
        (void)hipSetDevice(t->device);
        if (t->stream) (void)hipStreamSynchronize(t->stream);
        t->ticks2.erase(std::remove(t->ticks2.begin(), t->ticks2.end(), k), t->ticks2.end());
        tick2_release(k);
    }
    delete k;
    return 0;
}

// the device's address of memory the caller says is device-visible: page-locked host memory is mapped, device memory is what it is; pageable host
// memory (which a kernel would fault on) is refused where the runtime can tell
static int device_visible(rg_table *t, const void *p, const char *what, void **out)
{
    *out = nullptr;
    if (!p) return 0;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess || a.type == hipMemoryTypeUnregistered) {
        (void)hipGetLastError();
        return fail(t, -1, "rg_tick2_create: %s is neither device memory (rg_dev_alloc) nor page-locked host memory (rg_host_alloc)", what);
    }
    if (a.type == hipMemoryTypeHost) {
        if (hipHostGetDevicePointer(out, const_cast<void *>(p), 0) != hipSuccess) {
            (void)hipGetLastError();
            return fail(t, -1, "rg_tick2_create: %s is host memory the device cannot address", what);
        }
        return 0;
    }
    *out = const_cast<void *>(p);
    return 0;
}

int rg_tick2_create(rg_table_t *t, const rg_tick2_io_t *io, rg_tick2_t **tick)
{
    if (!t) return -1;
    if (int rc = compact_ok(t, "rg_tick2_create")) return rc;
    if (!tick) return fail(t, -1, "rg_tick2_create: tick is NULL");
    *tick = nullptr;
    if (!io) return fail(t, -1, "rg_tick2_create: io is NULL");
    if (io->rounds == 0 || io->rounds > 64) return fail(t, -1, "rg_tick2_create: %u rounds (1 .. 64: one clock per round travels with the tick)", io->rounds);
    if (!io->head || !io->abcd || !io->now || !io->row || !io->persist32) return fail(t, -1, "rg_tick2_create: head, abcd, now, row and persist32 are required");
    if (io->entry_capacity && !io->entry_terms) return fail(t, -1, "rg_tick2_create: entry_capacity %llu without entry_terms", (unsigned long long)io->entry_capacity);
    if (io->expired_gid && (!io->expired_count || io->expired_capacity == 0)) return fail(t, -1, "rg_tick2_create: the expiry step needs expired_count and a capacity");
    if ((io->send_head == nullptr) != (io->send == nullptr)) return fail(t, -1, "rg_tick2_create: send_head and send come together");
    rg_batch_t wide{};
    wide.rounds = io->rounds; wide.count = t->G; wide.head = io->head;
    wide.ab = wide.cd = reinterpret_cast<const rg_ev_pair_t *>(io->abcd);
    wide.entry_terms = reinterpret_cast<const int64_t *>(io->entry_terms); wide.entry_count = io->entry_capacity;
    rg_reply_t dummy_r; rg_logfx_t dummy_l; rg_persist_t dummy_p;
    const rg_outcome_t shape{&dummy_r, &dummy_l, &dummy_p};
    if (int rc = check_batch(t, &wide, &shape, false)) return rc;
    if (bind(t)) return -2;
    void *d_head, *d_abcd, *d_terms, *d_now, *d_hb, *d_fl, *d_row, *d_per, *d_egid, *d_eep, *d_ecnt, *d_sh, *d_ss, *d_ready;
    if (device_visible(t, io->head, "head", &d_head) || device_visible(t, io->abcd, "abcd", &d_abcd) || device_visible(t, io->entry_terms, "entry_terms", &d_terms) ||
        device_visible(t, io->now, "now", &d_now) || device_visible(t, io->heartbeat, "heartbeat", &d_hb) || device_visible(t, io->in_flight, "in_flight", &d_fl) ||
        device_visible(t, io->row, "row", &d_row) || device_visible(t, io->persist32, "persist32", &d_per) || device_visible(t, io->expired_gid, "expired_gid", &d_egid) ||
        device_visible(t, io->expired_epoch, "expired_epoch", &d_eep) || device_visible(t, io->expired_count, "expired_count", &d_ecnt) ||
        device_visible(t, io->send_head, "send_head", &d_sh) || device_visible(t, io->send, "send", &d_ss) || device_visible(t, io->ready, "ready", &d_ready))
        return -1;
    rg_tick2 *k = new rg_tick2();
    k->t = t;
    k->config_gen = t->config_gen;
    const uint32_t G = t->G;
    // the decisions: rg_submit32c on device-visible rows
    rg::StepParams sp = step_params(t, &wide);
    sp.head = (const rg_ev_head_t *)d_head; sp.abcd32 = (const rg::I32x4 *)d_abcd;
    sp.entry_terms32 = io->entry_capacity ? (const int32_t *)d_terms : nullptr;
    sp.out32 = (rg::I32x4 *)d_row; sp.persist32 = (rg::I32x4 *)d_per;
    // what the batch did to the timers and to the followers' health, from the compact rows where they lie
    rg::TimerParams tp = timer_params(t);
    tp.rounds = io->rounds; tp.count = G; tp.out32 = (const rg::I32x4 *)d_row; tp.persist32 = (const rg::I32x4 *)d_per; tp.now_mem = (const int64_t *)d_now;
    rg::HealthParams hp = health_params(t);
    hp.rounds = io->rounds; hp.count = G; hp.head = (const rg_ev_head_t *)d_head; hp.out32 = (const rg::I32x4 *)d_row; hp.now_mem = (const int64_t *)d_now;
    const int64_t *now_last = (const int64_t *)d_now + (io->rounds - 1);
    rg::HealthParams rp = health_params(t);
    rp.now_mem = now_last;
    rg::ReplicateParams qp{};
    qp.t = t->dt; qp.count = G; qp.heartbeat = (const uint8_t *)d_hb; qp.in_flight = (const uint16_t *)d_fl; qp.head = (rg_send_head_t *)d_sh; qp.send = (rg_send_t *)d_ss;
    hipStream_t s = t->stream;
    // What follows the decisions — what the batch did to the timers and to the followers' health, the list of the tickets that fired, the leaders' sends, the
    // readiness column — is one lane per group with same-group dependencies only, and a single-round tick is launch-bound (~14 us per graph node for ~10 us of
    // work in all at 65 536 groups). So the recording is ONE kernel node: rg_kernels.hip, tick_kernel — the workgroup that decided 64 groups does the rest for
    // them. RG_TICK_NODES=2 records step + tick_tail_kernel, RG_TICK_NODES=4 the step-by-step form (step, tick_fold, replicate, ready): same-box A/Bs and the
    // differential tests. timer_counts doubles as the per-wavefront masks of the expiry (64 bits each) + its ticket word.
    rg::TickFoldParams fp{};
    fp.tp = tp; fp.hp = hp; fp.now_last = now_last;
    fp.masks = (unsigned long long *)t->tick_masks; fp.ticket = t->tick_ticket;
    fp.out_gid = (uint32_t *)d_egid; fp.out_epoch = (uint32_t *)d_eep; fp.out_count = (uint32_t *)d_ecnt; fp.capacity = io->expired_capacity;
    fp.expire = io->expired_gid != nullptr;
    rg::TickTailParams tt{};
    tt.fp = fp; tt.qp = qp; tt.rp = rp; tt.critical_point = io->critical_point; tt.cool_down = io->cool_down_ms; tt.ready = (uint8_t *)d_ready;
    const char *nodes_env = getenv("RG_TICK_NODES");
    const int nodes = (nodes_env && (nodes_env[0] == '2' || nodes_env[0] == '4')) ? nodes_env[0] - '0' : (sp.force_wide != 0 ? 2 : 1);
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (nodes == 1) {
        if (e == hipSuccess) e = rg::launch_tick(sp, tt, (int)t->F, s);
    } else {
        if (e == hipSuccess) e = rg::launch_step(sp, (int)t->F, false, 32, s);
        if (nodes == 2) {
            if (e == hipSuccess) e = rg::launch_tick_tail(tt, (int)t->F, s);
        } else {
            if (e == hipSuccess) e = rg::launch_tick_fold(fp, s);
            if (e == hipSuccess && io->send_head) e = rg::launch_replicate(qp, (int)t->F, s);
            if (e == hipSuccess && io->ready) e = rg::launch_ready(rp, 0, io->critical_point, io->cool_down_ms, (uint8_t *)d_ready, s);
        }
    }
    hipGraph_t g = nullptr;
    const hipError_t e2 = hipStreamEndCapture(s, &g);           // (always: an open capture would poison the stream)
    k->graph = g;
    if (e == hipSuccess) e = e2;
    if (e == hipSuccess) e = hipGraphInstantiate(&k->exec, k->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rg_tick2_destroy(k);
        return fail(t, -2, "rg_tick2_create: %s", hipGetErrorString(e));
    }
    t->ticks2.push_back(k);
    *tick = k;
    return 0;
}

int rg_tick2_launch(rg_tick2_t *k)
{
    if (!k || !k->t) return -1;
    rg_table *t = k->t;
    if (k->config_gen != t->config_gen)
        return fail(t, -1, "rg_tick2_launch: the table's options or index bases changed after rg_tick2_create (they are part of the recording): create the tick again");
    if (bind(t)) return -2;
    HIP_TRY(t, hipGraphLaunch(k->exec, t->stream));             // (a launch behind one in flight is ordered after it on the stream: include/raftgpu.h)
    k->in_flight = true;
    return 0;
}

int rg_tick2_wait(rg_tick2_t *k)
{
    if (!k || !k->t) return -1;
    rg_table *t = k->t;
    HIP_TRY(t, hipSetDevice(t->device));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    k->in_flight = false;
    return 0;
}

int rg_submit_wait(rg_table_t *t)
{
    if (!t) return -1;
    if (bind(t, false)) return -2;
    return wait_oldest(t);
}

int rg_submit(rg_table_t *t, const rg_batch_t *in, const rg_outcome_t *out, int memspace)
{
    if (!t) return -1;
    if (int rc = check_batch(t, in, out, memspace == RG_MEM_HOST)) return rc;
    if (in->count == 0) return 0;
    if (bind(t)) return -2;
    const bool sparse = in->gid != nullptr;

    const size_t rows = (size_t)in->rounds * in->count;
    rg::StepParams p = step_params(t, in);

    if (memspace == RG_MEM_DEVICE) {
        p.gid = in->gid; p.head = in->head;
        p.ab = (const I64x2 *)in->ab; p.cd = (const I64x2 *)in->cd; p.hint = (const I64x2 *)in->hint;
        p.entry_terms = in->entry_terms;
        p.reply = out->reply; p.logfx = (I64x2 *)out->logfx; p.persist = out->persist;
        return launch(t, p, sparse);
    }
    if (memspace != RG_MEM_HOST) return fail(t, -1, "rg_submit: unknown memspace %d", memspace);

    hipStream_t s = t->stream;
    if (reserve(t, t->st_head, rows * sizeof(rg_ev_head_t)) || reserve(t, t->st_ab, rows * sizeof(I64x2)) ||
        reserve(t, t->st_cd, rows * sizeof(I64x2)) || reserve(t, t->st_reply, rows * sizeof(rg_reply_t)) ||
        reserve(t, t->st_logfx, rows * sizeof(I64x2)) || reserve(t, t->st_persist, rows * sizeof(rg_persist_t)))
        return -2;
    HIP_TRY(t, hipMemcpyAsync(t->st_head.ptr, in->head, rows * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->st_ab.ptr, in->ab, rows * sizeof(I64x2), hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->st_cd.ptr, in->cd, rows * sizeof(I64x2), hipMemcpyHostToDevice, s));
    p.head = (const rg_ev_head_t *)t->st_head.ptr;
    p.ab = (const I64x2 *)t->st_ab.ptr; p.cd = (const I64x2 *)t->st_cd.ptr;
    if (sparse) {
        if (reserve(t, t->st_gid, in->count * sizeof(uint32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_gid.ptr, in->gid, in->count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        p.gid = (const uint32_t *)t->st_gid.ptr;
    }
    if (in->hint) {
        if (reserve(t, t->st_hint, rows * sizeof(I64x2))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_hint.ptr, in->hint, rows * sizeof(I64x2), hipMemcpyHostToDevice, s));
        p.hint = (const I64x2 *)t->st_hint.ptr;
    }
    if (in->entry_count) {
        if (reserve(t, t->st_terms, in->entry_count * sizeof(int64_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_terms.ptr, in->entry_terms, in->entry_count * sizeof(int64_t), hipMemcpyHostToDevice, s));
        p.entry_terms = (const int64_t *)t->st_terms.ptr;
    }
    p.reply = (rg_reply_t *)t->st_reply.ptr; p.logfx = (I64x2 *)t->st_logfx.ptr; p.persist = (rg_persist_t *)t->st_persist.ptr;
    // conditional outputs: rows the kernel does not write come back zeroed (they are only valid when flagged)
    HIP_TRY(t, hipMemsetAsync(t->st_logfx.ptr, 0, rows * sizeof(I64x2), s));
    HIP_TRY(t, hipMemsetAsync(t->st_persist.ptr, 0, rows * sizeof(rg_persist_t), s));
    if (int rc = launch(t, p, sparse)) return rc;
    HIP_TRY(t, hipMemcpyAsync(out->reply, t->st_reply.ptr, rows * sizeof(rg_reply_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipMemcpyAsync(out->logfx, t->st_logfx.ptr, rows * sizeof(I64x2), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipMemcpyAsync(out->persist, t->st_persist.ptr, rows * sizeof(rg_persist_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

/* compact rows, HBM-resident or host: see include/raftgpu.h */
int rg_submit32(rg_table_t *t, const rg_batch32_t *in, const rg_outcome_t *out, int memspace)
{
    if (!t) return -1;
    if (int rc = compact_ok(t, "rg_submit32")) return rc;
    if (!in || !out) return fail(t, -1, "rg_submit32: NULL batch or outcome");
    if (!in->head || !in->abcd) return fail(t, -1, "rg_submit32: head and abcd are required");
    rg_batch_t wide{};                              // the same shape rules as every other submission (rounds, count, gid list, entry bound)
    wide.rounds = in->rounds; wide.count = in->count; wide.gid = in->gid; wide.head = in->head;
    wide.ab = wide.cd = reinterpret_cast<const rg_ev_pair_t *>(in->abcd);      // presence only: check_batch does not read event fields
    wide.entry_terms = reinterpret_cast<const int64_t *>(in->entry_terms); wide.entry_count = in->entry_count;
    if (int rc = check_batch(t, &wide, out, memspace == RG_MEM_HOST)) return rc;
    if (in->count == 0) return 0;
    if (bind(t)) return -2;
    const bool sparse = in->gid != nullptr;
    const size_t rows = (size_t)in->rounds * in->count;
    rg::StepParams p = step_params(t, &wide);
    if (memspace == RG_MEM_DEVICE) {
        p.gid = in->gid; p.head = in->head; p.abcd32 = (const rg::I32x4 *)in->abcd;
        p.entry_terms32 = in->entry_count ? in->entry_terms : nullptr;
        p.reply = out->reply; p.logfx = (I64x2 *)out->logfx; p.persist = out->persist;
        return launch(t, p, sparse);
    }
    if (memspace != RG_MEM_HOST) return fail(t, -1, "rg_submit32: unknown memspace %d", memspace);
    hipStream_t s = t->stream;
    if (reserve(t, t->st_head, rows * sizeof(rg_ev_head_t)) || reserve(t, t->st_abcd32, rows * sizeof(rg_ev_quad32_t)) ||
        reserve(t, t->st_reply, rows * sizeof(rg_reply_t)) || reserve(t, t->st_logfx, rows * sizeof(I64x2)) ||
        reserve(t, t->st_persist, rows * sizeof(rg_persist_t)))
        return -2;
    HIP_TRY(t, hipMemcpyAsync(t->st_head.ptr, in->head, rows * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->st_abcd32.ptr, in->abcd, rows * sizeof(rg_ev_quad32_t), hipMemcpyHostToDevice, s));
    p.head = (const rg_ev_head_t *)t->st_head.ptr; p.abcd32 = (const rg::I32x4 *)t->st_abcd32.ptr;
    if (sparse) {
        if (reserve(t, t->st_gid, in->count * sizeof(uint32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_gid.ptr, in->gid, in->count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        p.gid = (const uint32_t *)t->st_gid.ptr;
    }
    if (in->entry_count) {
        if (reserve(t, t->st_terms32, in->entry_count * sizeof(int32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_terms32.ptr, in->entry_terms, in->entry_count * sizeof(int32_t), hipMemcpyHostToDevice, s));
        p.entry_terms32 = (const int32_t *)t->st_terms32.ptr;
    }
    p.reply = (rg_reply_t *)t->st_reply.ptr; p.logfx = (I64x2 *)t->st_logfx.ptr; p.persist = (rg_persist_t *)t->st_persist.ptr;
    HIP_TRY(t, hipMemsetAsync(t->st_logfx.ptr, 0, rows * sizeof(I64x2), s));
    HIP_TRY(t, hipMemsetAsync(t->st_persist.ptr, 0, rows * sizeof(rg_persist_t), s));
    if (int rc = launch(t, p, sparse)) return rc;
    HIP_TRY(t, hipMemcpyAsync(out->reply, t->st_reply.ptr, rows * sizeof(rg_reply_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipMemcpyAsync(out->logfx, t->st_logfx.ptr, rows * sizeof(I64x2), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipMemcpyAsync(out->persist, t->st_persist.ptr, rows * sizeof(rg_persist_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

int rg_wide_body_workgroups(rg_table_t *t, uint64_t *count, int reset)
{
    if (!t || !count) return -1;
    if (bind(t)) return -2;
    unsigned long long v = 0;
    HIP_TRY(t, hipMemcpyAsync(&v, t->wide_bodies, sizeof v, hipMemcpyDeviceToHost, t->stream));
    if (reset) HIP_TRY(t, hipMemsetAsync(t->wide_bodies, 0, sizeof v, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    *count = v;
    return 0;
}

/* the index base of the compact formats: see include/raftgpu.h */
int rg_index_base_set(rg_table_t *t, uint32_t first, uint32_t count, const int64_t *base)
{
    if (!t) return -1;
    if (!base) return fail(t, -1, "rg_index_base_set: NULL base");
    if ((uint64_t)first + count > t->G) return fail(t, -1, "rg_index_base_set: groups [%u, %u) of %u", first, first + count, t->G);
    for (uint32_t i = 0; i < count; i++)
        if (base[i] < 0) return fail(t, -1, "rg_index_base_set: base[%u] = %lld is negative", i, (long long)base[i]);
    for (uint32_t i = 0; i < count; i++)
        if (base[i] != 0 && !t->has_bases) { t->has_bases = 1; t->config_gen += 1; }       // (sticky: the kernels read the column from now on; recorded ticks are stale)
    if (count == 0) return 0;
    if (bind(t)) return -2;
    HIP_TRY(t, hipMemcpyAsync(t->dt.ibase + first, base, (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

int rg_index_base_get(rg_table_t *t, uint32_t first, uint32_t count, int64_t *base)
{
    if (!t) return -1;
    if (!base) return fail(t, -1, "rg_index_base_get: NULL base");
    if ((uint64_t)first + count > t->G) return fail(t, -1, "rg_index_base_get: groups [%u, %u) of %u", first, first + count, t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    HIP_TRY(t, hipMemcpyAsync(base, t->dt.ibase + first, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

/* which of a row's fields a, b, c, d (bits 0..3) are log indices (rg_device.hpp: index_fields) */
static uint32_t host_index_fields(uint32_t kind)
{
    switch (kind) {
    case RG_EV_AE_REQ: return 0xAu;
    case RG_EV_AE_ACK: return 0x6u;
    case RG_EV_IS_ACK: case RG_EV_RV_REQ: case RG_EV_PV_REQ: case RG_EV_IS_REQ: return 0x2u;
    case RG_EV_LOG_FLUSH: return 0x1u;
    default: return 0u;
    }
}

/* compact rows in, compact outcome rows out (ABI 4): see include/raftgpu.h */
int rg_submit32c(rg_table_t *t, const rg_batch32_t *in, const rg_outcome32_t *out, int memspace)
{
    if (!t) return -1;
    if (int rc = compact_ok(t, "rg_submit32c")) return rc;
    if (!in || !out) return fail(t, -1, "rg_submit32c: NULL batch or outcome");
    if (!in->head || !in->abcd || !out->row || !out->persist) return fail(t, -1, "rg_submit32c: head, abcd, row and persist are required");
    if (in->gid) return fail(t, -1, "rg_submit32c: dense batches only (gid must be NULL)");
    const int wides = (out->wide.reply != nullptr) + (out->wide.logfx != nullptr) + (out->wide.persist != nullptr);
    if (wides != 0 && wides != 3) return fail(t, -1, "rg_submit32c: the overflow columns come as all three or none");
    rg_batch_t wide{};                              // the same shape rules as every other submission (rounds, count, entry bound)
    wide.rounds = in->rounds; wide.count = in->count; wide.head = in->head;
    wide.ab = wide.cd = reinterpret_cast<const rg_ev_pair_t *>(in->abcd);      // presence only: check_batch does not read event fields
    wide.entry_terms = reinterpret_cast<const int64_t *>(in->entry_terms); wide.entry_count = in->entry_count;
    rg_reply_t dummy_r; rg_logfx_t dummy_l; rg_persist_t dummy_p;
    const rg_outcome_t shape{&dummy_r, &dummy_l, &dummy_p};
    if (int rc = check_batch(t, &wide, &shape, memspace == RG_MEM_HOST)) return rc;
    if (in->count == 0) return 0;
    if (bind(t)) return -2;
    const size_t rows = (size_t)in->rounds * in->count;
    rg::StepParams p = step_params(t, &wide);
    if (memspace == RG_MEM_DEVICE) {
        p.head = in->head; p.abcd32 = (const rg::I32x4 *)in->abcd;
        p.entry_terms32 = in->entry_count ? in->entry_terms : nullptr;
        p.out32 = (rg::I32x4 *)out->row; p.persist32 = (rg::I32x4 *)out->persist;
        p.reply = out->wide.reply; p.logfx = (I64x2 *)out->wide.logfx; p.persist = out->wide.persist;
        return launch(t, p, false);
    }
    if (memspace != RG_MEM_HOST) return fail(t, -1, "rg_submit32c: unknown memspace %d", memspace);
    hipStream_t s = t->stream;
    if (reserve(t, t->st_head, rows * sizeof(rg_ev_head_t)) || reserve(t, t->st_abcd32, rows * sizeof(rg_ev_quad32_t)) ||
        reserve(t, t->st_out32, rows * sizeof(rg_out32_t)) || reserve(t, t->st_persist32, rows * sizeof(rg_persist32_t)))
        return -2;
    if (wides && (reserve(t, t->st_reply, rows * sizeof(rg_reply_t)) || reserve(t, t->st_logfx, rows * sizeof(I64x2)) ||
                  reserve(t, t->st_persist, rows * sizeof(rg_persist_t))))
        return -2;
    HIP_TRY(t, hipMemcpyAsync(t->st_head.ptr, in->head, rows * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->st_abcd32.ptr, in->abcd, rows * sizeof(rg_ev_quad32_t), hipMemcpyHostToDevice, s));
    p.head = (const rg_ev_head_t *)t->st_head.ptr; p.abcd32 = (const rg::I32x4 *)t->st_abcd32.ptr;
    if (in->entry_count) {
        if (reserve(t, t->st_terms32, in->entry_count * sizeof(int32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_terms32.ptr, in->entry_terms, in->entry_count * sizeof(int32_t), hipMemcpyHostToDevice, s));
        p.entry_terms32 = (const int32_t *)t->st_terms32.ptr;
    }
    p.out32 = (rg::I32x4 *)t->st_out32.ptr; p.persist32 = (rg::I32x4 *)t->st_persist32.ptr;
    HIP_TRY(t, hipMemsetAsync(t->st_persist32.ptr, 0, rows * sizeof(rg_persist32_t), s));       // conditional rows come back zeroed, as rg_submit's
    if (wides) {
        p.reply = (rg_reply_t *)t->st_reply.ptr; p.logfx = (I64x2 *)t->st_logfx.ptr; p.persist = (rg_persist_t *)t->st_persist.ptr;
        HIP_TRY(t, hipMemsetAsync(t->st_reply.ptr, 0, rows * sizeof(rg_reply_t), s));
        HIP_TRY(t, hipMemsetAsync(t->st_logfx.ptr, 0, rows * sizeof(I64x2), s));
        HIP_TRY(t, hipMemsetAsync(t->st_persist.ptr, 0, rows * sizeof(rg_persist_t), s));
    }
    if (int rc = launch(t, p, false)) return rc;
    HIP_TRY(t, hipMemcpyAsync(out->row, t->st_out32.ptr, rows * sizeof(rg_out32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipMemcpyAsync(out->persist, t->st_persist32.ptr, rows * sizeof(rg_persist32_t), hipMemcpyDeviceToHost, s));
    if (wides) {
        HIP_TRY(t, hipMemcpyAsync(out->wide.reply, t->st_reply.ptr, rows * sizeof(rg_reply_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(t, hipMemcpyAsync(out->wide.logfx, t->st_logfx.ptr, rows * sizeof(I64x2), hipMemcpyDeviceToHost, s));
        HIP_TRY(t, hipMemcpyAsync(out->wide.persist, t->st_persist.ptr, rows * sizeof(rg_persist_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

/* host-side: rg_out32_t / rg_persist32_t rows -> the wide columns (no device involved) */
int rg_outcome32_unpack(const rg_outcome32_t *in, uint32_t rounds, uint32_t count, uint32_t *role_epoch, const rg_outcome_t *out)
{
    return rg_outcome32_unpack_rel(in, rounds, count, role_epoch, nullptr, out);
}

int rg_outcome32_unpack_rel(const rg_outcome32_t *in, uint32_t rounds, uint32_t count, uint32_t *role_epoch, const int64_t *index_base, const rg_outcome_t *out)
{
    if (!in || !in->row || !in->persist || !role_epoch || !out || !out->reply || !out->logfx || !out->persist) return -1;
    auto absolute = [&](int32_t v, uint32_t i) { return v == 0 ? (int64_t)0 : (int64_t)v + (index_base ? index_base[i] : 0); };
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t i = 0; i < count; i++) {
            const size_t row = (size_t)r * count + i;
            const rg_out32_t c = in->row[row];
            const uint32_t flags = c.flags & ~RG_F_WIDE_VALUES;
            const bool has_lfx = (flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0 || RG_F_STATUS(flags) == RG_NEED_HOST;
            const bool has_from = (flags & (RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0 || RG_F_STATUS(flags) == RG_NEED_HOST;
            const bool has_per = (flags & RG_F_PERSIST) != 0;
            if (c.flags & RG_F_WIDE_VALUES) {
                if (!in->wide.reply || !in->wide.logfx || !in->wide.persist) return -3;
                out->reply[row] = in->wide.reply[row];
                out->logfx[row] = has_lfx ? in->wide.logfx[row] : rg_logfx_t{0, 0};
                out->persist[row] = has_per ? in->wide.persist[row] : rg_persist_t{0, 0, 0};
                if (out->reply[row].flags != flags) return -4;          // the two copies of a row disagree
                role_epoch[i] = out->reply[row].role_epoch;
                continue;
            }
            if (has_per) role_epoch[i] = in->persist[row].role_epoch;
            out->reply[row] = rg_reply_t{(flags & RG_F_REPLIED) ? (int64_t)c.resp_term : 0, flags, role_epoch[i]};
            out->logfx[row] = has_lfx ? rg_logfx_t{absolute(c.commit_index, i), has_from ? absolute(c.log_from, i) : 0} : rg_logfx_t{0, 0};
            out->persist[row] = has_per ? rg_persist_t{(int64_t)in->persist[row].term, in->persist[row].voted_for, in->persist[row].role}
                                        : rg_persist_t{0, 0, 0};
        }
    }
    return 0;
}

/* host-side packer: rg_batch_t -> rg_batch32_t (no device involved) */
int64_t rg_batch32_pack(const rg_batch_t *in, rg_ev_head_t *head, rg_ev_quad32_t *abcd, int32_t *entry_terms)
{
    return rg_batch32_pack_rel(in, nullptr, head, abcd, entry_terms);
}

int64_t rg_batch32_pack_rel(const rg_batch_t *in, const int64_t *index_base, rg_ev_head_t *head, rg_ev_quad32_t *abcd, int32_t *entry_terms)
{
    if (!in || !in->head || !in->ab || !in->cd || !head || !abcd) return -1;
    if (in->hint) return -2;                                    // hints answer RG_NEED_HOST rows: those batches stay wide
    const size_t rows = (size_t)in->rounds * in->count;
    auto fits = [](int64_t v) { return v >= 0 && v < (int64_t)1 << 31; };
    uint64_t out_terms = 0;
    for (size_t r = 0; r < rows; r++) {
        const uint32_t hdr = in->head[r].hdr & ~(RG_HDR_HINT_BIT | RG_HDR_SAME_TERM | (1u << 11)), aux = in->head[r].aux;
        int64_t a = in->ab[r].x, b = in->ab[r].y, c = in->cd[r].x, d = in->cd[r].y;
        if (RG_HDR_KIND(hdr) == RG_EV_NONE) a = b = c = d = 0;      // a row that is not addressed carries nothing: whatever its fields hold does not travel
        if (index_base) {                                       // a log index x of group g travels as x - base[g]; 0 ("none") as 0; anything at or below the base has no image
            const size_t i = r % in->count;
            const int64_t base = index_base[in->gid ? in->gid[i] : i];
            const uint32_t ix = host_index_fields(RG_HDR_KIND(hdr));
            int64_t *f[4] = {&a, &b, &c, &d};
            for (int k = 0; k < 4; k++)
                if (((ix >> k) & 1u) && *f[k] != 0) { if (*f[k] <= base) return -3; *f[k] -= base; }
        }
        if (!fits(a) || !fits(b) || !fits(c) || !fits(d)) return -3;
        abcd[r] = rg_ev_quad32_t{(int32_t)a, (int32_t)b, (int32_t)c, (int32_t)d};
        head[r].hdr = hdr; head[r].aux = aux;
        const uint32_t n = RG_HDR_N(hdr);
        if (RG_HDR_KIND(hdr) != RG_EV_AE_REQ || n == 0) continue;
        if (!in->entry_terms || (uint64_t)aux + n > in->entry_count) { head[r].aux = 0xFFFFFFFFu; continue; }   // unreadable stays unreadable (RG_BAD_EVENT): aux + n exceeds any entry_count
        const int64_t *e = in->entry_terms + aux;
        bool same = true;
        for (uint32_t k = 0; k < n; k++) { if (!fits(e[k])) return -3; same = same && e[k] == e[0]; }
        if (same) {
            head[r].hdr = hdr | RG_HDR_SAME_TERM; head[r].aux = (uint32_t)e[0];
        } else {
            if (!entry_terms) return -4;
            head[r].aux = (uint32_t)out_terms;
            for (uint32_t k = 0; k < n; k++) entry_terms[out_terms + k] = (int32_t)e[k];
            out_terms += n;
        }
    }
    return (int64_t)out_terms;
}

int rg_replicate(rg_table_t *t, uint32_t count, const uint32_t *gid, const uint8_t *heartbeat, const uint16_t *in_flight,
                 rg_send_head_t *head, rg_send_t *send, int memspace)
{
    if (!t) return -1;
    if (!head || !send) return fail(t, -1, "rg_replicate: head and send are required");
    if (gid ? count > t->G : count != t->G) return fail(t, -1, "rg_replicate: %u rows for %u groups", count, t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    rg::ReplicateParams p{};
    p.t = t->dt; p.count = count;
    if (memspace == RG_MEM_DEVICE) {
        p.gid = gid; p.heartbeat = heartbeat; p.in_flight = in_flight; p.head = head; p.send = send;
        HIP_TRY(t, rg::launch_replicate(p, (int)t->F, t->stream));
        return 0;
    }
    if (memspace != RG_MEM_HOST) return fail(t, -1, "rg_replicate: unknown memspace %d", memspace);
    if (gid)
        for (uint32_t i = 0; i < count; i++) {
            if (gid[i] >= t->G) return fail(t, -1, "rg_replicate: gid[%u]=%u out of range", i, gid[i]);
            if (i && gid[i] <= gid[i - 1]) return fail(t, -1, "rg_replicate: gid must be strictly ascending (row %u)", i);
        }
    hipStream_t s = t->stream;
    const size_t F = t->F;
    if (reserve(t, t->st_sh, count * sizeof(rg_send_head_t)) || reserve(t, t->st_ss, count * F * sizeof(rg_send_t))) return -2;
    if (gid) {
        if (reserve(t, t->st_gid, count * sizeof(uint32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_gid.ptr, gid, count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        p.gid = (const uint32_t *)t->st_gid.ptr;
    }
    if (heartbeat) {
        if (reserve(t, t->st_hb, count)) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_hb.ptr, heartbeat, count, hipMemcpyHostToDevice, s));
        p.heartbeat = (const uint8_t *)t->st_hb.ptr;
    }
    if (in_flight) {
        if (reserve(t, t->st_fl, count * F * sizeof(uint16_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_fl.ptr, in_flight, count * F * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        p.in_flight = (const uint16_t *)t->st_fl.ptr;
    }
    p.head = (rg_send_head_t *)t->st_sh.ptr; p.send = (rg_send_t *)t->st_ss.ptr;
    HIP_TRY(t, rg::launch_replicate(p, (int)t->F, s));
    HIP_TRY(t, hipMemcpyAsync(head, t->st_sh.ptr, count * sizeof(rg_send_head_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipMemcpyAsync(send, t->st_ss.ptr, count * F * sizeof(rg_send_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

/* ---- N4: timers --------------------------------------------------------------------------------------- */

int rg_timers_configure(rg_table_t *t, int64_t election_ms, int64_t heartbeat_ms, uint64_t seed)
{
    if (!t) return -1;
    if (election_ms <= 0 || heartbeat_ms <= 0 || election_ms > (INT64_MAX >> 2)) return fail(t, -1, "rg_timers_configure: bad intervals");
    t->election_ms = election_ms; t->heartbeat_ms = heartbeat_ms; t->timer_seed = seed;
    return 0;
}

static rg::TimerParams timer_params(rg_table *t)
{
    rg::TimerParams p{};
    p.deadline = t->timer_deadline; p.epoch = t->timer_epoch; p.ident = t->dt.ident; p.groups = t->G;
    p.election_ms = t->election_ms; p.heartbeat_ms = t->heartbeat_ms; p.seed = t->timer_seed;
    return p;
}

int rg_timers_update(rg_table_t *t, uint32_t rounds, uint32_t count, const uint32_t *gid, const rg_reply_t *reply,
                     const int64_t *now, int memspace)
{
    if (!t) return -1;
    if (!reply || !now || rounds == 0) return fail(t, -1, "rg_timers_update: reply, now and rounds are required");
    if (gid ? (rounds != 1 || count > t->G) : count != t->G) return fail(t, -1, "rg_timers_update: %u rows for %u groups", count, t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    const size_t rows = (size_t)rounds * count;
    const rg_reply_t *d_reply = reply;
    const uint32_t *d_gid = gid;
    if (memspace == RG_MEM_HOST) {
        if (reserve(t, t->st_reply, rows * sizeof(rg_reply_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_reply.ptr, reply, rows * sizeof(rg_reply_t), hipMemcpyHostToDevice, s));
        d_reply = (const rg_reply_t *)t->st_reply.ptr;
        if (gid) {
            for (uint32_t i = 0; i < count; i++)
                if (gid[i] >= t->G || (i && gid[i] <= gid[i - 1])) return fail(t, -1, "rg_timers_update: bad gid list at row %u", i);
            if (reserve(t, t->st_gid, count * sizeof(uint32_t))) return -2;
            HIP_TRY(t, hipMemcpyAsync(t->st_gid.ptr, gid, count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            d_gid = (const uint32_t *)t->st_gid.ptr;
        }
    } else if (memspace != RG_MEM_DEVICE) {
        return fail(t, -1, "rg_timers_update: unknown memspace %d", memspace);
    }
    for (uint32_t r0 = 0; r0 < rounds; r0 += 64) {               // <= 64 timestamps travel in the kernel arguments
        rg::TimerParams p = timer_params(t);
        p.rounds = rounds - r0 < 64 ? rounds - r0 : 64; p.count = count; p.gid = d_gid;
        p.reply = d_reply + (size_t)r0 * count;
        for (uint32_t k = 0; k < p.rounds; k++) p.now[k] = now[r0 + k];
        HIP_TRY(t, rg::launch_timers_update(p, s));
    }
    if (memspace == RG_MEM_HOST) HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

int rg_timers_update32(rg_table_t *t, uint32_t rounds, const rg_out32_t *row, const rg_persist32_t *persist32, const int64_t *now, int memspace)
{
    if (!t) return -1;
    if (!row || !persist32 || !now || rounds == 0) return fail(t, -1, "rg_timers_update32: row, persist32, now and rounds are required");
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    const uint32_t count = t->G;
    const size_t rows = (size_t)rounds * count;
    const rg::I32x4 *d_row = (const rg::I32x4 *)row, *d_per = (const rg::I32x4 *)persist32;
    if (memspace == RG_MEM_HOST) {
        if (reserve(t, t->st_out32, rows * sizeof(rg_out32_t)) || reserve(t, t->st_persist32, rows * sizeof(rg_persist32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_out32.ptr, row, rows * sizeof(rg_out32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(t, hipMemcpyAsync(t->st_persist32.ptr, persist32, rows * sizeof(rg_persist32_t), hipMemcpyHostToDevice, s));
        d_row = (const rg::I32x4 *)t->st_out32.ptr; d_per = (const rg::I32x4 *)t->st_persist32.ptr;
    } else if (memspace != RG_MEM_DEVICE) {
        return fail(t, -1, "rg_timers_update32: unknown memspace %d", memspace);
    }
    for (uint32_t r0 = 0; r0 < rounds; r0 += 64) {               // <= 64 timestamps travel in the kernel arguments
        rg::TimerParams p = timer_params(t);
        p.rounds = rounds - r0 < 64 ? rounds - r0 : 64; p.count = count;
        p.out32 = d_row + (size_t)r0 * count; p.persist32 = d_per + (size_t)r0 * count;
        for (uint32_t k = 0; k < p.rounds; k++) p.now[k] = now[r0 + k];
        HIP_TRY(t, rg::launch_timers_update(p, s));
    }
    if (memspace == RG_MEM_HOST) HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

int rg_timers_arm(rg_table_t *t, int64_t now)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    rg::TimerParams p = timer_params(t);
    p.now[0] = now;
    HIP_TRY(t, rg::launch_timers_arm(p, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

int rg_timers_expired_epochs(rg_table_t *t, int64_t now, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, uint32_t *out_count, int memspace)
{
    if (!t) return -1;
    if (!out_count || (capacity && !out_gid)) return fail(t, -1, "rg_timers_expired: out_gid/out_count are required");
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    uint32_t *d_out = out_gid, *d_ep = out_epoch;
    const size_t cap = capacity ? capacity : 1;
    if (memspace == RG_MEM_HOST) {
        if (reserve(t, t->st_tgid, 2 * cap * sizeof(uint32_t))) return -2;      // gids, then epochs
        d_out = (uint32_t *)t->st_tgid.ptr;
        d_ep = out_epoch ? d_out + cap : nullptr;
    } else if (memspace != RG_MEM_DEVICE) {
        return fail(t, -1, "rg_timers_expired: unknown memspace %d", memspace);
    }
    const uint32_t waves = (t->G + 63) / 64;
    HIP_TRY(t, rg::launch_timers_expired(t->timer_deadline, t->dt.ident, t->G, now, nullptr, t->timer_counts, t->timer_counts + waves, d_out, d_ep, capacity, s));
    uint32_t total = 0;
    HIP_TRY(t, hipMemcpyAsync(&total, t->timer_counts + waves, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    *out_count = total;
    const uint32_t n = total < capacity ? total : capacity;
    if (memspace == RG_MEM_HOST && n) {
        HIP_TRY(t, hipMemcpyAsync(out_gid, d_out, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        if (out_epoch) HIP_TRY(t, hipMemcpyAsync(out_epoch, d_ep, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(t, hipStreamSynchronize(s));
    }
    return 0;
}

int rg_timers_expired(rg_table_t *t, int64_t now, uint32_t *out_gid, uint32_t capacity, uint32_t *out_count, int memspace)
{
    return rg_timers_expired_epochs(t, now, out_gid, nullptr, capacity, out_count, memspace);
}

int rg_timers_read(rg_table_t *t, uint32_t first, uint32_t count, int64_t *deadline)
{
    if (!t || !deadline) return -1;
    if ((uint64_t)first + count > t->G) return fail(t, -1, "rg_timers_read: range exceeds %u groups", t->G);
    if (bind(t)) return -2;
    HIP_TRY(t, hipMemcpyAsync(deadline, t->timer_deadline + first, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

/* ---- N4b: health ------------------------------------------------------------------------------------- */

static rg::HealthParams health_params(rg_table *t)
{
    rg::HealthParams p{};
    p.ok = t->health_ok; p.fail = t->health_fail; p.recent = t->health_recent; p.t = t->dt;
    p.followers = t->F; p.self = t->self;
    return p;
}

int rg_health_update(rg_table_t *t, uint32_t rounds, uint32_t count, const uint32_t *gid, const rg_ev_head_t *head,
                     const rg_reply_t *reply, const int64_t *now, int memspace)
{
    if (!t) return -1;
    if (!head || !reply || !now || rounds == 0) return fail(t, -1, "rg_health_update: head, reply, now and rounds are required");
    if (gid ? (rounds != 1 || count > t->G) : count != t->G) return fail(t, -1, "rg_health_update: %u rows for %u groups", count, t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    const size_t rows = (size_t)rounds * count;
    const rg_ev_head_t *d_head = head; const rg_reply_t *d_reply = reply; const uint32_t *d_gid = gid;
    if (memspace == RG_MEM_HOST) {
        if (reserve(t, t->st_head, rows * sizeof(rg_ev_head_t)) || reserve(t, t->st_reply, rows * sizeof(rg_reply_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_head.ptr, head, rows * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, s));
        HIP_TRY(t, hipMemcpyAsync(t->st_reply.ptr, reply, rows * sizeof(rg_reply_t), hipMemcpyHostToDevice, s));
        d_head = (const rg_ev_head_t *)t->st_head.ptr; d_reply = (const rg_reply_t *)t->st_reply.ptr;
        if (gid) {
            for (uint32_t i = 0; i < count; i++)
                if (gid[i] >= t->G || (i && gid[i] <= gid[i - 1])) return fail(t, -1, "rg_health_update: bad gid list at row %u", i);
            if (reserve(t, t->st_gid, count * sizeof(uint32_t))) return -2;
            HIP_TRY(t, hipMemcpyAsync(t->st_gid.ptr, gid, count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            d_gid = (const uint32_t *)t->st_gid.ptr;
        }
    } else if (memspace != RG_MEM_DEVICE) {
        return fail(t, -1, "rg_health_update: unknown memspace %d", memspace);
    }
    for (uint32_t r0 = 0; r0 < rounds; r0 += 64) {
        rg::HealthParams p = health_params(t);
        p.rounds = rounds - r0 < 64 ? rounds - r0 : 64; p.count = count; p.gid = d_gid;
        p.head = d_head + (size_t)r0 * count; p.reply = d_reply + (size_t)r0 * count;
        for (uint32_t k = 0; k < p.rounds; k++) p.now[k] = now[r0 + k];
        HIP_TRY(t, rg::launch_health_update(p, s));
    }
    if (memspace == RG_MEM_HOST) HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

int rg_health_update32(rg_table_t *t, uint32_t rounds, const rg_ev_head_t *head, const rg_out32_t *row, const int64_t *now, int memspace)
{
    if (!t) return -1;
    if (!head || !row || !now || rounds == 0) return fail(t, -1, "rg_health_update32: head, row, now and rounds are required");
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    const uint32_t count = t->G;
    const size_t rows = (size_t)rounds * count;
    const rg_ev_head_t *d_head = head;
    const rg::I32x4 *d_row = (const rg::I32x4 *)row;
    if (memspace == RG_MEM_HOST) {
        if (reserve(t, t->st_head, rows * sizeof(rg_ev_head_t)) || reserve(t, t->st_out32, rows * sizeof(rg_out32_t))) return -2;
        HIP_TRY(t, hipMemcpyAsync(t->st_head.ptr, head, rows * sizeof(rg_ev_head_t), hipMemcpyHostToDevice, s));
        HIP_TRY(t, hipMemcpyAsync(t->st_out32.ptr, row, rows * sizeof(rg_out32_t), hipMemcpyHostToDevice, s));
        d_head = (const rg_ev_head_t *)t->st_head.ptr; d_row = (const rg::I32x4 *)t->st_out32.ptr;
    } else if (memspace != RG_MEM_DEVICE) {
        return fail(t, -1, "rg_health_update32: unknown memspace %d", memspace);
    }
    for (uint32_t r0 = 0; r0 < rounds; r0 += 64) {
        rg::HealthParams p = health_params(t);
        p.rounds = rounds - r0 < 64 ? rounds - r0 : 64; p.count = count;
        p.head = d_head + (size_t)r0 * count; p.out32 = d_row + (size_t)r0 * count;
        for (uint32_t k = 0; k < p.rounds; k++) p.now[k] = now[r0 + k];
        HIP_TRY(t, rg::launch_health_update(p, s));
    }
    if (memspace == RG_MEM_HOST) HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

int rg_health_failure(rg_table_t *t, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, int64_t now)
{
    if (!t) return -1;
    if (n && (!gid || !slot || !flags)) return fail(t, -1, "rg_health_failure: gid, slot and flags are required");
    if (n == 0) return 0;
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    if (reserve(t, t->st_hgid, n * sizeof(uint32_t)) || reserve(t, t->st_hslot, n) || reserve(t, t->st_hflag, n)) return -2;
    HIP_TRY(t, hipMemcpyAsync(t->st_hgid.ptr, gid, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->st_hslot.ptr, slot, n, hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->st_hflag.ptr, flags, n, hipMemcpyHostToDevice, s));
    rg::HealthParams p = health_params(t);
    p.now[0] = now;
    HIP_TRY(t, rg::launch_health_failure(p, n, (const uint32_t *)t->st_hgid.ptr, (const uint8_t *)t->st_hslot.ptr,
                                         (const uint8_t *)t->st_hflag.ptr, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    return 0;
}

int rg_ready(rg_table_t *t, int64_t now, int32_t critical_point, int64_t cool_down_ms, uint8_t *ready, int memspace)
{
    if (!t || !ready) return -1;
    if (bind(t)) return -2;
    hipStream_t s = t->stream;
    uint8_t *d_ready = ready;
    if (memspace == RG_MEM_HOST) {
        if (reserve(t, t->st_ready, t->G)) return -2;
        d_ready = (uint8_t *)t->st_ready.ptr;
    } else if (memspace != RG_MEM_DEVICE) {
        return fail(t, -1, "rg_ready: unknown memspace %d", memspace);
    }
    HIP_TRY(t, rg::launch_ready(health_params(t), now, critical_point, cool_down_ms, d_ready, s));
    if (memspace == RG_MEM_HOST) {
        HIP_TRY(t, hipMemcpyAsync(ready, d_ready, t->G, hipMemcpyDeviceToHost, s));
        HIP_TRY(t, hipStreamSynchronize(s));
    }
    return 0;
}

int rg_health_read(rg_table_t *t, uint32_t first, uint32_t count, int64_t *request_success, int64_t *request_failure,
                   int32_t *recent_failure)
{
    if (!t || !request_success || !request_failure || !recent_failure) return -1;
    if ((uint64_t)first + count > t->G) return fail(t, -1, "rg_health_read: range exceeds %u groups", t->G);
    if (count == 0) return 0;
    if (bind(t)) return -2;
    const size_t n = count, F = t->F, G = t->G;
    std::vector<int64_t> ok(n * F), fl(n * F);
    std::vector<int32_t> rc(n * F);
    for (size_t j = 0; j < F; j++) {
        HIP_TRY(t, hipMemcpyAsync(ok.data() + j * n, t->health_ok + j * G + first, n * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(t, hipMemcpyAsync(fl.data() + j * n, t->health_fail + j * G + first, n * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(t, hipMemcpyAsync(rc.data() + j * n, t->health_recent + j * G + first, n * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    }
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < F; j++) {
            request_success[i * F + j] = ok[j * n + i]; request_failure[i * F + j] = fl[j * n + i]; recent_failure[i * F + j] = rc[j * n + i];
        }
    return 0;
}

const char *rg_step_kernel(rg_table_t *t, uint32_t count)
{
    if (!t) return "";
    return step_lanes(t, count) == 0 ? "rg::step_split_kernel" : "rg::step_kernel";
}

int rg_sync(rg_table_t *t)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

/* ---- device memory helpers --------------------------------------------------------------------- */

int rg_host_alloc(rg_table_t *t, size_t bytes, void **hptr)
{
    if (!t || !hptr) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}

int rg_host_free(rg_table_t *t, void *hptr)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    HIP_TRY(t, hipHostFree(hptr));
    return 0;
}

int rg_dev_alloc(rg_table_t *t, size_t bytes, void **dptr)
{
    if (!t || !dptr) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}

int rg_dev_free(rg_table_t *t, void *dptr)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    HIP_TRY(t, hipFree(dptr));
    return 0;
}

int rg_copy_to_device(rg_table_t *t, void *dst, const void *src, size_t bytes)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

int rg_copy_to_host(rg_table_t *t, void *dst, const void *src, size_t bytes)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    HIP_TRY(t, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    return 0;
}

void *rg_stream(rg_table_t *t) { return t ? (void *)t->stream : nullptr; }

/* ---- measurement -------------------------------------------------------------------------------- */

int rg_timing_enable(rg_table_t *t, int on)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    if (drain_timing(t)) return -2;
    t->timing = on != 0;
    return 0;
}

int rg_timing_read(rg_table_t *t, uint64_t *launches, double *total_ms, int reset)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    if (drain_timing(t)) return -2;
    if (launches) *launches = t->t_launches;
    if (total_ms) *total_ms = t->t_total_ms;
    if (reset) { t->t_launches = 0; t->t_total_ms = 0.0; }
    return 0;
}

int rg_timing_begin(rg_table_t *t)
{
    if (!t) return -1;
    if (bind(t)) return -2;
    if (!t->region0) { HIP_TRY(t, hipEventCreate(&t->region0)); HIP_TRY(t, hipEventCreate(&t->region1)); }
    HIP_TRY(t, hipEventRecord(t->region0, t->stream));
    return 0;
}

int rg_timing_end(rg_table_t *t, double *elapsed_ms)
{
    if (!t || !elapsed_ms) return -1;
    if (bind(t)) return -2;
    if (!t->region0) return fail(t, -1, "rg_timing_end without rg_timing_begin");
    HIP_TRY(t, hipEventRecord(t->region1, t->stream));
    HIP_TRY(t, hipEventSynchronize(t->region1));
    float ms = 0.f;
    HIP_TRY(t, hipEventElapsedTime(&ms, t->region0, t->region1));
    *elapsed_ms = ms;
    return 0;
}

int rg_counters_read(rg_table_t *t, uint64_t counters[RG_NUM_COUNTERS], int reset)
{
    if (!t || !counters) return -1;
    if (bind(t)) return -2;
    const size_t n = t->counter_slots * RG_NUM_COUNTERS;
    std::vector<unsigned long long> host(n);
    HIP_TRY(t, hipMemcpyAsync(host.data(), t->counters, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, t->stream));
    if (reset) HIP_TRY(t, hipMemsetAsync(t->counters, 0, n * sizeof(unsigned long long), t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
#ifdef RG_PROBE_HWID            // experiment build: the raw slots (wavefront placement and times) go to a file, tools/placement.py reads it
    if (const char *path = getenv("RG_DUMP_COUNTERS")) {
        if (FILE *f = fopen(path, "wb")) { fwrite(host.data(), sizeof(unsigned long long), n, f); fclose(f); }
    }
#endif
    for (int i = 0; i < RG_NUM_COUNTERS; i++) counters[i] = 0;
    for (size_t s = 0; s < t->counter_slots; s++)
        for (int i = 0; i < RG_NUM_COUNTERS; i++) counters[i] += host[s * RG_NUM_COUNTERS + i];
    return 0;
}

int rg_copy_bandwidth(rg_table_t *t, size_t bytes, int iters, double *gbps)
{
    if (!t || !gbps || iters < 1) return -1;
    if (bind(t)) return -2;
    bytes &= ~(size_t)15;
    if (bytes == 0) return fail(t, -1, "rg_copy_bandwidth: bytes must be >= 16");
    void *src = nullptr, *dst = nullptr;
    HIP_TRY(t, hipMalloc(&src, bytes));
    HIP_TRY(t, hipMalloc(&dst, bytes));
    HIP_TRY(t, hipMemsetAsync(src, 1, bytes, t->stream));
    hipEvent_t a, b;
    HIP_TRY(t, hipEventCreate(&a));
    HIP_TRY(t, hipEventCreate(&b));
    HIP_TRY(t, rg::launch_copy(src, dst, bytes, t->stream));      /* warm-up */
    HIP_TRY(t, hipEventRecord(a, t->stream));
    for (int i = 0; i < iters; i++) HIP_TRY(t, rg::launch_copy(src, dst, bytes, t->stream));
    HIP_TRY(t, hipEventRecord(b, t->stream));
    HIP_TRY(t, hipStreamSynchronize(t->stream));
    float ms = 0.f;
    HIP_TRY(t, hipEventElapsedTime(&ms, a, b));
    *gbps = 2.0 * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(src); (void)hipFree(dst);
    return 0;
}

}  // extern "C"
