// tests/devemu/emu_runtime.cpp — the two grid executors of the host emulation (see hip/hip_runtime.h). TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

// The marker rafting_amd/engine.py looks for: a library that exports it is refused unless the caller says, through
// RG_ALLOW_HOST_EMULATION=1 (only tests/test_devemu_cpu.py does), that it knows it is not talking to a GPU.
extern "C" int rg_is_host_emulation() { return 1; }
// test hook: workgroups of step32_kernel that fell back to the 64-bit body since the last read
static std::atomic<long> g_fallbacks{0};
extern "C" void rg_emu_note_fallback() { g_fallbacks.fetch_add(1); }
static std::atomic<long> g_slow_rows{0}, g_slow_wave_rounds{0}, g_lane_rounds{0};
extern "C" void rg_emu_note_slow(int slow_row, int wave_round) { g_lane_rounds.fetch_add(1); g_slow_rows.fetch_add(slow_row); g_slow_wave_rounds.fetch_add(wave_round); }
extern "C" void rg_emu_slow_counts(long *lane_rounds, long *slow_rows, long *slow_wave_rounds)
{
    *lane_rounds = g_lane_rounds.exchange(0); *slow_rows = g_slow_rows.exchange(0); *slow_wave_rounds = g_slow_wave_rounds.exchange(0);
}
extern "C" long rg_emu_fallbacks(int reset) { return reset ? g_fallbacks.exchange(0) : g_fallbacks.load(); }

namespace hipemu {

thread_local dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
thread_local std::vector<std::function<void()>> *capture_ = nullptr;
thread_local hipError_t last_error = hipSuccess;

namespace {

// A set of lanes that meet: a generation completes when every lane that is still alive has arrived.
struct Meeting {
    std::mutex m;
    std::condition_variable cv;
    int alive = 0, arrived = 0;
    unsigned long long gen = 0;
    std::vector<unsigned long long> slot, out;
    std::vector<char> here, out_here;

    void init(int n)
    {
        alive = n; arrived = 0; gen = 0;
        slot.assign(n, 0); out.assign(n, 0); here.assign(n, 0); out_here.assign(n, 0);
    }
    void complete()
    {
        out = slot; out_here = here;
        std::fill(here.begin(), here.end(), 0);
        arrived = 0; gen++;
        cv.notify_all();
    }
    // contribute v as member idx, wait for the others, then read the completed generation under the lock
    template <class Reader>
    unsigned long long meet(int idx, unsigned long long v, Reader read)
    {
        std::unique_lock<std::mutex> lk(m);
        slot[idx] = v; here[idx] = 1; arrived++;
        if (arrived == alive) {
            complete();
        } else {
            const unsigned long long g = gen;
            if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return gen != g; })) throw Deadlock();   // a lane never came
        }
        return read(out, out_here);
    }
    void leave()
    {
        std::unique_lock<std::mutex> lk(m);
        alive--;
        if (alive > 0 && arrived == alive) complete();
    }
};

thread_local Meeting *my_wave = nullptr, *my_block = nullptr;
thread_local int my_lane = 0;

bool waves_mode()
{
    static const bool on = [] { const char *e = getenv("RG_EMU_WAVES"); return e && atoi(e) != 0; }();
    return on;
}

}  // namespace

static std::mutex pinned_m;
static std::map<uintptr_t, size_t> pinned_ranges;
void pinned_add(void *p, size_t n) { std::lock_guard<std::mutex> l(pinned_m); pinned_ranges[(uintptr_t)p] = n; }
void pinned_remove(void *p) { std::lock_guard<std::mutex> l(pinned_m); pinned_ranges.erase((uintptr_t)p); }
bool pinned_has(const void *p)
{
    std::lock_guard<std::mutex> l(pinned_m);
    auto it = pinned_ranges.upper_bound((uintptr_t)p);
    if (it == pinned_ranges.begin()) return false;
    --it;
    return (uintptr_t)p < it->first + it->second;
}

void lane_yield() { std::this_thread::yield(); }

unsigned long long wave_ballot(bool p)
{
    if (!my_wave) return p ? 1ull : 0ull;
    return my_wave->meet(my_lane, p ? 1ull : 0ull, [](const std::vector<unsigned long long> &out, const std::vector<char> &here) {
        unsigned long long mask = 0;
        for (size_t i = 0; i < out.size(); i++) if (here[i] && out[i]) mask |= 1ull << i;
        return mask;
    });
}

unsigned long long wave_exchange_xor(unsigned long long bits, int lane_xor)
{
    if (!my_wave) return 0ull;
    const int partner = my_lane ^ lane_xor;
    return my_wave->meet(my_lane, bits, [partner](const std::vector<unsigned long long> &out, const std::vector<char> &here) {
        return partner >= 0 && (size_t)partner < out.size() && here[partner] ? out[partner] : 0ull;
    });
}

unsigned long long wave_first(unsigned long long v)
{
    if (!my_wave) return v;
    return my_wave->meet(my_lane, v, [](const std::vector<unsigned long long> &out, const std::vector<char> &here) {
        for (size_t i = 0; i < out.size(); i++) if (here[i]) return out[i];
        return 0ull;
    });
}

unsigned long long wave_lane_value(unsigned long long v, int lane)
{
    if (!my_wave) return v;
    return my_wave->meet(my_lane, v, [lane](const std::vector<unsigned long long> &out, const std::vector<char> &here) {
        return lane >= 0 && (size_t)lane < out.size() && here[lane] ? out[lane] : 0ull;
    });
}

int wave_lane_index() { return my_wave ? my_lane : 0; }

void workgroup_barrier(bool required)
{
    if (!my_block) {
        if (required) throw Deadlock();            // a lane-serial grid cannot pass an s_barrier
        return;                                    // __syncthreads on a lane-serial grid: a no-op (documented as inexact)
    }
    my_block->meet((int)threadIdx_.x, 0ull, [](const std::vector<unsigned long long> &, const std::vector<char> &) { return 0ull; });
}

// kernels whose lanes exchange data through LDS behind a __syncthreads (not a required s_barrier): a lane-serial grid would
// silently compute garbage for them, so they always run with one OS thread per lane
static bool lanes_must_meet(const char *kernel)
{
    if (!kernel) return false;
    for (const char *name : {"replicate_kernel", "outcome_count_kernel", "outcome_emit_kernel", "timers_scan_kernel"})   // (the last: LDS prefix sum)
        if (strstr(kernel, name)) return true;
    return false;
}

// One OS thread per lane of a workgroup, kept across launches (creating them per launch costs more than running the kernel).
struct LanePool {
    std::mutex busy;                                   // one grid at a time
    std::mutex m;
    std::condition_variable start, finished;
    std::vector<std::thread> workers;
    unsigned long long generation = 0;
    unsigned want = 0, done = 0;
    const std::function<void(unsigned)> *job = nullptr;

    void worker(unsigned idx)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *fn = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                start.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (idx < want) fn = job;
            }
            if (!fn) continue;
            (*fn)(idx);
            std::lock_guard<std::mutex> lk(m);
            if (++done == want) finished.notify_all();
        }
    }
    void run(unsigned n, const std::function<void(unsigned)> &fn)
    {
        std::unique_lock<std::mutex> lk(m);
        while (workers.size() < n) { const unsigned idx = (unsigned)workers.size(); workers.emplace_back([this, idx] { worker(idx); }); workers.back().detach(); }
        job = &fn; want = n; done = 0; generation++;
        start.notify_all();
        finished.wait(lk, [&] { return done == want; });
        job = nullptr;
    }
};
static LanePool &lane_pool() { static LanePool *p = new LanePool; return *p; }      // never destroyed: its threads live until the process ends

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body, const char *kernel)
{
    // (copy_kernel: a grid-stride copy, 2 048 x 256 lanes that never meet — one OS thread per lane would only make it slow)
    if ((!waves_mode() && !lanes_must_meet(kernel)) || (kernel && strstr(kernel, "copy_kernel"))) {
        // `__shared__` is `static` here (hip/hip_runtime.h): ONE copy for the process, so grids launched from several host threads — the feeder
        // threads of MultiDeviceManager — must not overlap any more than two workgroups of one grid do (found as a one-in-ten flake of the
        // ThreadSanitizer case: a leader's matchIndex row in "LDS" overwritten by another table's launch, a commit flag more or less)
        std::lock_guard<std::mutex> one_grid(lane_pool().busy);
        gridDim_ = grid; blockDim_ = block;
        try {
            for (unsigned b = 0; b < grid.x; b++)
                for (unsigned t = 0; t < block.x; t++) {
                    blockIdx_ = dim3(b, 0, 0); threadIdx_ = dim3(t, 0, 0);
                    body();
                }
        } catch (const Deadlock &) {
            last_error = hipErrorNotSupported;      // a kernel that needs its lanes to meet at a barrier
        }
        return;
    }
    const unsigned waves = (block.x + 63u) / 64u;
    std::mutex err_m;
    bool failed = false;
    std::lock_guard<std::mutex> one_grid(lane_pool().busy);            // launches from several host threads take turns
    for (unsigned b = 0; b < grid.x; b++) {
        Meeting wg;
        std::vector<Meeting> wv(waves);
        wg.init((int)block.x);
        for (unsigned w = 0; w < waves; w++) wv[w].init((int)std::min(64u, block.x - w * 64u));
        lane_pool().run(block.x, [&, b](unsigned t) {
            gridDim_ = grid; blockDim_ = block; blockIdx_ = dim3(b, 0, 0); threadIdx_ = dim3(t, 0, 0);
            my_block = &wg; my_wave = &wv[t / 64u]; my_lane = (int)(t % 64u);
            try {
                body();
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_m);
                failed = true;
            }
            my_wave->leave(); my_block->leave();          // a lane that has returned is not waited for any more
            my_wave = nullptr; my_block = nullptr;
        });
        if (failed) break;
    }
    if (failed) last_error = hipErrorNotSupported;
}

}  // namespace hipemu
