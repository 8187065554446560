// tests/devemu/hip/hip_runtime.h — TEST INFRASTRUCTURE, not a product path.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile rafting_amd/csrc/{rg_kernels.hip, raftgpu.cpp} UNCHANGED
// into tests/devemu/libraftgpu_emu.so, so that `pytest -m "not gpu"` can run the real device decision code
// (rg_device.hpp: Stepper::try_fast / run and everything under them) and the real host side of the C-ABI against the
// oracle on a machine without a GPU. "Device memory" is the heap, a kernel launch runs every (block, thread) of the
// grid one after the other on the calling thread.
//
// Two ways to run a grid (emu_runtime.cpp):
//   lane-serial (default): every (block, thread) one after the other on the calling thread. Fast, and exact for every
//     kernel whose lanes are independent — step_kernel (one lane = one raft group), the timers
//     update/arm kernels, the health kernels, copy_kernel. It cannot reproduce what needs lanes to MEET: wavefront
//     shuffles/ballots (the decision counters come out per lane, not summed; rg_timers_expired's ballot compaction is
//     wrong) and barriers (step_split_kernel is refused, not dead-locked). Tests on it force RG_SPLIT=0 and do not look
//     at counters or expired-timer lists. replicate_kernel transposes its outputs through LDS behind __syncthreads, so it
//     always runs the second way.
//   wavefronts (RG_EMU_WAVES=1): every lane of a workgroup is an OS thread; shuffles, ballots, readfirstlane meet per
//     64-lane wavefront, __syncthreads / s_barrier per workgroup, a lane that returns stops being waited for. Slow, but
//     it runs ALL kernels, including the two-wavefront step kernel with its LDS hand-over and the ballot compaction.
//     (It checks the protocol, not the hardware: LDS visibility and waitcnt placement are the GPU tests' business.)
// Nothing in rafting_amd/ knows about this file; libraftgpu.so itself has no CPU path and fails without a HIP device.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define __device__
#define __host__
#define __global__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };

#include <functional>
#include <vector>
namespace hipemu {
// stream capture (hipStreamBeginCapture .. hipGraphLaunch): while a capture is open, copies, memsets and launches are recorded as closures instead
// of being executed; hipGraphLaunch replays them in order. One stream is as good as another here (everything is synchronous).
extern thread_local std::vector<std::function<void()>> *capture_;
extern thread_local dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
struct Deadlock {};
void pinned_add(void *p, size_t n);
void pinned_remove(void *p);
bool pinned_has(const void *p);
void lane_yield();
unsigned long long wave_ballot(bool p);                    // lane-serial: a wavefront of one lane
unsigned long long wave_exchange_xor(unsigned long long bits, int lane_xor);   // lane-serial: no partner, 0
unsigned long long wave_first(unsigned long long v);
unsigned long long wave_lane_value(unsigned long long v, int lane);   // v_readlane: lane-serial: the lane's own value (it is lane 0 of its wavefront of one)
int wave_lane_index();                                      // 0 on a lane-serial grid
void workgroup_barrier(bool required);                      // required (s_barrier) throws Deadlock on a lane-serial grid
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body, const char *kernel);
}
extern "C" void rg_emu_note_fallback();                    // step32_kernel: a workgroup left the 32-bit domain (emu_runtime.cpp counts them)
#define RG_NOTE_FALLBACK() rg_emu_note_fallback()
extern "C" void rg_emu_note_slow(int slow_row, int wave_round);   // step32_kernel's deciding wavefront, per lane and round
#define RG_NOTE_SLOW(slow, first_lane) do { const unsigned long long any_ = ::hipemu::wave_ballot(slow); /* every lane meets */ \
                                            rg_emu_note_slow((slow) ? 1 : 0, ((first_lane) && any_ != 0) ? 1 : 0); } while (0)
#define threadIdx (::hipemu::threadIdx_)
#define blockIdx (::hipemu::blockIdx_)
#define blockDim (::hipemu::blockDim_)
#define gridDim (::hipemu::gridDim_)

// ---- device intrinsics the kernels use ---------------------------------------------------------------------------
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_memtime() (0ull)
#define __builtin_amdgcn_readfirstlane(x) ((decltype(x))::hipemu::wave_first((unsigned long long)(x)))
#define __builtin_amdgcn_ballot_w64(x) (::hipemu::wave_ballot(x))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define RG_GLOBAL_AS                 /* rg_step.hpp: global-memory pointers made from integers */
#define RG_OWN_SGPRS(v) ((void)0)
#define RG_FRESH_VGPR(v) ((void)0)
#define RG_AGENT_LOAD(p) (*(p))      /* rg_kernels.hip: tick_fold_kernel's cross-workgroup reads (workgroups run one after the other here) */
#define RG_AGENT_STORE(p, v) (*(p) = (v))
#define __builtin_amdgcn_s_sleep(x) (::hipemu::lane_yield())
#define __builtin_amdgcn_s_barrier() (::hipemu::workgroup_barrier(true))
#define __builtin_amdgcn_wave_barrier() ((void)::hipemu::wave_ballot(true))      /* the lanes of a wavefront meet */
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))      /* g++ has no such builtin; vectors via ext_vector_type are clang-only: */
#define __syncthreads() (::hipemu::workgroup_barrier(false))
#define __ballot(x) (::hipemu::wave_ballot(x))
#define __popcll(x) __builtin_popcountll(x)
template <class T> static inline T __shfl_xor(T v, int lane_xor, int) { return (T)::hipemu::wave_exchange_xor((unsigned long long)v, lane_xor); }
template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---- the slice of the runtime API raftgpu.cpp uses -----------------------------------------------------------------
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801 } hipError_t;
typedef enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event { double t_ms; } *hipEvent_t;
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };

namespace hipemu { extern thread_local hipError_t last_error; }

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { hipError_t e = ::hipemu::last_error; ::hipemu::last_error = hipSuccess; return e; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 256; strcpy(p->name, "lane-serial host emulation"); return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { const hipError_t e = hipMalloc(p, n); if (e == hipSuccess) ::hipemu::pinned_add(*p, n ? n : 1); return e; }
static inline hipError_t hipHostFree(void *p) { ::hipemu::pinned_remove(p); free(p); return hipSuccess; }
// page-locked memory is mapped at its own address; anything else is not device-visible (as on the real runtime)
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { if (!::hipemu::pinned_has(h)) return hipErrorInvalidValue; *d = h; return hipSuccess; }
// what kind of memory a pointer is: page-locked ranges are known; everything else on the heap stands for device memory here
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; void *devicePointer; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
    a->type = ::hipemu::pinned_has(p) ? hipMemoryTypeHost : hipMemoryTypeDevice; a->devicePointer = const_cast<void *>(p);
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
    if (::hipemu::capture_) { ::hipemu::capture_->push_back([=]() { memmove(d, s, n); }); return hipSuccess; }
    memmove(d, s, n); return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t)
{
    if (::hipemu::capture_) { ::hipemu::capture_->push_back([=]() { memset(d, v, n); }); return hipSuccess; }
    memset(d, v, n); return hipSuccess;
}
typedef struct hipemu_graph { std::vector<std::function<void()>> ops; } *hipGraph_t, *hipGraphExec_t;
typedef void *hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { if (::hipemu::capture_) return hipErrorInvalidValue; ::hipemu::capture_ = new std::vector<std::function<void()>>(); return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g)
{
    if (!::hipemu::capture_) return hipErrorInvalidValue;
    *g = new hipemu_graph{std::move(*::hipemu::capture_)};
    delete ::hipemu::capture_; ::hipemu::capture_ = nullptr;
    return hipSuccess;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, hipGraphNode_t *, char *, size_t) { *e = new hipemu_graph{g->ops}; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto &op : e->ops) op(); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t_ms = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) do { \
        if (::hipemu::capture_) { const dim3 g_ = (grid), b_ = (block); ::hipemu::capture_->push_back([=]() { ::hipemu::run_grid(g_, b_, [=]() { kernel(__VA_ARGS__); }, #kernel); }); } \
        else ::hipemu::run_grid((grid), (block), [&]() { kernel(__VA_ARGS__); }, #kernel); } while (0)
