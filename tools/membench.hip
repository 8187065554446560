// tools/membench.hip — standalone HBM micro-benchmarks behind DESIGN.md's "memory floor" numbers (not product code).
//   hipcc -O3 --offload-arch=gfx950 -o build/membench tools/membench.hip && build/membench
// (1) plain copies in several shapes: what does a streaming kernel reach on this device, and with which access shape;
// (2) "row streams": the step kernel's traffic WITHOUT its decisions — one wavefront per 64 groups walks R rounds of a
//     [round][group] batch, K rounds ahead, and writes a reply row per event (plus a log-effect row for a share of them).
//     wide  = 8 + 16 + 16 B per row in three arrays + four gathered 8-byte entry terms (the rg_batch_t layout)
//     compact = 8 + 16 B per row in two arrays (rg_batch32_t with the shared entry term in the row)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void copy_v0(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_unroll(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// every workgroup walks its own contiguous slice, U x 4 KiB at a time
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(src + i + k * 256) : src[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * 256); else dst[i + k * 256] = v[k]; }
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}
template <int U>
__global__ __launch_bounds__(256) void read_only(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int k = 0; k < U; k++) acc ^= __builtin_nontemporal_load(src + i + k * stride);
    }
    if (acc.x == 0x12345u) dst[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write_only(u32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32x4 v = {1, 2, 3, (uint32_t)i};
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int k = 0; k < U; k++) __builtin_nontemporal_store(v, dst + i + k * stride);
    }
}

// ---- row streams -------------------------------------------------------------------------------------------------------
struct RowBufs {
    const u32x2 *head; const u32x4 *ab, *cd; const uint64_t *terms;   // inputs
    u32x4 *reply, *logfx;                                             // outputs
    uint32_t count, rounds;
};
// MODE 0: wide rows (8+16+16 + 4 gathered terms), 1: compact rows (8+16). NT: non-temporal loads and stores. LOGFX_PCT: share of rows writing a log-effect row
template <int MODE, bool NT, int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rows(const RowBufs b, uint32_t logfx_mask)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64u + lane;
    if (i >= b.count) return;
    // WAVES > 1: the waves of a workgroup share the 64 groups and take the rounds round-robin
    u32x2 h[D]; u32x4 q[D], q2[D]; uint64_t e[D][4];
    auto issue = [&](int k, uint32_t r) {
        const size_t row = (size_t)(r < b.rounds ? r : b.rounds - 1) * b.count + i;
        h[k] = NT ? __builtin_nontemporal_load(b.head + row) : b.head[row];
        q[k] = NT ? __builtin_nontemporal_load(b.ab + row) : b.ab[row];
        if (MODE == 0) q2[k] = NT ? __builtin_nontemporal_load(b.cd + row) : b.cd[row];
    };
    auto tails = [&](int k) {
        if (MODE == 0) {
            const uint32_t o = h[k].y;
#pragma unroll
            for (int t = 0; t < 4; t++) e[k][t] = b.terms[o + (uint32_t)t];
        }
    };
    uint32_t r0 = wave;
#pragma unroll
    for (int k = 0; k < D; k++) issue(k, r0 + k * WAVES);
    uint32_t acc = 0;
    for (uint32_t r = r0; r < b.rounds; r += D * WAVES) {
#pragma unroll
        for (int k = 0; k < D; k++) {
            const uint32_t rr = r + k * WAVES;
            if (MODE == 0) tails(k);
            u32x4 v = q[k];
            if (MODE == 0) { v ^= q2[k]; v.x ^= (uint32_t)(e[k][0] ^ e[k][1] ^ e[k][2] ^ e[k][3]); }
            v.y ^= h[k].x; acc += v.x;
            const u32x2 hh = h[k];
            issue(k, rr + D * WAVES);
            if (rr < b.rounds) {
                const size_t row = (size_t)rr * b.count + i;
                if (NT) __builtin_nontemporal_store(v, b.reply + row); else b.reply[row] = v;
                if ((hh.x & logfx_mask) != 0u) { if (NT) __builtin_nontemporal_store(v, b.logfx + row); else b.logfx[row] = v; }
            }
        }
    }
    if (acc == 0x7654321u) b.reply[0] = u32x4{acc, 0, 0, 0};
}


#include <functional>
static double time_ms(hipStream_t s, int iters, const std::function<void()> &f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char **argv)
{
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t bytes = 1ull << 30, n = bytes / 16;
    u32x4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 2, bytes));
    auto rep = [&](const char *name, double ms, double b) { printf("%-44s %8.4f ms  %7.1f GB/s\n", name, ms, b / ms / 1e6); fflush(stdout); };
    rep("copy v0 grid-stride 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL(copy_v0, dim3(2048), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy unroll4 plain 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<4, false>), dim3(2048), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy unroll4 nt 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<4, true>), dim3(2048), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy unroll8 nt 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<8, true>), dim3(2048), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy unroll4 nt 4096x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<4, true>), dim3(4096), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy unroll4 nt 1024x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<4, true>), dim3(1024), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy unroll2 nt 8192x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<2, true>), dim3(8192), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy chunk4 nt 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(2048), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy chunk8 plain 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_chunk<8, false>), dim3(2048), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("copy one pass n/(256*4) blocks, nt", time_ms(s, 10, [&] { hipLaunchKernelGGL((copy_unroll<4, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, s, src, dst, n); }), 2.0 * bytes);
    rep("read-only unroll8 nt 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((read_only<8>), dim3(2048), dim3(256), 0, s, src, dst, n); }), 1.0 * bytes);
    rep("write-only unroll8 nt 2048x256", time_ms(s, 10, [&] { hipLaunchKernelGGL((write_only<8>), dim3(2048), dim3(256), 0, s, dst, n); }), 1.0 * bytes);
    CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    rep("hipMemcpyAsync D2D", time_ms(s, 10, [&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); }), 2.0 * bytes);

    // ---- row streams: 65 536 and 131 072 groups x 64 rounds, fresh buffers per timed launch (4 sets, > the 256 MB Infinity Cache together)
    for (uint32_t groups : {65536u, 131072u}) {
        const uint32_t rounds = 64; const size_t rowsn = (size_t)groups * rounds;
        const int SETS = 4;
        std::vector<RowBufs> sets(SETS);
        std::vector<uint32_t> hh(rowsn * 2);
        uint64_t x = 88172645463325252ull; uint32_t off = 0;
        for (size_t r = 0; r < rowsn; r++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hh[2 * r] = (uint32_t)x; hh[2 * r + 1] = off; off += (uint32_t)(x >> 40) % 4u; }
        const size_t nterms = (size_t)off + 8;
        for (int k = 0; k < SETS; k++) {
            u32x2 *head; u32x4 *ab, *cd, *reply, *logfx; uint64_t *terms;
            CK(hipMalloc(&head, rowsn * 8)); CK(hipMalloc(&ab, rowsn * 16)); CK(hipMalloc(&cd, rowsn * 16)); CK(hipMalloc(&terms, nterms * 8));
            CK(hipMalloc(&reply, rowsn * 16)); CK(hipMalloc(&logfx, rowsn * 16));
            CK(hipMemcpy(head, hh.data(), rowsn * 8, hipMemcpyHostToDevice));
            CK(hipMemset(ab, 3, rowsn * 16)); CK(hipMemset(cd, 5, rowsn * 16)); CK(hipMemset(terms, 7, nterms * 8));
            sets[k] = RowBufs{head, ab, cd, terms, reply, logfx, groups, rounds};
        }
        const double logfx_share = 0.5;                     // one header bit set: half of the rows write a log-effect row
        const double wide_b = rowsn * (40.0 + 8.0 * 1.5 + 16.0 + 16.0 * logfx_share), comp_b = rowsn * (24.0 + 16.0 + 16.0 * logfx_share);
        int cur = 0;
        char name[128];
#define ROWS(MODE, NT, D, W) do { \
            const double ms = time_ms(s, 12, [&] { hipLaunchKernelGGL((rows<MODE, NT, D, W>), dim3((groups + 63) / 64), dim3(64 * W), 0, s, sets[cur++ % SETS], 1u); }); \
            snprintf(name, sizeof name, "rows %s G=%u D=%d waves=%d %s", MODE ? "compact" : "wide", groups, D, W, NT ? "nt" : "plain"); \
            rep(name, ms, MODE ? comp_b : wide_b); } while (0)
        ROWS(0, false, 4, 1); ROWS(0, true, 4, 1); ROWS(0, true, 8, 1);
        ROWS(1, false, 4, 1); ROWS(1, true, 4, 1); ROWS(1, true, 8, 1); ROWS(1, true, 2, 1);
        ROWS(1, true, 4, 2); ROWS(1, true, 2, 2); ROWS(0, true, 4, 2);
        for (auto &b : sets) { hipFree((void *)b.head); hipFree((void *)b.ab); hipFree((void *)b.cd); hipFree((void *)b.terms); hipFree(b.reply); hipFree(b.logfx); }
    }
    return 0;
}
