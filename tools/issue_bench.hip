// issue_bench.hip — what ONE wavefront per SIMD pays per instruction on gfx950 (the deciding wavefront of step32_kernel at 65 536 groups is
// exactly that: DESIGN.md section 6). Each case is an unrolled block of N copies of a small instruction pattern, executed REPS times between two
// s_memtime reads by every wavefront of a 1024 x 64 launch (one wavefront per SIMD of a 256-CU part); printed: shader-clock ticks per
// instruction (median over wavefronts), so patterns can be priced against each other when tier 1 is restructured.
//   hipcc -O3 --offload-arch=gfx950 -o build/issue_bench tools/issue_bench.hip && build/issue_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

#define CASE(name, instrs_per_copy, body)                                                                                  \
    __global__ __launch_bounds__(64) void name(unsigned long long *out, int reps, int seed)                               \
    {                                                                                                                      \
        int v0 = seed + threadIdx.x, v1 = seed * 3 + 1, v2 = seed ^ 5, v3 = 7, v4 = 9, v5 = 11, v6 = 13, v7 = 15;            \
        int s0 = seed, s1 = seed + 2, s2 = 3, s3 = 4;                                                                      \
        unsigned long long m0 = 0, m1 = 0;                                                                                 \
        __shared__ int lds[1024];                                                                                           \
        lds[threadIdx.x] = seed; lds[threadIdx.x + 64] = 1;                                                                \
        int la = (threadIdx.x * 16) & 1023;                                                                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                              \
        for (int r = 0; r < reps; r++) {                                                                                   \
            asm volatile(R64(body)                                                                                         \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), \
                           "+s"(m0), "+s"(m1), "+v"(la)                                                                    \
                         :: "vcc", "memory");                                                                              \
        }                                                                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                              \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                   \
        if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + s1 + s2 + s3 + (int)m0 + (int)m1 + la == 0x7fffffff) out[0] = 0;  \
    }                                                                                                                      \
    static const int name##_n = (instrs_per_copy);

// %0..%7 = v0..v7, %8..%11 = s0..s3, %12/%13 = m0/m1 (64-bit sgpr pairs), %14 = lds address
CASE(valu_dependent, 1, "v_add_u32 %0, %0, %1\n")
CASE(valu_independent4, 4, "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n")
CASE(salu_dependent, 1, "s_add_u32 %8, %8, %9\n")
CASE(salu_independent4, 4, "s_add_u32 %8, %8, %9\n s_add_u32 %10, %10, %9\n s_add_u32 %11, %11, %9\n s_and_b64 %12, %12, %13\n")
CASE(valu_salu_alternating, 2, "v_add_u32 %0, %0, %1\n s_add_u32 %8, %8, %9\n")
CASE(cmp_and_cndmask_chain, 3, "v_cmp_lt_i32 %12, %0, %1\n s_and_b64 %12, %12, %13\n v_cndmask_b32 %0, %0, %2, %12\n")
CASE(cmp_vcc_cndmask_chain, 2, "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n")
CASE(cmp_cmp_and_independent, 3, "v_cmp_lt_i32 %12, %0, %1\n v_cmp_eq_u32 %13, %2, %3\n s_and_b64 vcc, %12, %13\n")
CASE(bitop3_chain, 1, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6\n")
CASE(cmpx_chain, 1, "v_cmpx_le_i32 vcc, %1, %1\n")
CASE(branch_taken_skip0, 1, "s_branch 1f\n1:\n")
CASE(branch_taken_skip8, 1, "s_branch 1f\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n1:\n")
CASE(cbranch_not_taken, 2, "s_cmp_eq_u32 %8, 0x12345\n s_cbranch_scc1 1f\n1:\n")
CASE(saveexec_execz_not_taken, 4, "v_cmp_le_i32 vcc, %1, %1\n s_and_saveexec_b64 %12, vcc\n s_cbranch_execz 1f\n1:\n s_or_b64 exec, exec, %12\n")
CASE(saveexec_execz_taken, 4, "v_cmp_lt_i32 vcc, %1, %1\n s_and_saveexec_b64 %12, vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n1:\n s_or_b64 exec, exec, %12\n")
CASE(ballot_cbranch_vccz_taken, 3, "v_cmp_lt_i32 vcc, %1, %1\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n1:\n v_add_u32 %2, %2, %1\n")
CASE(readlane_salu_writelane, 3, "v_readlane_b32 %8, %0, 5\n s_add_u32 %8, %8, 1\n v_writelane_b32 %0, %8, 7\n")
CASE(readlane_independent, 2, "v_readlane_b32 %8, %0, 5\n v_readlane_b32 %10, %2, 6\n")
CASE(lds_read_dependent, 2, "ds_read_b32 %14, %14\n s_waitcnt lgkmcnt(0)\n")

CASE(lds_write_b128, 1, "ds_write_b32 %14, %0 offset:2048\n")
CASE(min_max_chain, 2, "v_min_i32 %0, %0, %1\n v_max_i32 %0, %0, %2\n")
CASE(sdwa_cmp, 1, "v_cmp_eq_u16_sdwa %12, %0, %1 src0_sel:BYTE_1 src1_sel:DWORD\n")
CASE(s_nop0, 1, "s_nop 0\n")
CASE(waitcnt_only, 1, "s_waitcnt lgkmcnt(0)\n")

struct Case { const char *name; void (*k)(unsigned long long *, int, int); int n; };

int main()
{
    unsigned long long *d;
    const int WG = 1024, REPS = 64;
    hipMalloc(&d, WG * sizeof(*d));
    std::vector<unsigned long long> h(WG);
#define C(x) Case{#x, x, x##_n}
    const Case cases[] = {C(valu_dependent), C(valu_independent4), C(salu_dependent), C(salu_independent4), C(valu_salu_alternating), C(cmp_and_cndmask_chain),
                          C(cmp_vcc_cndmask_chain), C(cmp_cmp_and_independent), C(bitop3_chain), C(cmpx_chain), C(branch_taken_skip0), C(branch_taken_skip8), C(cbranch_not_taken),
                          C(saveexec_execz_not_taken), C(saveexec_execz_taken), C(ballot_cbranch_vccz_taken), C(readlane_salu_writelane), C(readlane_independent),
                          C(lds_read_dependent), C(lds_write_b128), C(min_max_chain), C(sdwa_cmp), C(s_nop0), C(waitcnt_only)};
    printf("%-30s %10s %12s   (ticks of s_memtime; 64 copies x %d reps per wavefront; 1024 wavefronts = one per SIMD)\n", "case", "instr/copy", "ticks/instr", REPS);
    for (const Case &c : cases) {
        for (int pass = 0; pass < 2; pass++) hipLaunchKernelGGL(c.k, dim3(WG), dim3(64), 0, 0, d, REPS, 3);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, WG * sizeof(*d), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per = (double)h[WG / 2] / (64.0 * REPS);
        printf("%-30s %10d %12.2f   per copy %.1f  (min wave %.2f, max %.2f per instr)\n", c.name, c.n, per / c.n, per, (double)h[0] / (64.0 * REPS) / c.n, (double)h[WG - 1] / (64.0 * REPS) / c.n);
    }
    return 0;
}
