// pcie_duplex.hip — what the host link gives: H2D alone, D2H alone, and both at once on two streams (page-locked memory).
// hipcc --offload-arch=gfx950 -O2 -o build/pcie_duplex tools/pcie_duplex.hip && build/pcie_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 256u << 20;
    void *h1, *h2, *d1, *d2;
    CK(hipHostMalloc(&h1, n, hipHostMallocDefault)); CK(hipHostMalloc(&h2, n, hipHostMallocDefault));
    CK(hipMalloc(&d1, n)); CK(hipMalloc(&d2, n));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, a));
        CK(hipStreamSynchronize(a));
        double t1 = now();
        for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, b));
        CK(hipStreamSynchronize(b));
        double t2 = now();
        for (int i = 0; i < 4; i++) { CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, a)); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, b)); }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        double t3 = now();
        if (rep) printf("{\"h2d_gbps\": %.1f, \"d2h_gbps\": %.1f, \"both_at_once_gbps_total\": %.1f}\n", 4.0 * n / (t1 - t0) / 1e9, 4.0 * n / (t2 - t1) / 1e9,
                        8.0 * n / (t3 - t2) / 1e9);
    }
    return 0;
}
