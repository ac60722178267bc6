// Dev-only microbenchmarks (not part of the product): Philox4x32-10 issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../insilicoseq_amd/csrc/iss_kernels.hip.h"

template <int ROUNDS_N>
__global__ __launch_bounds__(256) void k_philox(uint32_t *out, int n_calls, uint32_t k0, uint32_t k1) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < n_calls; ++i) {
        iss::u32x4 w = iss::philox4x32_10(t, (uint32_t)i, 3u << 24, 0, k0, k1);
        acc ^= w.x ^ w.y ^ w.z ^ w.w;
    }
    out[t] = acc;
}

int main() {
    const int blocks = 256 * 8 * 4, threads = 256, calls = 256;
    uint32_t *d;
    hipMalloc(&d, sizeof(uint32_t) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_philox<10>, dim3(blocks), dim3(threads), 0, 0, d, calls, 42u, 7u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double n = (double)blocks * threads * calls;
        printf("philox4x32-10: %.3f ms, %.2f Gcalls/s, %.1f cycles/wave-call/SIMD (at 2.4GHz, 1024 SIMDs)\n", ms,
               n / ms / 1e6, (ms * 1e-3 * 2.4e9 * 1024) / (n / 64));
    }
    return 0;
}
