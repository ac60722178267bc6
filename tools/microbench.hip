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
        iss::u32x4 w = iss::philox4x32<ROUNDS_N>(t, (uint32_t)i, 3u << 24, 0, k0, k1);
        acc ^= w.x ^ w.y ^ w.z ^ w.w;
    }
    out[t] = acc;
}

// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in k_main's access patterns:
// one dword store per lane, contiguous across lanes (k_main's output stores), and one dwordx2 load
// per lane (its genome window loads).
__global__ __launch_bounds__(256) void k_fill_dword(uint32_t *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_read_dwordx2(const uint2 *in, uint32_t *out, size_t n) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = in[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    {
        const size_t n = (size_t)1 << 28;  // 1 GiB of dwords written, then read back as 2^27 dwordx2
        uint32_t *buf, *sink;
        hipMalloc(&buf, n * 4);
        hipMalloc(&sink, 64);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            float ms;
            hipEventRecord(a);
            hipLaunchKernelGGL(k_fill_dword, dim3(2048), dim3(256), 0, 0, buf, n);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("k_fill_dword: %zu bytes written, %.3f ms, %.1f GB/s\n", n * 4, ms, n * 4 / ms / 1e6);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_read_dwordx2, dim3(2048), dim3(256), 0, 0, (const uint2 *)buf, sink, n / 2);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("k_read_dwordx2: %zu bytes read, %.3f ms, %.1f GB/s\n", n * 4, ms, n * 4 / ms / 1e6);
        }
        hipFree(buf); hipFree(sink);
    }
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
