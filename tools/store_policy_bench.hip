// Dev-only (round 6): k_main's write stream under the store instruction's cache-policy bits and with whole 128-byte lines per
// instruction -- same grid, row addressing and pacing as tools/store_bench.hip.  LINE 0: two 16-byte stores 64 bytes apart per lane
// (a pair's four lanes write half a line per instruction); LINE 1: whole lines per instruction (lanes l / l + 32 hold different
// pairs: the first instruction writes pairs 0-7's lines, the second pairs 8-15's).  POL 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int POL>
__device__ __forceinline__ void store16(uint8_t *p, uint4 v) {
    const v4u w = {v.x, v.y, v.z, v.w};
    const uint64_t a = (uint64_t)p;
    if (POL == 0) *reinterpret_cast<uint4 *>(p) = v;
    else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(a), "v"(w) : "memory");
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(a), "v"(w) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(a), "v"(w) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(a), "v"(w) : "memory");
}

template <int LINE, int POL>
__global__ __launch_bounds__(1024) void k_store(uint8_t *out, uint32_t n_pairs, uint32_t row, uint32_t n_iter, uint32_t filler, int spin,
                                               const uint2 *gen) {
    const uint32_t lane = threadIdx.x & 63u, j4 = lane & 3u, wave_pair0 = (threadIdx.x >> 6) * 16u;
    const uint32_t n_pass = (n_pairs + 255u) / 256u;
    float acc = (float)filler;
    for (uint32_t blk = blockIdx.x; blk < n_pass; blk += gridDim.x) {
        const uint32_t pair = blk * 256u + wave_pair0 + (lane >> 2);
        if (pair >= n_pairs) continue;
        for (uint32_t it = 0; it < n_iter; ++it) {
            uint2 gw = gen[((pair + filler) * 37u + it * 2u) & 0xfffffu];
            for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
            const uint4 v = make_uint4(pair + filler, it + gw.x, lane + gw.y, __float_as_uint(acc));
            uint8_t *base = out + (size_t)pair * row + it * 128u + j4 * 16u;
            if (LINE == 0) {
                store16<POL>(base, v);
                store16<POL>(base + 64, v);
            } else {
                // lanes 0-31: pairs p0 .. p0+7, lanes 32-63: pairs p0+8 .. p0+15.  First instruction: lines of pairs p0 .. p0+7
                // (lanes 0-31 their forward pieces, lanes 32-63 the reverse pieces of the pair 8 below); second: pairs p0+8 ..
                const bool hi = lane >= 32u;
                uint8_t *a = hi ? base - 8u * (size_t)row + 64 : base;
                uint8_t *b = hi ? base + 64 : base + 8u * (size_t)row;
                store16<POL>(a, v);
                store16<POL>(b, v);
            }
        }
    }
}

int main() {
    const uint32_t n_pairs = 5000000, row = 640, n_iter = 5;
    uint8_t *buf;
    hipMalloc(&buf, (size_t)n_pairs * row);
    uint2 *gen;
    hipMalloc(&gen, (size_t)8 << 20);
    hipMemset(gen, 0, (size_t)8 << 20);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
    for (int spin : {0, 140})
    for (int line = 0; line < 2; ++line)
    for (int pol = 0; pol < 5; ++pol) {
        const int K = 10;
        auto launch = [&](uint32_t f) {
#define L(LN, P) if (line == LN && pol == P) hipLaunchKernelGGL((k_store<LN, P>), dim3(256), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, f, spin, gen);
            L(0, 0) L(0, 1) L(0, 2) L(0, 3) L(0, 4) L(1, 0) L(1, 1) L(1, 2) L(1, 3) L(1, 4)
#undef L
        };
        for (int w = 0; w < 2; ++w) launch((uint32_t)w);
        hipEventRecord(a, 0);
        for (int k = 0; k < K; ++k) launch((uint32_t)k);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("rep %d spin %3d line %d pol %d: %.4f ms per launch, %.0f GB/s\n", rep, spin, line, pol, ms / K, (double)n_pairs * row / (ms / K * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
