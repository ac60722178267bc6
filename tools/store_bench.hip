// Dev-only: what does k_main's WRITE STREAM cost by itself?  The same grid (one 1024-lane workgroup per CU), the same row
// addressing (a wavefront holds 16 pairs, four lanes each; per iteration a lane stores two 16-byte pieces 64 bytes apart;
// 5 iterations per pass; a workgroup takes every n_wg-th block of 256 pairs), no computation.  Variants: the two stores as
// they are; whole lines per store instruction (lanes l / l + 32 swapped); one 32-byte... Prints ms per 5 M pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(1024) void k_store(uint8_t *out, uint32_t n_pairs, uint32_t row, uint32_t n_iter, uint32_t filler) {
    const uint32_t lane = threadIdx.x & 63u, j4 = lane & 3u, wave_pair0 = (threadIdx.x >> 6) * 16u;
    const uint32_t n_pass = (n_pairs + 255u) / 256u;
    const uint32_t swap_a = lane < 32u ? 0u : 64u - 8u * row, swap_b = lane < 32u ? 8u * row : 64u;
    for (uint32_t blk = blockIdx.x; blk < n_pass; blk += gridDim.x) {
        const uint32_t pair = blk * 256u + wave_pair0 + (lane >> 2);
        if (pair >= n_pairs) continue;
        uint32_t out_b = pair * row + j4 * 16u;
        for (uint32_t it = 0; it < n_iter; ++it) {
            const uint4 v = make_uint4(pair + filler, it, lane, blk);
            if (MODE == 0) {
                uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)out_b);
                dst[0] = v;
                dst[4] = v;
            } else if (MODE == 1) {
                *reinterpret_cast<uint4 *>(out + (size_t)(out_b + swap_a)) = v;
                *reinterpret_cast<uint4 *>(out + (size_t)(out_b + swap_b)) = v;
            } else if (MODE == 2) {  // both pieces of a pair's line from ONE lane pair: lanes 2k, 2k+1 ... (8 lanes x 16 B = a line per pair-half): here
                                     // simply 32 contiguous bytes per lane (what a "sector per lane" layout would store)
                uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)(pair * row + it * 128u + j4 * 32u));
                dst[0] = v;
                dst[1] = v;
            }
            out_b += 128u;
        }
    }
}

// LOADS: 0 none; 1 a 8-byte load per iteration issued at its start and used at its end (k_main's genome windows: the wait for
// it is a wait for every older vector memory operation too); 2 the same, used one iteration later
template <int LOADS>
__global__ __launch_bounds__(1024) void k_patch(uint8_t *out, uint32_t n_pairs, uint32_t row, uint32_t n_iter, uint32_t filler, int delay,
                                                int spin, const uint2 *gen) {
    const uint32_t lane = threadIdx.x & 63u, j4 = lane & 3u, wave_pair0 = (threadIdx.x >> 6) * 16u;
    const uint32_t n_pass = (n_pairs + 255u) / 256u;
    uint32_t g = 0;
    float acc = (float)filler;
    uint2 g_prev = {0u, 0u};
    for (uint32_t blk = blockIdx.x; blk < n_pass; blk += gridDim.x) {
        const uint32_t pair = blk * 256u + wave_pair0 + (lane >> 2);
        uint32_t out_b = pair * row + j4 * 16u;
        for (uint32_t it = 0; it < n_iter; ++it, ++g) {
            uint2 gw = {0u, 0u};
            if (LOADS) gw = gen[((pair + filler) * 37u + it * 2u) & 0xfffffu];
            if (LOADS == 2) { const uint2 t = gw; gw = g_prev; g_prev = t; }
            for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
            const uint4 v = make_uint4(pair + filler, it + gw.x, lane + gw.y, __float_as_uint(acc));
            if (pair < n_pairs) {
                uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)out_b);
                dst[0] = v;
                dst[4] = v;
            }
            const uint32_t hsh = (pair * 2654435761u + it * 40503u + lane * 97u + filler) >> 7;
            if (delay >= 0 && g >= (uint32_t)delay && hsh % 7u == 0u) {
                const uint32_t g2 = g - (uint32_t)delay, pass2 = g2 / n_iter, it2 = g2 - pass2 * n_iter;
                const uint32_t pair2 = (blockIdx.x + pass2 * gridDim.x) * 256u + wave_pair0 + (lane >> 2);
                if (pair2 < n_pairs) out[(size_t)pair2 * row + j4 * 16u + it2 * 128u + (hsh & 15u) + ((hsh & 16u) ? 64u : 0u)] = (uint8_t)hsh;
            }
            out_b += 128u;
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t n_pairs = 5000000, row = 640, n_iter = 5;
    uint8_t *buf;
    hipMalloc(&buf, (size_t)n_pairs * row);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int dev_cus = 256;
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            const int K = 20;
            for (int w = 0; w < 3; ++w) {
                if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, (uint32_t)w);
                if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, (uint32_t)w);
                if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, (uint32_t)w);
            }
            hipEventRecord(a, 0);
            for (int k = 0; k < K; ++k) {
                if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, (uint32_t)k);
                if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, (uint32_t)k);
                if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, (uint32_t)k);
            }
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("mode %d: %.4f ms per launch, %.1f GB/s\n", mode, ms / K, (double)n_pairs * row / (ms / K * 1e-3) / 1e9);
        }
    // mode 3: the row stores as in mode 0 + BYTE PATCHES into the rows the wavefront wrote `delay` iterations earlier (one lane-
    // iteration in seven patches one byte, NovaSeq's rate), paced by `spin` dependent multiply-adds per iteration so that a launch
    // takes as long as k_main's: how long does a written line stay where a byte store is cheap?
    uint2 *gen;
    hipMalloc(&gen, (size_t)8 << 20);
    hipMemset(gen, 0, (size_t)8 << 20);
    for (int loads = 0; loads < 3; ++loads)
    for (int spin : {0, 100, 140, 180})
        for (int delay : {-1, 0, 2, 4, 7, 12}) {
            const int K = 10;
            auto launch = [&](uint32_t f) {
                if (loads == 0) hipLaunchKernelGGL(k_patch<0>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, f, delay, spin, gen);
                if (loads == 1) hipLaunchKernelGGL(k_patch<1>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, f, delay, spin, gen);
                if (loads == 2) hipLaunchKernelGGL(k_patch<2>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, f, delay, spin, gen);
            };
            for (int w = 0; w < 2; ++w) launch((uint32_t)w);
            hipEventRecord(a, 0);
            for (int k = 0; k < K; ++k) launch((uint32_t)k);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("patches: loads %d spin %d delay %d iterations: %.4f ms per launch\n", loads, spin, delay, ms / K);
        }
    return 0;
}
