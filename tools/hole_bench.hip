// Dev-only (round 6): what does it cost to let a flagged lane-item SKIP its row store in k_main's hot loop and have the exact
// path store the whole piece later -- against today's byte patches (a masked write into a line that has left the L2: a
// read-modify-write behind the L2)?  Same grid, row addressing and pacing as tools/store_bench.hip's k_patch (an 8-byte load per
// iteration, `spin` dependent multiply-adds); one lane-item in RATE is flagged.
//   MODE 0  no flagged items (the write stream alone)
//   MODE 1  today: every item stored hot, a flagged one patches ONE byte `delay` iterations later
//   MODE 2  today's layout (a lane's two 16-byte pieces 64 bytes apart), flagged items skip the hot store, both pieces late
//   MODE 3  a lane's two pieces side by side (32 contiguous bytes per lane and iteration), flagged items skip, 32 bytes late
//   MODE 4  as 3, but nothing is skipped: the 32 bytes are written twice
//   MODE 5  two lanes per pair, 64 contiguous bytes per lane and iteration (four 16-byte stores), flagged items skip, 64 bytes late
//   MODE 6  layout of 3, every item stored hot, ONE byte patched late (is the byte patch cheaper in that layout?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(1024) void k_hole(uint8_t *out, uint32_t n_pairs, uint32_t row, uint32_t n_iter, uint32_t filler, int delay,
                                               int spin, const uint2 *gen, uint32_t rate) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t LPP = MODE == 5 ? 2u : 4u;            // lanes per pair
    constexpr uint32_t PPW = 64u / LPP;                      // pairs per wavefront
    constexpr uint32_t PPB = 16u * PPW;                      // pairs per block (pass of the workgroup)
    const uint32_t jl = lane & (LPP - 1u), wave_pair0 = (threadIdx.x >> 6) * PPW;
    const uint32_t n_pass = (n_pairs + PPB - 1u) / PPB;
    const uint32_t n_it = MODE == 5 ? (n_iter + 1u) / 2u : n_iter;  // (64 bytes per lane: half the iterations, the last one half used -- rounded up)
    uint32_t g = 0;
    float acc = (float)filler;
    auto piece_off = [&](uint32_t it) -> uint32_t {  // byte offset of the lane's first piece of iteration `it` inside the row
        if (MODE == 3 || MODE == 4 || MODE == 6) return it * 128u + jl * 32u;
        if (MODE == 5) return it * 128u + jl * 64u;
        return it * 128u + jl * 16u;
    };
    for (uint32_t blk = blockIdx.x; blk < n_pass; blk += gridDim.x) {
        const uint32_t pair = blk * PPB + wave_pair0 + lane / LPP;
        for (uint32_t it = 0; it < n_it; ++it, ++g) {
            uint2 gw = gen[((pair + filler) * 37u + it * 2u) & 0xfffffu];
            for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
            const uint4 v = make_uint4(pair + filler, it + gw.x, lane + gw.y, __float_as_uint(acc));
            const uint32_t hsh = (pair * 2654435761u + it * 40503u + lane * 97u + filler) >> 7;
            const bool flagged = MODE != 0 && hsh % rate == 0u;
            const bool skip = flagged && (MODE == 2 || MODE == 3 || MODE == 5);
            if (pair < n_pairs && !skip) {
                uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)pair * row + piece_off(it));
                if (MODE == 3 || MODE == 4 || MODE == 6) { dst[0] = v; dst[1] = v; }
                else if (MODE == 5) { dst[0] = v; dst[1] = v; dst[2] = v; dst[3] = v; }
                else { dst[0] = v; dst[4] = v; }
            }
            // the late work for the item of `delay` iterations ago (same lane, same wavefront)
            if (MODE != 0 && g >= (uint32_t)delay) {
                const uint32_t g2 = g - (uint32_t)delay, pass2 = g2 / n_it, it2 = g2 - pass2 * n_it;
                const uint32_t pair2 = (blockIdx.x + pass2 * gridDim.x) * PPB + wave_pair0 + lane / LPP;
                const uint32_t hsh2 = (pair2 * 2654435761u + it2 * 40503u + lane * 97u + filler) >> 7;
                if (pair2 < n_pairs && hsh2 % rate == 0u) {
                    uint8_t *p = out + (size_t)pair2 * row + piece_off(it2);
                    if (MODE == 1) p[(hsh2 & 15u) + ((hsh2 & 16u) ? 64u : 0u)] = (uint8_t)hsh2;
                    else if (MODE == 6) p[hsh2 & 31u] = (uint8_t)hsh2;
                    else {
                        uint4 *dst = reinterpret_cast<uint4 *>(p);
                        if (MODE == 2) { dst[0] = v; dst[4] = v; }
                        else if (MODE == 5) { dst[0] = v; dst[1] = v; dst[2] = v; dst[3] = v; }
                        else { dst[0] = v; dst[1] = v; }
                    }
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t n_pairs = 5000000, row = 640, n_iter = 5;
    uint8_t *buf;
    hipMalloc(&buf, (size_t)n_pairs * row);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int dev_cus = 256;
    uint2 *gen;
    hipMalloc(&gen, (size_t)8 << 20);
    hipMemset(gen, 0, (size_t)8 << 20);
    for (int rep = 0; rep < 2; ++rep)
    for (int spin : {140, 0})
    for (uint32_t rate : {7u, 4u, 2u})
    for (int delay : {2, 7, 14})
    for (int mode = 0; mode < 7; ++mode) {
        if (mode == 0 && (delay != 2 || rate != 7u)) continue;
        const int K = 10;
        auto launch = [&](uint32_t f) {
#define L(M) if (mode == M) hipLaunchKernelGGL(k_hole<M>, dim3(dev_cus), dim3(1024), 0, 0, buf, n_pairs, row, n_iter, f, delay, spin, gen, rate);
            L(0) L(1) L(2) L(3) L(4) L(5) L(6)
#undef L
        };
        for (int w = 0; w < 2; ++w) launch((uint32_t)w);
        hipEventRecord(a, 0);
        for (int k = 0; k < K; ++k) launch((uint32_t)k);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("rep %d spin %3d rate 1/%u delay %2d mode %d: %.4f ms per launch\n", rep, spin, rate, delay, mode, ms / K);
        fflush(stdout);
    }
    return 0;
}
