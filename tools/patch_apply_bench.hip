// Dev-only (round 6): what does it cost to apply k_main's late byte patches in a kernel of their own?  3.2 GB of rows are streamed
// out (as k_main does), then N byte patches -- one per ~400 bytes, in row order as a patch list would hold them, or shuffled --
// are applied by one lane each.    hipcc --offload-arch=gfx950 -O3 -o tools/patch_apply_bench tools/patch_apply_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

__global__ void k_fill(uint4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void k_apply(uint8_t *out, const uint64_t *rec, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint64_t r = rec[i]; out[r >> 8] = (uint8_t)r; }
}
int main() {
    const size_t bytes = (size_t)5000000 * 640;
    uint8_t *buf; hipMalloc(&buf, bytes);
    for (int mode = 0; mode < 3; ++mode) {
        for (uint32_t n : {1000000u, 4000000u, 9000000u}) {
            std::vector<uint64_t> rec(n);
            std::mt19937_64 g(7);
            const size_t step = bytes / n;
            for (uint32_t i = 0; i < n; ++i) rec[i] = ((uint64_t)(i * step + g() % step) << 8) | 0x41u;
            if (mode == 1) std::shuffle(rec.begin(), rec.end(), g);
            if (mode == 2) {  // in row order inside windows of 64 K records (a moving window of the chip's concurrent waves), shuffled inside
                for (size_t a = 0; a < n; a += 65536) std::shuffle(rec.begin() + a, rec.begin() + std::min<size_t>(n, a + 65536), g);
            }
            uint64_t *d; hipMalloc(&d, (size_t)n * 8);
            hipMemcpy(d, rec.data(), (size_t)n * 8, hipMemcpyHostToDevice);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                k_fill<<<2048, 1024>>>(reinterpret_cast<uint4 *>(buf), bytes / 16);
                hipEventRecord(a);
                k_apply<<<(n + 255) / 256, 256>>>(buf, d, n);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                best = std::min(best, ms);
            }
            printf("mode %d (%s) n %u: apply %.4f ms\n", mode, mode == 0 ? "row order" : mode == 1 ? "shuffled" : "windows", n, best);
            hipFree(d);
        }
    }
    return 0;
}
