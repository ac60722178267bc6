// Dev-only microbenchmark (not part of the product): issue cost of the gfx950 instructions k_main is made of.
// Every kernel runs REPS x 32 copies of ONE instruction on 8 independent register chains; the grid fills every
// SIMD with W wavefronts.  Reported: cycles per wavefront-instruction per SIMD from s_memtime (clock independent,
// averaged over the waves of the launch) and from wall time at the clock the launch sustained (s_memtime / wall).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

static int REPS = 128;  // x 32 instructions (argv[2])

#define KERNEL(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint64_t *ticks, uint32_t s0, int reps) {            \
        uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 11 + 4,             \
                 r5 = r0 * 13 + 5, r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;                                               \
        uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;                             \
        uint32_t lds_a = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1024, lds_a8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 1024;                                        \
        __shared__ uint32_t lds[8192];                                                                               \
        for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;                                      \
        __syncthreads();                                                                                             \
        const uint64_t t0 = __builtin_readcyclecounter();                                                            \
        for (int i = 0; i < reps; ++i) { REP32(ASM) }                                                               \
        const uint64_t t1 = __builtin_readcyclecounter();                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7); \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                            \
    }

#define R(n) r##n
#define Q(n) q##n
// one VGPR chain: dst = op(dst, other chain, scalar)
#define A_ADD(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_XOR(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_LSHL(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(R(n)));
#define A_BFE(n) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(R(n)));
#define A_PERM(n) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_ALIGNBIT(n) asm volatile("v_alignbit_b32 %0, %0, %0, 31" : "+v"(R(n)));
#define A_CNDMASK(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(R(n)) : "v"(r0) : "vcc");
#define A_CMP(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(R(n)), "v"(r0) : "vcc");
#define A_ADD3(n) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_OR3(n) asm volatile("v_or3_b32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_BITOP3(n) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96" : "+v"(R(n)) : "s"(s0));
#define A_XAD(n) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_LSHLADD(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %0" : "+v"(R(n)));
#define A_ANDOR(n) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_MAD24(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_MUL24(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_MULHI24(n) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_MULLO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_MULHI(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_MAD64(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(Q(n)) : "v"(R(n)), "s"(s0) : "vcc");
#define A_MAD64C(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(Q(n)) : "v"(R(n)), "s"(s0) : "vcc");
#define A_MADU16(n) asm volatile("v_mad_u16 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_SDWA(n) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(R(n)) : "v"(r0));
#define A_SDWAW(n) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(R(n)) : "v"(r0));
#define A_PKADD(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(R(n)) : "v"(r0));
#define A_PKSUB(n) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(R(n)) : "v"(r0));
#define A_PKMAX(n) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(R(n)) : "v"(r0));
#define A_PKMIN(n) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(R(n)) : "v"(r0));
#define A_PKMUL(n) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(R(n)) : "v"(r0));
#define A_PKMAD(n) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(R(n)) : "v"(r0));
#define A_PKLSHR(n) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(R(n)));
#define A_FMA(n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_PKFMA(n) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(Q(n)));
#define A_MOV(n) asm volatile("v_mov_b32 %0, %1" : "=v"(R(n)) : "v"(r0));
#define A_MOVDPP(n) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(R(n)));
#define A_SAD(n) asm volatile("v_sad_u32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_MIN3(n) asm volatile("v_min3_u32 %0, %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_MAXU(n) asm volatile("v_max_u32 %0, %0, %1" : "+v"(R(n)) : "s"(s0));
#define A_SUBB(n) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(R(n)) : "v"(r0) : "vcc");
#define A_READLANE(n) asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0" : "+v"(R(n)) : : "s20");
#define A_MBCNT(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(R(n)) : "s"(s0));
#define A_ADD64(n) asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(Q(n)));
#define A_DOT4(n) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(R(n)) : "v"(r0));
#define A_CVT(n) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(R(n)));
#define A_LDSB32(n) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(R(n)) : "v"(lds_a));
#define A_LDSU8(n) asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(R(n)) : "v"(lds_a));
#define A_LDSU16(n) asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(R(n)) : "v"(lds_a));
#define A_LDSB64(n) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(6)" : "=v"(Q(n)) : "v"(lds_a8));
#define A_LDS2B32(n) asm volatile("ds_read2_b32 %0, %1 offset1:1\n s_waitcnt lgkmcnt(6)" : "=v"(Q(n)) : "v"(lds_a8));
#define A_SALU(n) asm volatile("s_add_u32 s20, s20, %0" : : "s"(s0) : "s20");
// mixes: what co-issues?
#define A_MIX_VS(n) asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, %1" : "+v"(R(n)) : "s"(s0) : "s20");
#define A_MIX_VL(n) asm volatile("v_add_u32 %0, %0, %2\n ds_read_b64 %1, %3\n s_waitcnt lgkmcnt(6)" : "+v"(R(n)), "=v"(Q(n)) : "s"(s0), "v"(lds_a8));
#define A_MIX_MADADD(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n v_add_u32 %1, %1, %2" : "+v"(Q(n)), "+v"(R(n)) : "s"(s0) : "vcc");

#define ALL(X)                                                                                                       \
    X(add, A_ADD, 1) X(xor_, A_XOR, 1) X(lshl, A_LSHL, 1) X(bfe, A_BFE, 1) X(perm, A_PERM, 1) X(alignbit, A_ALIGNBIT, 1)          \
    X(cndmask, A_CNDMASK, 1) X(cmp, A_CMP, 1) X(add3, A_ADD3, 1) X(or3, A_OR3, 1) X(bitop3, A_BITOP3, 1) X(xad, A_XAD, 1)         \
    X(lshl_add, A_LSHLADD, 1) X(and_or, A_ANDOR, 1) X(mad_u32_u24, A_MAD24, 1) X(mul_u32_u24, A_MUL24, 1)                      \
    X(mul_hi_u32_u24, A_MULHI24, 1) X(mul_lo_u32, A_MULLO, 1) X(mul_hi_u32, A_MULHI, 1) X(mad_u64_u32, A_MAD64, 1)           \
    X(mad_u64_u32_acc, A_MAD64C, 1) X(mad_u16, A_MADU16, 1) X(add_sdwa_byte, A_SDWA, 1) X(sub_sdwa_word, A_SDWAW, 1)           \
    X(pk_add_u16, A_PKADD, 1) X(pk_sub_i16, A_PKSUB, 1) X(pk_max_u16, A_PKMAX, 1) X(pk_min_u16, A_PKMIN, 1)                   \
    X(pk_mul_lo_u16, A_PKMUL, 1) X(pk_mad_u16, A_PKMAD, 1) X(pk_lshrrev_b16, A_PKLSHR, 1) X(fma_f32, A_FMA, 1)                \
    X(pk_fma_f32, A_PKFMA, 1) X(mov, A_MOV, 1) X(mov_dpp, A_MOVDPP, 1) X(sad_u32, A_SAD, 1) X(min3, A_MIN3, 1) X(max_u32, A_MAXU, 1) \
    X(subb, A_SUBB, 1) X(readlane_add, A_READLANE, 2) X(mbcnt, A_MBCNT, 1) X(lshl_add_u64, A_ADD64, 1) X(dot4_u32_u8, A_DOT4, 1)  \
    X(cvt_f32_u32, A_CVT, 1) X(ds_read_b32, A_LDSB32, 1) X(ds_read_u8, A_LDSU8, 1) X(ds_read_u16, A_LDSU16, 1)                 \
    X(ds_read_b64, A_LDSB64, 1) X(ds_read2_b32, A_LDS2B32, 1)               \
    X(mix_valu_lds, A_MIX_VL, 2) X(mix_mad64_add, A_MIX_MADADD, 2)

#define DEF(NAME, ASM, N) KERNEL(k_##NAME, ASM)
ALL(DEF)

struct Entry {
    const char *name;
    void (*fn)(uint32_t *, uint64_t *, uint32_t, int);
    int n;
};
#define ENT(NAME, ASM, N) {#NAME, k_##NAME, N},
static Entry entries[] = {ALL(ENT)};

int main(int argc, char **argv) {
    if (argc > 2) REPS = atoi(argv[2]);
    const int n_cu = 256;
    uint32_t *out;
    uint64_t *ticks;
    hipMalloc(&out, sizeof(uint32_t) * n_cu * 8 * 256);
    hipMalloc(&ticks, sizeof(uint64_t) * n_cu * 8 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("%-18s %5s %12s %12s %10s %10s\n", "instruction", "W", "tick(memtime)", "cyc(wall@2.4)", "ns/inst", "wall_ms");
    for (const Entry &e : entries) {
        for (int W : {1, 2, 4, 8}) {
            if (argc > 1 && !strstr(argv[1], (std::string(",") + e.name + ",").c_str())) continue;
            const int blocks = n_cu * W;  // 256-thread blocks: 4 waves, one per SIMD
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, ticks, 3u, REPS);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            std::vector<uint64_t> t(blocks * 4);
            hipMemcpy(t.data(), ticks, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost);
            double sum = 0;
            for (uint64_t v : t) sum += (double)v;
            const double n_inst = (double)REPS * 32 * e.n;
            // every wave sees its SIMD shared by W waves: its own elapsed ticks / (instructions x W)
            printf("%-18s %5d %12.2f %12.2f %10.3f %10.4f\n", e.name, W, sum / t.size() / n_inst / W,
                   best * 1e-3 * 2.4e9 / (n_inst * W), best * 1e6 / (n_inst * W), best);
        }
    }
    return 0;
}
