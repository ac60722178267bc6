// iss_kernels.hip.h -- gfx950 kernels of the read-generation path (included by iss_mi355x.hip).
//
// Reference semantics (InSilicoSeq v2.0.1), restated for a counter-based RNG:
//   simulate_read                      iss/generator.py:98-192
//   introduce_indels/adjust_seq_length iss/error_models/__init__.py:158-228, 114-156
//   gen_phred_scores/random_insert_size iss/error_models/kde.py:52-98
//   mut_sequence                       iss/error_models/__init__.py:69-112
//
// Kernel plan (one iss_generate call = up to four launches on one stream):
//   k_setup  : 1 lane / pair   -> PairDesc {forward_start, reverse_end, bin slots, attempt, insert}
//   k_main   : persistent workgroups (1024 lanes, one per CU); the compressed per-position quality
//              CDF rows of a position tile are staged ONCE per workgroup in LDS; 1 lane /
//              (pair, 4 consecutive positions, both mates): two Philox calls give the sixteen
//              16-bit leading digits of its 16 uniforms; CDF inversion = LDS guide byte + packed
//              (threshold, phred) entries; bases come from the 2-bit genome with funnel shifts
//              and one v_perm; four packed dword stores per lane, contiguous across lanes.
//              Assumes "no indel in this read" (true for all but ~1e-4 of reads of shipped models).
//   k_indel_scan : 1 lane / (pair, position group with a non-zero indel probability): draws the
//              indel digits and flags reads in which an indel MAY fire (conservative).
//   k_indel_fixup: 1 wavefront / flagged read: exact sequential indel semantics (lane 0 walks the
//              token transducer over an event mask computed by all lanes) + re-mutation by all
//              lanes, rewrites that read's base row.
// No MFMA anywhere: this is sampling/indexing.  All f64 comparisons of the reference are exact
// integer comparisons here (thresholds prepared on the host, see iss_mi355x.h / DESIGN.md).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iss {

// ---------------------------------------------------------------- RNG address map
enum : uint32_t {
    K_PAIR = 0, K_FS = 1, K_RS = 2, K_QM = 3, K_SUB = 4, K_INS = 5, K_DEL = 6, K_QM_LO = 7, K_INS_LO = 8,
    K_DEL_LO = 9
};

struct u32x4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

struct Addr {  // per-pair part of the Philox address
    uint32_t c0, c1, k0, k1;
};

__device__ __forceinline__ Addr make_addr(uint64_t seed, uint64_t ordinal, uint32_t attempt) {
    return {(uint32_t)ordinal, (uint32_t)((ordinal >> 32) & 0xffffu) | (attempt << 16), (uint32_t)seed,
            (uint32_t)(seed >> 32)};
}
__device__ __forceinline__ u32x4 draw_block(const Addr &a, uint32_t kind, uint32_t index, uint32_t sub) {
    return philox4x32_10(a.c0, a.c1, (kind << 24) | (index & 0xffffffu), sub, a.k0, a.k1);
}
__device__ __forceinline__ uint32_t word_of(const u32x4 &v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
// full-width numerator, == genrand_res53: (w0>>5)*2^26 + (w1>>6)
__device__ __forceinline__ uint64_t mk53(uint32_t w0, uint32_t w1) { return ((uint64_t)(w0 >> 5) << 26) | (w1 >> 6); }
// digit draws: m = (h16 << 37) | l37
__device__ __forceinline__ uint32_t digit16(const u32x4 &v, int d) { return (word_of(v, d >> 1) >> (16 * (d & 1))) & 0xffffu; }
__device__ __forceinline__ uint64_t lo37(const u32x4 &v, int pair) {
    return pair ? (((uint64_t)v.z << 5) | (v.w >> 27)) : (((uint64_t)v.x << 5) | (v.y >> 27));
}
__device__ __forceinline__ uint64_t mk_digit(uint32_t h16, uint64_t l37) { return ((uint64_t)h16 << 37) | l37; }

// ---------------------------------------------------------------- device-side tables
struct DevModel {
    int32_t RL, n_isize, n_q, G, pitch;  // G = pitch/4 = position groups per read
    // compressed quality rows for k_main (built at upload, see iss_mi355x.hip: build_qrows)
    int32_t NB;          // bin slots per orientation (non-empty bins, compacted)
    int32_t stride_w;    // u32 words per row: 16 guide words + (S_max + 1) entries, multiple of 4
    int32_t TG, TP;      // position groups / positions per tile
    int32_t n_tiles;
    int32_t tile_words;  // 2 * NB * TP * stride_w
    int8_t bin_slot[8];  // [o][bin] -> slot (or -1)
    int8_t slot_bin[8];  // [o][slot] -> bin
    const uint32_t *qrows;      // [n_tiles][2][NB][TP][stride_w]
    const uint32_t *mut16;      // [n_q+1]  mut_thr >> 37
    const uint64_t *isize_thr;  // [n_isize]
    const uint64_t *bin_thr;    // [2][4]
    const uint64_t *q_thr;      // [2][4][RL][n_q]   (exact tie resolution)
    const uint64_t *subst_thr;  // [2][RL][4][3]
    const uint8_t *subst_alt;   // [2][RL][4][3]
    const uint64_t *ins_thr;    // [2][RL][4]
    const uint8_t *ins_letter;  // [2][RL][4]
    const uint64_t *del_thr;    // [2][RL][4]
    const uint64_t *del_thr_max;  // [2][RL]  max over bases
    const uint64_t *mut_thr;      // [n_q+1]
    const uint8_t *ins_any;       // [2][RL] any insertion threshold non-zero at (o, n)
    const int32_t *active_groups;  // groups (4 positions) containing an indel-active (o, n)
    const uint8_t *active_mask;    // [G] bit (o*4+c): (o, 4*g+c) has a non-zero indel threshold
    int32_t n_active_groups;
};

struct DevGenome {
    const uint32_t *packed;  // 2-bit codes, 16 bases / word (A,T,C,G = 0..3; exceptions 0); words -1 and
                             // ceil(L/16)..+1 are readable padding
    const uint32_t *mask;    // 1 bit / base: 1 = read the ASCII copy (IUPAC or lower case); same padding
    const uint8_t *ascii;
    int64_t L;
};

struct PairDesc {
    int32_t fs;     // forward_start
    int32_t re;     // reverse_end (reverse_start = re - RL)
    uint32_t meta;  // bits 0-1 bin slot fwd, 2-3 bin slot rev, 16-31 attempt
    int32_t isz;    // insert size
};

struct RunArgs {
    int64_t n_pairs;
    uint64_t first_ordinal;
    uint64_t seed;
    int32_t sequence_type;
    int32_t gc_bias;
    uint64_t gc_thr;  // ceil(0.90 * 2^53): accept iff m < gc_thr (generator.py:88)
    uint8_t *out[4];  // rows of this launch: R1 base, R1 qual, R2 base, R2 qual
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint8_t code_to_ascii(uint32_t code) { return (uint8_t)((0x47435441u >> (8 * code)) & 0xffu); }
// A,T,C,G (either case) -> 0..3, anything else (IUPAC ambiguity codes) -> -1
__device__ __forceinline__ int base_index(int c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'T': case 't': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        default: return -1;
    }
}
// iss/util.py:57-88 (letters were validated at upload)
__device__ __forceinline__ int complement_ascii(int c) {
    const int lower = c & 0x20;
    int r;
    switch (c & ~0x20) {
        case 'A': r = 'T'; break; case 'T': r = 'A'; break; case 'C': r = 'G'; break; case 'G': r = 'C'; break;
        case 'Y': r = 'R'; break; case 'R': r = 'Y'; break; case 'K': r = 'M'; break; case 'M': r = 'K'; break;
        case 'B': r = 'V'; break; case 'V': r = 'B'; break; case 'D': r = 'H'; break; case 'H': r = 'D'; break;
        default: r = c & ~0x20; break;  // W, S, N
    }
    return r | lower;
}

__device__ __forceinline__ int fetch_ascii(const DevGenome &g, int64_t pos) {
    const uint32_t w = g.packed[pos >> 4];
    const uint32_t mk = g.mask[pos >> 5];
    if ((mk >> (pos & 31)) & 1u) return g.ascii[pos];
    return code_to_ascii((w >> ((pos & 15) * 2)) & 3u);
}
// E_fwd(k) = g[fs+k] ('A' past the end); E_rev(k) = comp(g[re-1-k]) ('A' before the start)
// -- template (k < RL) and adjust_seq_length padding (k >= RL) in one rule, __init__.py:141-155
__device__ __forceinline__ int read_dir_base(const DevGenome &g, int o, const PairDesc &d, int k) {
    if (o == 0) {
        const int64_t pos = (int64_t)d.fs + k;
        return pos < g.L ? fetch_ascii(g, pos) : 'A';
    }
    const int64_t pos = (int64_t)d.re - 1 - k;
    return pos >= 0 ? complement_ascii(fetch_ascii(g, pos)) : 'A';
}

// #(thr[i] < m), thr sorted ascending (np.searchsorted side='left')
__device__ __forceinline__ int count_lt(const uint64_t *thr, int n, uint64_t m) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (thr[mid] < m) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// #(thr[i] <= m) (side='right')
__device__ __forceinline__ int count_le(const uint64_t *thr, int n, uint64_t m) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (thr[mid] <= m) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// CPython _randbelow_with_getrandbits(n), 1 <= n < 2^32, words from the K_FS / K_RS streams
__device__ __forceinline__ uint32_t randbelow(const Addr &a, uint32_t kind, uint32_t n) {
    const int k = 32 - __clz(n);
    u32x4 blk = {0, 0, 0, 0};
    for (uint32_t t = 0;; ++t) {
        if ((t & 3u) == 0) blk = draw_block(a, kind, t >> 2, 0);
        const uint32_t r = word_of(blk, (int)(t & 3u)) >> (32 - k);
        if (r < n) return r;
    }
}

// exact phred for a tied leading digit: second-stage bits + search of the full thresholds
__device__ __forceinline__ int quality_exact(const DevModel &M, const Addr &a, int o, int slot, int p, uint32_t h) {
    const u32x4 lo = draw_block(a, K_QM_LO, (uint32_t)p, (uint32_t)o);
    const uint64_t m = mk_digit(h, lo37(lo, 0));
    const int bin = M.slot_bin[o * 4 + slot];
    return count_lt(M.q_thr + ((size_t)(o * 4 + bin) * M.RL + p) * M.n_q, M.n_q, m);
}
// exact "is it an error" for a tied leading digit
__device__ __forceinline__ bool mut_exact(const DevModel &M, const Addr &a, int o, int p, uint32_t h, int q) {
    const u32x4 lo = draw_block(a, K_QM_LO, (uint32_t)p, (uint32_t)o);
    return mk_digit(h, lo37(lo, 1)) > M.mut_thr[q];
}
// np.random.choice(alternatives, p=...) for an erroneous, non-ambiguous base (__init__.py:95-97)
__device__ __forceinline__ int substitute(const DevModel &M, const Addr &a, int o, int p, int base) {
    const int bi = base_index(base);
    if (bi < 0) return base;  // nucl.upper() in "RYWSMKHBVDN": left alone
    const u32x4 s = draw_block(a, K_SUB, (uint32_t)p, 0);
    const uint64_t m = o ? mk53(s.z, s.w) : mk53(s.x, s.y);
    const size_t row = ((size_t)(o * M.RL + p) * 4 + bi) * 3;
    const int k = (m >= M.subst_thr[row]) + (m >= M.subst_thr[row + 1]);
    return M.subst_alt[row + k];
}

// quality CDF inversion on a compressed row (LDS or global): 64 guide bytes, then packed entries
// (t16 << 8 | phred) sorted by t16 with a sentinel; phred = #(thresholds < m) if no tie.
__device__ __forceinline__ uint32_t qrow_lookup(const uint32_t *row, uint32_t h) {
    uint32_t j = reinterpret_cast<const uint8_t *>(row)[h >> 10];
    uint32_t e = row[16 + j];
    while ((e >> 8) < h) e = row[16 + (++j)];
    return e;  // (e >> 8) == h  <=>  tie
}

// ================================================================== k_setup
__global__ __launch_bounds__(256) void k_setup(DevModel M, DevGenome g, RunArgs A, PairDesc *desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_pairs) return;
    const uint64_t ordinal = A.first_ordinal + (uint64_t)i;
    uint32_t attempt = 0;
    if (A.gc_bias) {  // generator.py:82-92 -- the 40<gc<60 window is dead, every candidate pair
                      // costs one uniform and survives iff u < 0.90
        const uint32_t th = (uint32_t)(A.gc_thr >> 26);
        for (; attempt < 0xffffu; ++attempt) {
            const Addr a = make_addr(A.seed, ordinal, attempt);
            const u32x4 w = draw_block(a, K_PAIR, 0, 0);
            const uint32_t mh = w.w >> 5;
            bool ok = mh < th;
            if (mh == th) ok = mk53(w.w, draw_block(a, K_PAIR, 0, 1).w) < A.gc_thr;
            if (ok) break;
        }
    }
    const Addr a = make_addr(A.seed, ordinal, attempt);
    const u32x4 w0 = draw_block(a, K_PAIR, 0, 0);
    const u32x4 w1 = draw_block(a, K_PAIR, 0, 1);
    const int RL = M.RL;
    const int64_t L = g.L;
    const int isz = count_lt(M.isize_thr, M.n_isize, mk53(w0.x, w1.x));  // kde.py:97
    int bin_f = count_le(M.bin_thr, 4, mk53(w0.y, w1.y));                  // kde.py:74
    int bin_r = count_le(M.bin_thr + 4, 4, mk53(w0.z, w1.z));
    bin_f = bin_f > 3 ? 3 : bin_f;  // kde.py:77-78
    bin_r = bin_r > 3 ? 3 : bin_r;
    int64_t fs, rs, re;
    if (A.sequence_type == 0) {
        const int64_t width = L - ((int64_t)isz + 2 * RL);  // generator.py:135
        if (width > 0) fs = randbelow(a, K_FS, (uint32_t)width);
        else fs = randbelow(a, K_FS, (uint32_t)(L - RL));   // generator.py:144
        rs = fs + RL + isz;                                 // generator.py:165
        re = rs + RL;
    } else {
        fs = 0;                                             // generator.py:137
        rs = L - RL;                                        // generator.py:168
        re = L;
    }
    if (re > L) {                                           // generator.py:172-176
        re = RL + (int64_t)randbelow(a, K_RS, (uint32_t)(L - RL));
        rs = re - RL;
    }
    PairDesc d;
    d.fs = (int32_t)fs;
    d.re = (int32_t)re;
    d.meta = (uint32_t)(M.bin_slot[bin_f] & 3) | ((uint32_t)(M.bin_slot[4 + bin_r] & 3) << 2) | (attempt << 16);
    d.isz = isz;
    desc[i] = d;
}

// ================================================================== k_main
__device__ __forceinline__ uint32_t funnel_r(uint32_t lo, uint32_t hi, uint32_t sh) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31));
}
// 4 two-bit codes (bits 0..7 of b) -> 4 ASCII bytes, byte k = base of code k
__device__ __forceinline__ uint32_t codes_to_ascii4(uint32_t b) {
    const uint32_t x = (b | (b << 12)) & 0x000f000fu;   // (c1 c0) in the low half, (c3 c2) in the high half
    const uint32_t sel = (x | (x << 6)) & 0x03030303u;  // one code per byte
    return __builtin_amdgcn_perm(0u, 0x47435441u, sel);  // selector values 0..3 pick bytes of "ATCG"
}

__global__ __launch_bounds__(1024) void k_main(DevModel M, DevGenome g, RunArgs A, const PairDesc *__restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tile = blockIdx.x % M.n_tiles;
    const uint32_t wg = blockIdx.x / M.n_tiles, n_wg = gridDim.x / M.n_tiles;
    {   // stage this tile's quality rows + the substitution-test thresholds in LDS (once per workgroup)
        const uint4 *src = reinterpret_cast<const uint4 *>(M.qrows + (size_t)tile * M.tile_words);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < M.tile_words / 4; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i <= M.n_q; i += blockDim.x) lds[M.tile_words + i] = M.mut16[i];
    }
    __syncthreads();
    const uint32_t *mut16 = lds + M.tile_words;
    const int g0 = tile * M.TG;
    const uint32_t tg = (uint32_t)min(M.TG, M.G - g0);  // groups of this tile
    const uint32_t n_items = (uint32_t)A.n_pairs * tg;
    const uint32_t step = n_wg * blockDim.x;
    const uint32_t step_pair = step / tg, step_grp = step - step_pair * tg;
    uint32_t it = wg * blockDim.x + threadIdx.x;
    uint32_t pair = it / tg, grp = it - pair * tg;
    const int RL = M.RL;
    const size_t mate_rows = (size_t)M.NB * M.TP;  // rows per orientation inside the tile
    for (; it < n_items; it += step) {
        const int p0 = (g0 + (int)grp) * 4;
        const PairDesc d = desc[pair];
        const Addr a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
        const uint32_t slot_f = d.meta & 3u, slot_r = (d.meta >> 2) & 3u;
        // ---- template bases: forward g[fs+p0 .. +3]; reverse comp(g[re-1-p0 .. -3])
        uint32_t fb, fm, rb, rm;
        {
            const int32_t pf = d.fs + p0;
            const uint32_t *pw = g.packed + (pf >> 4);
            fb = funnel_r(pw[0], pw[1], (uint32_t)(pf & 15) * 2) & 0xffu;
            const uint32_t *mw = g.mask + (pf >> 5);
            fm = funnel_r(mw[0], mw[1], (uint32_t)(pf & 31)) & 0xfu;
            const int32_t pr = d.re - 4 - p0;  // lowest genome position of the 4 reverse bases
            const uint32_t *qw = g.packed + (pr >> 4);
            rb = funnel_r(qw[0], qw[1], (uint32_t)(pr & 15) * 2) & 0xffu;
            const uint32_t *nw = g.mask + (pr >> 5);
            rm = funnel_r(nw[0], nw[1], (uint32_t)(pr & 31)) & 0xfu;
        }
        uint32_t base_f = codes_to_ascii4(fb);
        uint32_t base_r = __builtin_amdgcn_perm(0u, codes_to_ascii4(rb ^ 0x55u), 0x00010203u);  // complement, reversed
        if (fm | rm) {  // IUPAC / lower-case letters: patch from the ASCII copy
            for (int c = 0; c < 4; ++c) {
                if ((fm >> c) & 1u) {
                    const uint32_t ch = g.ascii[(int64_t)d.fs + p0 + c];
                    base_f = (base_f & ~(0xffu << (8 * c))) | (ch << (8 * c));
                }
                if ((rm >> (3 - c)) & 1u) {
                    const uint32_t ch = (uint32_t)complement_ascii(g.ascii[(int64_t)d.re - 1 - p0 - c]);
                    base_r = (base_r & ~(0xffu << (8 * c))) | (ch << (8 * c));
                }
            }
        }
        // ---- phred scores and the substitution test, 16-bit leading digits
        const uint32_t *rows_f = lds + ((size_t)slot_f * M.TP + (p0 - g0 * 4)) * M.stride_w;
        const uint32_t *rows_r = lds + (mate_rows + (size_t)slot_r * M.TP + (p0 - g0 * 4)) * M.stride_w;
        uint32_t qual_f = 0, qual_r = 0;
        uint32_t slow = 0;  // per base c (fwd) / 4+c (rev): bit 0-7 quality tie, 8-15 error, 16-23 error tie
        const u32x4 wq0 = draw_block(a, K_QM, (uint32_t)(p0 >> 1), 0);
        const u32x4 wq1 = draw_block(a, K_QM, (uint32_t)(p0 >> 1) + 1, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int pc = min(p0 + c, RL - 1) - p0;  // clamp the padding lanes of the last group
            const u32x4 &w = (c >> 1) ? wq1 : wq0;
            const uint32_t wf = (c & 1) ? w.z : w.x, wr = (c & 1) ? w.w : w.y;
            {
                const uint32_t e = qrow_lookup(rows_f + (size_t)pc * M.stride_w, wf & 0xffffu);
                const uint32_t q = e & 0xffu, hm = wf >> 16, t = mut16[q];
                qual_f |= q << (8 * c);
                slow |= ((e >> 8) == (wf & 0xffffu) ? 1u : 0u) << c;
                slow |= (hm > t ? 1u : 0u) << (8 + c);
                slow |= (hm == t ? 1u : 0u) << (16 + c);
            }
            {
                const uint32_t e = qrow_lookup(rows_r + (size_t)pc * M.stride_w, wr & 0xffffu);
                const uint32_t q = e & 0xffu, hm = wr >> 16, t = mut16[q];
                qual_r |= q << (8 * c);
                slow |= ((e >> 8) == (wr & 0xffffu) ? 1u : 0u) << (4 + c);
                slow |= (hm > t ? 1u : 0u) << (12 + c);
                slow |= (hm == t ? 1u : 0u) << (20 + c);
            }
        }
        if (slow) {  // ties of a leading digit (~2e-4 / draw) and substitution errors (~1e-3 / base)
            for (int s = 0; s < 8; ++s) {
                if (!((slow >> s) & 0x010101u)) continue;
                const int o = s >> 2, c = s & 3, p = p0 + c;
                if (p >= RL) continue;
                const uint32_t wd = (c >> 1) ? (o ? ((c & 1) ? wq1.w : wq1.y) : ((c & 1) ? wq1.z : wq1.x))
                                             : (o ? ((c & 1) ? wq0.w : wq0.y) : ((c & 1) ? wq0.z : wq0.x));
                uint32_t &qual = o ? qual_r : qual_f;
                uint32_t &bases = o ? base_r : base_f;
                uint32_t q = (qual >> (8 * c)) & 0xffu;
                bool err = (slow >> (8 + s)) & 1u, tie_m = (slow >> (16 + s)) & 1u;
                if ((slow >> s) & 1u) {  // quality tie: exact phred, then redo the error test
                    q = (uint32_t)quality_exact(M, a, o, o ? slot_r : slot_f, p, wd & 0xffffu);
                    qual = (qual & ~(0xffu << (8 * c))) | (q << (8 * c));
                    const uint32_t t = mut16[q];
                    err = (wd >> 16) > t;
                    tie_m = (wd >> 16) == t;
                }
                if (tie_m) err = mut_exact(M, a, o, p, wd >> 16, (int)q);
                if (err) {
                    const uint32_t nb = (uint32_t)substitute(M, a, o, p, (int)((bases >> (8 * c)) & 0xffu));
                    bases = (bases & ~(0xffu << (8 * c))) | (nb << (8 * c));
                }
            }
        }
        const int nvalid = RL - p0;  // zero the padding bytes of the last group
        const uint32_t keep = nvalid >= 4 ? 0xffffffffu : ((1u << (8 * nvalid)) - 1u);
        const size_t dw = (size_t)pair * M.G + (size_t)(g0 + grp);
        reinterpret_cast<uint32_t *>(A.out[0])[dw] = base_f & keep;
        reinterpret_cast<uint32_t *>(A.out[1])[dw] = qual_f & keep;
        reinterpret_cast<uint32_t *>(A.out[2])[dw] = base_r & keep;
        reinterpret_cast<uint32_t *>(A.out[3])[dw] = qual_r & keep;
        pair += step_pair;
        grp += step_grp;
        if (grp >= tg) { grp -= tg; ++pair; }
    }
}

// ================================================================== k_indel_scan
// Conservative: flags mate o of a pair when some indel uniform's leading digit is <= the leading
// digit of a non-zero threshold (max over bases for deletions).  No flag  =>  provably no indel
// event (the first event in loop order would have been flagged), so k_main's output stands.
__global__ __launch_bounds__(256) void k_indel_scan(DevModel M, RunArgs A, const PairDesc *__restrict__ desc,
                                                    uint32_t *flags, uint32_t *fix_list, uint32_t *fix_count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_items = (uint32_t)A.n_pairs * (uint32_t)M.n_active_groups;
    if (t >= n_items) return;
    const uint32_t pair = t / (uint32_t)M.n_active_groups;
    const int grp = M.active_groups[t - pair * (uint32_t)M.n_active_groups];
    const uint32_t amask = M.active_mask[grp];
    const Addr a = make_addr(A.seed, A.first_ordinal + pair, desc[pair].meta >> 16);
    uint32_t cand = 0;
    u32x4 dl = {0, 0, 0, 0};
    bool have_del = false;
    for (int c = 0; c < 4; ++c) {
        const int n = grp * 4 + c;
        if (n > M.RL - 2) break;  // loop is range(read_length - 1), __init__.py:187
        if (!((amask >> c) & 0x11u)) continue;
        u32x4 wi = {0, 0, 0, 0};
        bool have_ins = false;
        for (int o = 0; o < 2; ++o) {
            if (!((amask >> (o * 4 + c)) & 1u)) continue;
            const size_t e = (size_t)o * M.RL + n;
            if (M.ins_any[e]) {
                if (!have_ins) { wi = draw_block(a, K_INS, (uint32_t)n, 0); have_ins = true; }
                for (int x = 0; x < 4; ++x) {
                    const uint64_t T = M.ins_thr[e * 4 + x];
                    if (T && digit16(wi, o * 4 + x) <= (uint32_t)(T >> 37)) cand |= 1u << o;
                }
            }
            const uint64_t Td = M.del_thr_max[e];
            if (Td) {
                if (!have_del) { dl = draw_block(a, K_DEL, (uint32_t)grp, 0); have_del = true; }
                if (digit16(dl, c * 2 + o) <= (uint32_t)(Td >> 37)) cand |= 1u << o;
            }
        }
    }
    if (cand) {
        const uint32_t old = atomicOr(&flags[pair], cand);
        uint32_t fresh = cand & ~old;
        while (fresh) {
            const int o = __ffs(fresh) - 1;
            fresh &= fresh - 1;
            fix_list[atomicAdd(fix_count, 1u)] = pair * 2u + (uint32_t)o;
        }
    }
}

// ================================================================== k_indel_fixup
// One wavefront per flagged read.  Exact introduce_indels + adjust_seq_length as a token transducer:
// the list prefix [0, n) is final when step n starts; the not-yet-visited suffix is
// (stack of freshly inserted letters, LIFO) ++ E(k), E(k+1), ...   (DESIGN.md "indel transducer").
//   phase 1 (all lanes): event mask per loop step n, independent of the token:
//            bits 0-3 insertion of letter slot x fires, bits 4-7 deletion fires if the token is base b
//   phase 2 (lane 0):    walk the steps, emit map[j] = source index k (>= 0) or -(inserted letter)
//   phase 3 (all lanes): token -> mut_sequence -> store base j
constexpr int FIX_MAX_RL = 1024;  // read_length limit (checked at model upload)
constexpr int FIX_WAVES = 4;      // wavefronts (reads) per workgroup

__global__ __launch_bounds__(64 * FIX_WAVES) void k_indel_fixup(DevModel M, DevGenome g, RunArgs A,
                                                                const PairDesc *__restrict__ desc,
                                                                const uint32_t *__restrict__ fix_list,
                                                                const uint32_t *__restrict__ fix_count,
                                                                uint64_t *stats) {
    __shared__ uint8_t s_ev[FIX_WAVES][FIX_MAX_RL];
    __shared__ int16_t s_map[FIX_WAVES][FIX_MAX_RL];
    __shared__ uint8_t s_stk[FIX_WAVES][FIX_MAX_RL];
    const uint32_t n_fix = *fix_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long *)stats, (unsigned long long)n_fix);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t *ev = s_ev[wv];
    int16_t *map = s_map[wv];
    uint8_t *stk = s_stk[wv];
    const int RL = M.RL;
    for (uint32_t i = blockIdx.x * FIX_WAVES + wv; i < n_fix; i += gridDim.x * FIX_WAVES) {
        const uint32_t e = fix_list[i];
        const uint32_t pair = e >> 1;
        const int o = (int)(e & 1u);
        const PairDesc d = desc[pair];
        const Addr a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
        // ---- phase 1
        for (int n = lane; n < RL - 1; n += 64) {
            const size_t en = (size_t)o * RL + n;
            uint32_t m8 = 0;
            if (M.ins_any[en]) {  // :193-196
                const u32x4 w = draw_block(a, K_INS, (uint32_t)n, 0);
                for (int x = 0; x < 4; ++x) {
                    const uint64_t T = M.ins_thr[en * 4 + x];
                    if (!T) continue;
                    const uint32_t h = digit16(w, o * 4 + x), th = (uint32_t)(T >> 37);
                    bool hit = h < th;
                    if (h == th) {
                        const u32x4 l = draw_block(a, K_INS_LO, (uint32_t)n, (uint32_t)(o * 2 + (x >> 1)));
                        hit = mk_digit(h, lo37(l, x & 1)) < T;
                    }
                    if (hit) m8 |= 1u << x;
                }
            }
            if (M.del_thr_max[en]) {  // :209-210
                const u32x4 w = draw_block(a, K_DEL, (uint32_t)n >> 2, 0);
                const uint32_t h = digit16(w, (n & 3) * 2 + o);
                for (int b = 0; b < 4; ++b) {
                    const uint64_t T = M.del_thr[en * 4 + b];
                    if (!T) continue;
                    const uint32_t th = (uint32_t)(T >> 37);
                    bool hit = h < th;
                    if (h == th) hit = mk_digit(h, lo37(draw_block(a, K_DEL_LO, (uint32_t)n, 0), o)) < T;
                    if (hit) m8 |= 16u << b;
                }
            }
            ev[n] = (uint8_t)m8;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // LDS writes of the wave visible to lane 0
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2
        if (lane == 0) {
            int sp = 0, k = 0;  // stack depth (bounded: only the top RL entries can ever surface), source index
            for (int n = 0; n < RL - 1; ++n) {
                const uint32_t m8 = ev[n];
                int tok;  // >= 0: source index, < 0: -(letter)
                if (sp > 0) tok = -(int)stk[--sp];
                else tok = k++;
                if (m8 == 0 || tok >= RL) { map[n] = (int16_t)tok; continue; }  // no event / n >= len(seq), :223
                const int ch = tok < 0 ? -tok : read_dir_base(g, o, d, tok);
                const int bi = base_index(ch);
                if (bi < 0) { map[n] = (int16_t)tok; continue; }  // ambiguous: skipped, :190-192
                for (int x = 0; x < 4; ++x)
                    if ((m8 >> x) & 1u) {
                        if (sp == FIX_MAX_RL) { for (int z = 1; z < sp; ++z) stk[z - 1] = stk[z]; --sp; }
                        stk[sp++] = M.ins_letter[((size_t)o * RL + n) * 4 + x];
                    }
                if ((m8 >> (4 + bi)) & 1u) {  // deleted: the next token slides in unvisited
                    if (sp > 0) tok = -(int)stk[--sp];
                    else tok = k++;
                }
                map[n] = (int16_t)tok;
            }
            map[RL - 1] = (int16_t)(sp > 0 ? -(int)stk[sp - 1] : k);  // index RL-1 is never visited
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 3
        uint8_t *out_base = A.out[2 * o] + (size_t)pair * M.pitch;
        const uint8_t *out_qual = A.out[2 * o + 1] + (size_t)pair * M.pitch;
        for (int j = lane; j < RL; j += 64) {
            const int tok = map[j];
            int base = tok < 0 ? -tok : read_dir_base(g, o, d, tok);
            const u32x4 w = draw_block(a, K_QM, (uint32_t)j >> 1, 0);
            const uint32_t h = digit16(w, (j & 1) * 4 + 2 * o + 1);
            const int q = out_qual[j];
            const uint32_t t = M.mut16[q];
            bool err = h > t;
            if (h == t) err = mut_exact(M, a, o, j, h, q);
            if (err) base = substitute(M, a, o, j, base);
            out_base[j] = (uint8_t)base;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace iss
