// iss_kernels.hip.h -- gfx950 kernels of the read-generation path (included by iss_mi355x.hip).
//
// Reference semantics (InSilicoSeq v2.0.1), restated for a counter-based RNG:
//   simulate_read                      iss/generator.py:98-192
//   introduce_indels/adjust_seq_length iss/error_models/__init__.py:158-228, 114-156
//   gen_phred_scores/random_insert_size iss/error_models/kde.py:52-98
//   mut_sequence                       iss/error_models/__init__.py:69-112
//
// Kernel plan (one iss_generate call = up to four launches on one stream):
//   k_setup  : 1 lane / pair   -> PairDesc {forward_start, reverse_end, bins, attempt, insert}
//   k_main   : 1 lane / (pair, 4 consecutive positions, both mates): quality CDF inversion,
//              substitution test + choice, packed dword stores (fully coalesced: item t writes
//              dword t of each of the four output arrays).  Assumes "no indel in this read".
//   k_indel_scan : 1 lane / (pair, group with a non-zero indel probability): draws the
//              indel uniforms and flags reads in which an indel MAY fire (conservative).
//   k_indel_fixup: 1 lane / flagged read: exact sequential indel semantics + re-mutation,
//              rewrites that read's base row.
// No MFMA anywhere: this is sampling/indexing.  All f64 comparisons of the reference are
// exact integer comparisons here (thresholds prepared on the host, see iss_mi355x.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iss {

// ---------------------------------------------------------------- RNG address map
enum : uint32_t { K_PAIR = 0, K_FS = 1, K_RS = 2, K_QM = 3, K_SUB = 4, K_INS = 5, K_DEL = 6 };

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
// 53-bit uniform numerator, == genrand_res53: (w0>>5)*2^26 + (w1>>6)
__device__ __forceinline__ uint64_t mk53(uint32_t w0, uint32_t w1) { return ((uint64_t)(w0 >> 5) << 26) | (w1 >> 6); }

// ---------------------------------------------------------------- device-side tables
struct DevModel {
    int32_t RL, n_isize, n_q, G, pitch;  // G = pitch/4 = position groups per read
    const uint64_t *isize_thr;           // [n_isize]
    const uint64_t *bin_thr;             // [2][4]
    const uint64_t *q_thr;               // [2][4][RL][n_q]
    const uint32_t *q_thr_hi;            // same shape, thr >> 26
    const uint64_t *subst_thr;           // [2][RL][4][3]
    const uint8_t *subst_alt;            // [2][RL][4][3]
    const uint64_t *ins_thr;             // [2][RL][4]
    const uint8_t *ins_letter;           // [2][RL][4]
    const uint64_t *del_thr;             // [2][RL][4]
    const uint64_t *del_thr_max;         // [2][RL]  max over bases
    const uint64_t *mut_thr;             // [n_q+1]
    const uint32_t *mut_thr_hi;          // [n_q+1]
    const uint8_t *ins_any;              // [2][RL] any insertion threshold non-zero at (o, n)
    const int32_t *active_groups;        // groups (4 positions) containing an indel-active (o, n)
    const uint8_t *active_mask;          // [G] bit (o*4+c): (o, 4*g+c) has a non-zero indel threshold
    int32_t n_active_groups;
};

struct DevGenome {
    const uint32_t *packed;  // 2-bit codes, 16 bases / word (A,T,C,G = 0..3; exceptions 0)
    const uint32_t *mask;    // 1 bit / base: 1 = read the ASCII copy (IUPAC or lower case)
    const uint8_t *ascii;
    int64_t L;
};

struct PairDesc {
    int32_t fs;     // forward_start
    int32_t re;     // reverse_end (reverse_start = re - RL)
    uint32_t meta;  // bits 0-1 bin_fwd, 2-3 bin_rev, 16-31 attempt
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
__device__ __forceinline__ int upper_c(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
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

// phred = #(q_thr < m) for (orientation o, bin, position p); first word w0 decides unless its 27 bits
// tie with a threshold's high part, then the second word is drawn (K_QM sub 1).
__device__ __forceinline__ int quality_lookup(const DevModel &M, const Addr &a, int o, int bin, int p, uint32_t w0) {
    const uint32_t mh = w0 >> 5;
    const size_t row = ((size_t)(o * 4 + bin) * M.RL + p) * M.n_q;
    const uint32_t *hi_row = M.q_thr_hi + row;
    int lo = 0, hi = M.n_q;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (hi_row[mid] < mh) lo = mid + 1; else hi = mid;
    }
    if (lo < M.n_q && hi_row[lo] == mh) {
        const u32x4 b = draw_block(a, K_QM, (uint32_t)p, 1);
        const uint64_t m = mk53(w0, word_of(b, 2 * o));
        const uint64_t *full = M.q_thr + row;
        while (lo < M.n_q && full[lo] < m) ++lo;
    }
    return lo;
}

// mut_sequence for one base: w0 = first word of the "is it an error" uniform.
__device__ __forceinline__ int mutate_base(const DevModel &M, const Addr &a, int o, int p, int base, int q,
                                           uint32_t w0) {
    const uint32_t mh = w0 >> 5;
    const uint32_t th = M.mut_thr_hi[q];
    bool err = mh > th;
    if (mh == th) {
        const u32x4 b = draw_block(a, K_QM, (uint32_t)p, 1);
        err = mk53(w0, word_of(b, 2 * o + 1)) > M.mut_thr[q];
    }
    if (err) {
        const int bi = base_index(base);
        if (bi >= 0) {  // nucl.upper() not in "RYWSMKHBVDN"
            const u32x4 s = draw_block(a, K_SUB, (uint32_t)p, 0);
            const uint64_t m = o ? mk53(s.z, s.w) : mk53(s.x, s.y);
            const size_t row = ((size_t)(o * M.RL + p) * 4 + bi) * 3;
            int k = (m >= M.subst_thr[row]) + (m >= M.subst_thr[row + 1]);
            base = M.subst_alt[row + k];
        }
    }
    return base;
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
    d.meta = (uint32_t)bin_f | ((uint32_t)bin_r << 2) | (attempt << 16);
    d.isz = isz;
    desc[i] = d;
}

// ================================================================== k_main
__global__ __launch_bounds__(256) void k_main(DevModel M, DevGenome g, RunArgs A, const PairDesc *__restrict__ desc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_items = (uint32_t)A.n_pairs * (uint32_t)M.G;
    if (t >= n_items) return;
    const uint32_t pair = t / (uint32_t)M.G;
    const int p0 = (int)(t - pair * (uint32_t)M.G) * 4;
    const PairDesc d = desc[pair];
    const Addr a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
    const int bin_f = d.meta & 3u, bin_r = (d.meta >> 2) & 3u;
    uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int p = p0 + c;
        if (p < M.RL) {
            const u32x4 w = draw_block(a, K_QM, (uint32_t)p, 0);
            int base = read_dir_base(g, 0, d, p);
            int q = quality_lookup(M, a, 0, bin_f, p, w.x);
            base = mutate_base(M, a, 0, p, base, q, w.y);
            pk[0] |= (uint32_t)base << (8 * c);
            pk[1] |= (uint32_t)q << (8 * c);
            base = read_dir_base(g, 1, d, p);
            q = quality_lookup(M, a, 1, bin_r, p, w.z);
            base = mutate_base(M, a, 1, p, base, q, w.w);
            pk[2] |= (uint32_t)base << (8 * c);
            pk[3] |= (uint32_t)q << (8 * c);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<uint32_t *>(A.out[k])[t] = pk[k];
}

// ================================================================== k_indel_scan
// Conservative: flags mate o of a pair when some indel uniform's first 27 bits are <= the high part
// of a non-zero threshold (max over bases for deletions).  No flag  =>  provably no indel event
// (the first event in loop order would have been flagged), so k_main's output stands.
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
    for (int c = 0; c < 4; ++c) {
        const int n = grp * 4 + c;
        if (n > M.RL - 2) break;  // loop is range(read_length - 1), __init__.py:187
        if (!((amask >> c) & 1u) && !((amask >> (4 + c)) & 1u)) continue;
        u32x4 dl = {0, 0, 0, 0};
        bool have_del = false;
        for (int o = 0; o < 2; ++o) {
            if (!((amask >> (o * 4 + c)) & 1u)) continue;
            const size_t e = (size_t)o * M.RL + n;
            if (M.ins_any[e]) {
                const u32x4 w = draw_block(a, K_INS, (uint32_t)n, 2 * o);
                for (int x = 0; x < 4; ++x) {
                    const uint64_t T = M.ins_thr[e * 4 + x];
                    if (T && (word_of(w, x) >> 5) <= (uint32_t)(T >> 26)) cand |= 1u << o;
                }
            }
            const uint64_t Td = M.del_thr_max[e];
            if (Td) {
                if (!have_del) { dl = draw_block(a, K_DEL, (uint32_t)n, 0); have_del = true; }
                if (((o ? dl.z : dl.x) >> 5) <= (uint32_t)(Td >> 26)) cand |= 1u << o;
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
// Exact introduce_indels + adjust_seq_length for one flagged read, as a token transducer:
// the list prefix [0, n) is final when step n starts; the not-yet-visited suffix is
// (stack of freshly inserted letters, LIFO) ++ E(k), E(k+1), ...   (see DESIGN.md).
constexpr int FIX_STACK = 1024;  // >= read_length (checked at model upload)

__global__ __launch_bounds__(64) void k_indel_fixup(DevModel M, DevGenome g, RunArgs A, const PairDesc *__restrict__ desc,
                                                    const uint32_t *__restrict__ fix_list,
                                                    const uint32_t *__restrict__ fix_count, uint64_t *stats) {
    const uint32_t n_fix = *fix_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long *)stats, (unsigned long long)n_fix);
    uint8_t stk[FIX_STACK];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_fix; i += gridDim.x * blockDim.x) {
        const uint32_t e = fix_list[i];
        const uint32_t pair = e >> 1;
        const int o = (int)(e & 1u);
        const PairDesc d = desc[pair];
        const Addr a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
        const int RL = M.RL;
        uint8_t *out_base = A.out[2 * o] + (size_t)pair * M.pitch;
        const uint8_t *out_qual = A.out[2 * o + 1] + (size_t)pair * M.pitch;
        int top = 0, cnt = 0, k = 0, j = 0;
        auto push = [&](int v) { stk[top] = (uint8_t)v; top = (top + 1) & (FIX_STACK - 1); if (cnt < FIX_STACK) ++cnt; };
        auto pop = [&]() { top = (top - 1) & (FIX_STACK - 1); --cnt; return (int)stk[top]; };
        auto emit = [&](int base) {  // list index j is final: apply mut_sequence and store
            const u32x4 w = draw_block(a, K_QM, (uint32_t)j, 0);
            out_base[j] = (uint8_t)mutate_base(M, a, o, j, base, out_qual[j], o ? w.w : w.y);
            ++j;
        };
        for (int n = 0; n < RL - 1; ++n) {
            int tkn;
            if (cnt > 0) tkn = pop();
            else if (k < RL) tkn = read_dir_base(g, o, d, k++);
            else { emit(read_dir_base(g, o, d, k++)); continue; }  // n >= len(seq): IndexError swallowed, :223
            const int bi = base_index(tkn);
            if (bi < 0) { emit(tkn); continue; }                     // ambiguous: skipped, :190-192
            const size_t en = (size_t)o * RL + n;
            if (M.ins_any[en]) {                                     // :193-196
                const u32x4 w = draw_block(a, K_INS, (uint32_t)n, 2 * o);
                u32x4 w2 = {0, 0, 0, 0};
                bool have2 = false;
                for (int x = 0; x < 4; ++x) {
                    const uint64_t T = M.ins_thr[en * 4 + x];
                    if (!T) continue;
                    const uint32_t mh = word_of(w, x) >> 5, th = (uint32_t)(T >> 26);
                    bool hit = mh < th;
                    if (mh == th) {
                        if (!have2) { w2 = draw_block(a, K_INS, (uint32_t)n, 2 * o + 1); have2 = true; }
                        hit = mk53(word_of(w, x), word_of(w2, x)) < T;
                    }
                    if (hit) push(M.ins_letter[en * 4 + x]);
                }
            }
            bool deleted = false;
            const uint64_t Td = M.del_thr[en * 4 + bi];             // :209-210
            if (Td) {
                const u32x4 w = draw_block(a, K_DEL, (uint32_t)n, 0);
                deleted = (o ? mk53(w.z, w.w) : mk53(w.x, w.y)) < Td;
            }
            if (deleted) emit(cnt > 0 ? pop() : read_dir_base(g, o, d, k++));  // slides in unvisited
            else emit(tkn);
        }
        emit(cnt > 0 ? pop() : read_dir_base(g, o, d, k++));  // index RL-1 is never visited
    }
}

}  // namespace iss
