// iss_mt_compat.hip.h -- reference-compatible RNG mode: the two sequential MT19937 streams of the
// reference (CPython `random` and numpy's legacy global RandomState) generated and consumed ON THE
// DEVICE in the reference's exact order, so that the GPU output equals the reference's output for a
// given (seed, cpu_number) byte for byte.
//
//   k_mt_fill : one workgroup per stream; MT19937 block recurrence in four parallel phases
//               (k < 227 | 227 <= k < 454 | 454 <= k < 623 | k = 623), tempering, coalesced stores.
//   k_mt_walk : ONE wavefront per worker walks the pairs sequentially (a pair's stream offsets
//               depend on everything before it: rejection sampling in randrange, one extra numpy
//               double per substitution event); inside a pair the 64 lanes work across positions.
//               Stream consumption order is the reference's:
//                 np: insert size | py: randrange | [per mate] py: 5 doubles per visited indel step,
//                 np: bin + RL phred doubles, py: RL substitution-test doubles, np: 1 per event |
//                 py: reverse-end fallback randrange between the mates | np: gc_bias double.
//
// This mode is sequential by construction (~1e5 pairs/s): it exists for bit-identity with the
// reference, not for throughput; the Philox path (iss_kernels.hip.h) is the performance path.
#pragma once
#include "iss_kernels.hip.h"

namespace iss {

struct MtState {
    uint32_t mt[624];  // state at a block boundary (the next output needs a twist)
};

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// grid = 2 (stream 0: CPython random, stream 1: numpy), block = 256
__global__ __launch_bounds__(256) void k_mt_fill(MtState *states, uint32_t *out0, uint32_t *out1, uint32_t blocks0,
                                                 uint32_t blocks1) {
    __shared__ uint32_t buf[2][624];
    const int s = blockIdx.x;
    uint32_t *out = s ? out1 : out0;
    const uint32_t n_blocks = s ? blocks1 : blocks0;
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += 256) buf[0][k] = states[s].mt[k];
    __syncthreads();
    int cur = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t *o = buf[cur];
        uint32_t *n = buf[cur ^ 1];
        if (tid < 227) n[tid] = o[tid + 397] ^ mt_twist(o[tid], o[tid + 1]);
        __syncthreads();
        if (tid < 227) n[227 + tid] = n[tid] ^ mt_twist(o[227 + tid], o[228 + tid]);
        __syncthreads();
        if (tid < 169) n[454 + tid] = n[227 + tid] ^ mt_twist(o[454 + tid], o[455 + tid]);
        __syncthreads();
        if (tid == 0) n[623] = n[396] ^ mt_twist(o[623], n[0]);
        __syncthreads();
        for (int k = tid; k < 624; k += 256) out[(size_t)b * 624 + k] = mt_temper(n[k]);
        cur ^= 1;
        __syncthreads();
    }
    for (int k = tid; k < 624; k += 256) states[s].mt[k] = buf[cur][k];
}

struct MtWalkResult {
    uint32_t py_used, np_used;  // words consumed from each stream buffer
    int64_t n_done;             // pairs emitted
    int32_t starved;            // stopped because a stream buffer could run dry
    int32_t pad;
    int64_t n_mut;              // mutation records written (may exceed the capacity: the surplus is dropped)
    int32_t need_host;          // custom fragment length: the next pair's int(normal(...)) is too close to an
                                // integer boundary to trust the device's log(): the host evaluates it (libm)
    int32_t host_cached;        // ... from the cached second gaussian (f * x1) instead of the fresh one (f * x2)
    double host_x1, host_x2;    // the polar pair that produced it
};

// numpy's legacy gaussian state (polar Box-Muller caches its second value) + what produced the cached value
struct MtGauss {
    int32_t has_gauss, pad;
    double gauss, x1, x2;
};

struct MtWalkArgs {
    const uint32_t *py, *np;  // stream words, starting at the current consumption point
    uint32_t py_avail, np_avail;
    int64_t n_pairs;
    int32_t sequence_type, gc_bias;
    uint64_t gc_thr;
    uint8_t *out[4];
    MtWalkResult *res;
    int32_t use_rows;  // the compressed quality rows (single tile) are staged in LDS
    int32_t win_words; // LDS words for the slow indel path's stream window (10 * (RL - 1))
    MutRecord *mut;    // --store_mutations rows (NULL: off)
    int64_t mut_cap, mut_base;  // capacity; rows already written by earlier launches of this call
    int64_t pair_base;          // pair index of this launch's first pair within the call
    int32_t has_frag;           // error_model.fragment_length / fragment_sd given (generator.py:121-123)
    int32_t ov_valid;           // the first pair's fragment length was evaluated on the host
    double frag_mu, frag_sd;
    double guard;               // |x - round(x)| below this goes to the host (1e-6; tests widen it)
    int64_t ov_frag;
    MtGauss *gauss;             // persistent across launches
};

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// worst-case stream words one attempt at a pair can consume (randrange bounded at 64 words each)
__host__ __device__ inline uint32_t mt_py_need(int RL) { return 128u + 2u * (10u * (uint32_t)(RL - 1) + 2u * (uint32_t)RL); }
__host__ __device__ inline uint32_t mt_np_need(int RL) { return 64u /* polar loop */ + 2u + 2u * (2u + 4u * (uint32_t)RL) + 2u; }
__host__ __device__ inline size_t mt_walk_fixed_lds_bytes(int RL) {
    const size_t rlp = (size_t)((RL + 63) & ~63);
    return 64 * 8 /* mut_thr */ + (rlp + 64) /* tmpl */ + 3 * rlp /* read, qual, stack */ + (size_t)10 * RL * 4 /* window */;
}

// CPython _randbelow_with_getrandbits on the stream: 64 candidate words per round, first one < n wins
__device__ __forceinline__ uint32_t mt_randbelow(const uint32_t *py, uint32_t &opy, uint32_t n, int lane) {
    const int k = 32 - __clz(n);
    for (;;) {
        const uint32_t r = py[opy + lane] >> (32 - k);
        const unsigned long long ok = __ballot(r < n);
        if (ok) {
            const int t = __ffsll(ok) - 1;
            opy += (uint32_t)t + 1u;
            return (uint32_t)__shfl((int)r, t);
        }
        opy += 64u;
    }
}

__global__ __launch_bounds__(64) void k_mt_walk(DevModel M, DevGenome g, MtWalkArgs A, PairDesc *desc) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const int RL = M.RL;
    const int rlp = (RL + 63) & ~63;
    // LDS carve: [rows (optional)] [mut_thr u64 x 64] [tmpl rlp+64] [read rlp] [qual rlp] [stk rlp] [win]
    uint32_t *rows = lds;
    uint64_t *mut_thr = reinterpret_cast<uint64_t *>(lds + (A.use_rows ? M.tile_words : 0));
    uint8_t *tmpl = reinterpret_cast<uint8_t *>(mut_thr + 64);
    uint8_t *rd = tmpl + rlp + 64;
    uint8_t *ql = rd + rlp;
    uint8_t *stk = ql + rlp;
    uint32_t *win = reinterpret_cast<uint32_t *>(stk + rlp);
    if (A.use_rows)
        for (int i = lane; i < M.tile_words; i += 64) rows[i] = M.qrows[i];
    for (int i = lane; i <= M.n_q; i += 64) mut_thr[i] = M.mut_thr[i];
    __syncthreads();
    const uint32_t py_need = mt_py_need(RL), np_need = mt_np_need(RL);
    const uint32_t *py = A.py, *np = A.np;
    uint32_t opy = 0, onp = 0;
    const int64_t L = g.L;
    const uint32_t gbytes = 1u << M.GB;
    int64_t i = 0;
    int starved = 0, need_host = 0, host_cached = 0, ov_valid = A.ov_valid;
    double host_x1 = 0, host_x2 = 0;
    MtGauss gs = *A.gauss;  // wave-uniform copy; written back at the end
    int64_t n_mut = 0;  // wave-uniform
    auto put_mut = [&](int mate, int type, int pos, int ref, int alt, int qual, int64_t at) {
        if (A.mut && A.mut_base + at < A.mut_cap) {
            MutRecord r;
            r.pair = (int32_t)(A.pair_base + i); r.mate = (int8_t)mate; r.type = (int8_t)type; r.position = (int16_t)pos;
            r.ref = (uint8_t)ref; r.alt = (uint8_t)alt; r.quality = (int16_t)qual;
            A.mut[A.mut_base + at] = r;
        }
    };
    while (i < A.n_pairs) {
        const int64_t mut_mark = n_mut;
        if (opy + py_need > A.py_avail || onp + np_need > A.np_avail) { starved = 1; break; }
        const uint32_t opy0 = opy, onp0 = onp;
        const MtGauss gs0 = gs;
        int64_t isz, frag;
        if (A.has_frag) {
            // int(np.random.normal(mu, sd)): legacy polar Box-Muller with a cached second value (generator.py:122)
            double gval;
            int cached = 0;
            if (gs.has_gauss) {
                gval = gs.gauss;
                gs.has_gauss = 0;
                cached = 1;
            } else {
                double x1, x2, r2;
                do {
                    x1 = __dadd_rn(__dmul_rn(2.0, (double)mk53(np[onp], np[onp + 1]) * (1.0 / 9007199254740992.0)), -1.0);
                    x2 = __dadd_rn(__dmul_rn(2.0, (double)mk53(np[onp + 2], np[onp + 3]) * (1.0 / 9007199254740992.0)), -1.0);
                    onp += 4;
                    r2 = __dadd_rn(__dmul_rn(x1, x1), __dmul_rn(x2, x2));
                } while (r2 >= 1.0 || r2 == 0.0);
                const double f = sqrt(__ddiv_rn(__dmul_rn(-2.0, log(r2)), r2));
                gs.gauss = __dmul_rn(f, x1);
                gs.has_gauss = 1;
                gs.x1 = x1;
                gs.x2 = x2;
                gval = __dmul_rn(f, x2);
            }
            const double x = __dadd_rn(A.frag_mu, __dmul_rn(A.frag_sd, gval));
            if (ov_valid) {
                frag = A.ov_frag;
                ov_valid = 0;
            } else if (!(fabs(x) < 1e15) || fabs(x - rint(x)) < A.guard) {
                // device log() is within an ulp or two of libm's, not identical: let the host decide this one
                need_host = 1;
                host_cached = cached;
                host_x1 = cached ? gs0.x1 : gs.x1;
                host_x2 = cached ? gs0.x2 : gs.x2;
                opy = opy0; onp = onp0; gs = gs0;
                break;
            } else {
                frag = (int64_t)x;  // int(): truncation toward zero
            }
            isz = frag - 2 * (int64_t)RL;
        } else {
            // insert size: np.searchsorted(cdf, np.random.rand())  (kde.py:97)
            const uint64_t m = mk53(np[onp], np[onp + 1]);
            onp += 2;
            int cnt = 0;
            for (int j = lane; j < M.n_isize; j += 64) cnt += M.isize_thr[j] < m ? 1 : 0;
            isz = wave_sum(cnt);
            frag = isz + 2 * (int64_t)RL;
        }
        PairDesc d;
        d.isz = (int32_t)isz;
        d.meta = 0;
        int64_t fs, rs = 0, re = 0;
        if (A.sequence_type == 0) {  // generator.py:134-135, 142-144
            const int64_t width = L - frag;
            fs = mt_randbelow(py, opy, (uint32_t)(width > 0 ? width : L - RL), lane);
        } else {
            fs = 0;
        }
        const int64_t fe = fs + RL;
        d.fs = (int32_t)fs;
        for (int o = 0; o < 2; ++o) {
            // ---- template = Python slice of the genome (may be shorter than RL with odd fragment lengths),
            //      then the adjust_seq_length padding rule (__init__.py:141-155)
            int64_t lo, hi;  // normalised slice bounds
            if (o == 0) {
                lo = fs < L ? fs : L;
                hi = fe < L ? fe : L;
            } else {  // generator.py:164-177
                if (A.sequence_type == 0) { rs = fe + isz; re = rs + RL; }
                else { rs = L - RL; re = L; }
                if (re > L) { re = RL + (int64_t)mt_randbelow(py, opy, (uint32_t)(L - RL), lane); rs = re - RL; }
                lo = rs; hi = re;
                if (lo < 0) { lo += L; if (lo < 0) lo = 0; } else if (lo > L) lo = L;
                if (hi < 0) { hi += L; if (hi < 0) hi = 0; } else if (hi > L) hi = L;
            }
            if (hi < lo) hi = lo;
            const int t_len = (int)(hi - lo);  // <= RL
            auto E = [&](int k) -> int {  // token k of the read direction: template, then padding
                if (k < t_len) return o == 0 ? fetch_ascii(g, lo + k) : complement_ascii(fetch_ascii(g, hi - 1 - k));
                const int64_t i2 = k - t_len;
                if (o == 0) { const int64_t idx = fe + i2; return idx >= L ? 'A' : fetch_ascii(g, idx); }
                const int64_t idx = rs - 1 - i2;
                return idx < 0 ? 'A' : complement_ascii(fetch_ascii(g, idx < L ? idx : L - 1));
            };
            for (int k = lane; k < RL + 64; k += 64) tmpl[k] = (uint8_t)E(k);
            __syncthreads();
            const int n_visit = t_len < RL - 1 ? t_len : RL - 1;  // loop steps that find a template token
            // ---- introduce_indels: fast check (no event, no ambiguous letter => 5 doubles per step, read unchanged)
            bool any = false;
            for (int n = lane; n < n_visit; n += 64) {
                const int bi = base_index(tmpl[n]);
                const uint32_t *w = py + opy + 10u * (uint32_t)n;
                const size_t en = ((size_t)o * RL + n) * 4;
                bool hit = bi < 0;
                for (int x = 0; x < 4; ++x) hit |= mk53(w[2 * x], w[2 * x + 1]) < M.ins_thr[en + x];
                if (bi >= 0) hit |= mk53(w[8], w[9]) < M.del_thr[en + bi];
                any |= hit;
            }
            if (!__ballot(any)) {
                for (int p = lane; p < RL; p += 64) rd[p] = tmpl[p];
                opy += 10u * (uint32_t)n_visit;
            } else {
                // exact sequential list semantics (all lanes run the same walk on wave-uniform values)
                for (int k = lane; k < 10 * (RL - 1); k += 64) win[k] = py[opy + k];
                __syncthreads();
                int sp = 0, k = 0, j = 0;
                uint32_t pos = 0;
                auto src = [&](int kk) { return kk < RL + 64 ? (int)tmpl[kk] : E(kk); };
                for (int n = 0; n < RL - 1; ++n) {
                    int tok;
                    if (sp > 0) tok = stk[--sp];
                    else if (k < t_len) tok = src(k++);
                    else { if (lane == 0) rd[j] = (uint8_t)src(k); ++k; ++j; continue; }  // n >= len(seq)
                    const int bi = base_index(tok);
                    if (bi < 0) { if (lane == 0) rd[j] = (uint8_t)tok; ++j; continue; }  // ambiguous: no draws
                    const size_t en = ((size_t)o * RL + n) * 4;
                    for (int x = 0; x < 4; ++x) {
                        const uint64_t m = mk53(win[pos], win[pos + 1]);
                        pos += 2;
                        if (m < M.ins_thr[en + x]) {
                            if (sp == rlp) { if (lane == 0) for (int z = 1; z < sp; ++z) stk[z - 1] = stk[z]; --sp; }
                            if (lane == 0) stk[sp] = M.ins_letter[en + x];
                            ++sp;
                            __syncthreads();
                            if (lane == 0) put_mut(o, 1, n, tok, M.ins_letter[en + x], -1, n_mut);  // ref = seq[position]
                            ++n_mut;
                        }
                    }
                    const uint64_t m = mk53(win[pos], win[pos + 1]);
                    pos += 2;
                    if (m < M.del_thr[en + bi]) {  // next token slides in
                        const bool exists = sp > 0 || k < t_len;  // else mutable_seq[position] raises IndexError: no row
                        tok = sp > 0 ? (int)stk[--sp] : src(k++);
                        if (exists) { if (lane == 0) put_mut(o, 2, n, tok, '.', -1, n_mut); ++n_mut; }
                    }
                    if (lane == 0) rd[j] = (uint8_t)tok;
                    ++j;
                }
                if (lane == 0) rd[j] = (uint8_t)(sp > 0 ? (int)stk[sp - 1] : src(k));
                opy += pos;
            }
            __syncthreads();
            // ---- gen_phred_scores: bin choice + one CDF inversion per position (kde.py:72-85)
            int bin;
            {
                const uint64_t m = mk53(np[onp], np[onp + 1]);
                onp += 2;
                bin = count_le(M.bin_thr + 4 * o, 4, m);
                bin = bin > 3 ? 3 : bin;
            }
            const int slot = M.bin_slot[o * 4 + bin] & 3;
            d.meta |= (uint32_t)slot << (2 * o);
            for (int p = lane; p < RL; p += 64) {
                const uint64_t m = mk53(np[onp + 2u * (uint32_t)p], np[onp + 2u * (uint32_t)p + 1]);
                int q;
                const uint64_t *full = M.q_thr + ((size_t)(o * 4 + bin) * RL + p) * M.n_q;
                if (A.use_rows) {
                    const uint32_t h = (uint32_t)(m >> 37);
                    const uint32_t row = ((uint32_t)(o * M.NB + slot) * (uint32_t)M.TG + (uint32_t)(p >> 2)) * (uint32_t)M.GS +
                                         (uint32_t)(p & 3) * (uint32_t)M.stride_w;
                    uint32_t j = reinterpret_cast<const uint8_t *>(rows)[row * 4 + (h >> (16 - M.GB))];
                    uint32_t e = rows[row + gbytes / 4 + j];
                    while ((e >> 15) < h) e = rows[row + gbytes / 4 + (++j)];
                    q = (int)((e >> 2) & 0xffu);
                    if ((e >> 15) == h) q = count_lt(full, M.n_q, m);
                } else {
                    q = count_lt(full, M.n_q, m);
                }
                ql[p] = (uint8_t)q;
            }
            onp += 2u * (uint32_t)RL;
            __syncthreads();
            // ---- mut_sequence: one py double per position, one np double per substitution event, in order
            uint8_t *ob = A.out[2 * o] + (size_t)i * M.pitch;
            uint8_t *oq = A.out[2 * o + 1] + (size_t)i * M.pitch;
            uint32_t nev = 0;
            for (int p0 = 0; p0 < RL; p0 += 64) {
                const int p = p0 + lane;
                bool err = false, rec = false;
                int ch = 0, q = 0, bi = -1, ref_ch = 0;
                if (p < RL) {
                    ch = rd[p];
                    q = ql[p];
                    bi = base_index(ch);
                    const uint64_t m = mk53(py[opy + 2u * (uint32_t)p], py[opy + 2u * (uint32_t)p + 1]);
                    err = m > mut_thr[q] && bi >= 0;
                }
                const unsigned long long evm = __ballot(err);
                if (err) {
                    const uint32_t rank = nev + (uint32_t)__popcll(evm & ((1ull << lane) - 1ull));
                    const uint64_t ms = mk53(np[onp + 2u * rank], np[onp + 2u * rank + 1]);
                    const size_t row = ((size_t)(o * RL + p) * 4 + bi) * 3;
                    const int k = (ms >= M.subst_thr[row]) + (ms >= M.subst_thr[row + 1]);
                    const int alt = M.subst_alt[row + k];
                    rec = alt != (int)tmpl[p];  // only if it differs from the ORIGINAL read (:98)
                    ref_ch = ch;
                    ch = alt;
                }
                const unsigned long long recm = __ballot(rec);
                if (rec) put_mut(o, 0, p, ref_ch, ch, q, n_mut + (int64_t)__popcll(recm & ((1ull << lane) - 1ull)));
                n_mut += (int64_t)__popcll(recm);
                nev += (uint32_t)__popcll(evm);
                if (p < RL) { ob[p] = (uint8_t)ch; oq[p] = (uint8_t)q; }
            }
            opy += 2u * (uint32_t)RL;
            onp += 2u * nev;
            __syncthreads();
        }
        d.re = (int32_t)re;
        bool keep = true;
        if (A.gc_bias) {  // generator.py:82-92
            keep = mk53(np[onp], np[onp + 1]) < A.gc_thr;
            onp += 2;
        }
        if (keep) {
            if (lane == 0) desc[i] = d;
            ++i;
        } else {
            n_mut = mut_mark;  // the rejected pair's mutations are not yielded (generator.py:88-92)
        }
    }
    if (lane == 0) {
        A.res->py_used = opy;
        A.res->np_used = onp;
        A.res->n_done = i;
        A.res->starved = starved;
        A.res->n_mut = n_mut;
        A.res->need_host = need_host;
        A.res->host_cached = host_cached;
        A.res->host_x1 = host_x1;
        A.res->host_x2 = host_x2;
        *A.gauss = gs;
    }
}

}  // namespace iss
