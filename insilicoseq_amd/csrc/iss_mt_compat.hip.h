// iss_mt_compat.hip.h -- reference-compatible RNG mode: the two sequential MT19937 streams of the
// reference (CPython `random` and numpy's legacy global RandomState) generated and consumed ON THE
// DEVICE in the reference's exact order, so that the GPU output equals the reference's output for a
// given (seed, cpu_number) byte for byte.
//
//   k_mt_fill : one workgroup per stream; every word of a 624-word block written in terms of the previous
//               block (one barrier per block), tempered and stored by its producer.
//   k_mt_walk : ONE wavefront per worker walks the pairs sequentially (a pair's stream offsets
//               depend on everything before it: rejection sampling in randrange, one extra numpy
//               double per substitution event); inside a pair the 64 lanes work across positions.
//               Stream consumption order is the reference's:
//                 np: insert size | py: randrange | [per mate] py: 5 doubles per visited indel step,
//                 np: bin + RL phred doubles, py: RL substitution-test doubles, np: 1 per event |
//                 py: reverse-end fallback randrange between the mates | np: gc_bias double.
//
//   k_mt_resolve / k_mt_emit : the fast path for plain pairs (see below): only the stream offsets are
//               chained, by one workgroup working from LDS; the reads are then built in parallel.
//
//   k_mt_fill_w / k_mt_resolve_w / k_mt_walk_w / k_mt_emit_w / k_mt_move_w (round 5): the same bodies for W workers per
//               launch -- the reference's own parallelism is N workers with seeds seed + cpu_number (iss/app.py:99-106,
//               iss/generator.py:234-236): one workgroup (wavefront, grid row) per worker, the jobs in a table in HBM.
//
// This mode is chained by construction (~3.8e5 pairs/s per worker, 5.1e7 with 256 workers side by side): it exists for
// bit-identity with the reference; the Philox path (iss_kernels.hip.h) is the performance path.
#pragma once
#include "iss_kernels.hip.h"

namespace iss {

struct MtState {
    uint32_t mt[624];  // state at a block boundary (the next output needs a twist)
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. it
// would wait for the global stores / prefetch loads these kernels deliberately leave in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A pointer read from a job table in HBM is a GENERIC pointer to the compiler: its loads and stores become flat_* instructions,
// which count on the LDS counter as well -- lds_barrier() above then waits for the stream words the resolver deliberately leaves
// in flight, and every access pays the flat path (found in round 5 in the ISA: 37 flat operations in k_mt_resolve_w, 148 in
// k_mt_walk_w, none in the single-worker kernels, whose pointers are kernel arguments).  What does turn them into global_*
// accesses with this compiler: the pointer rebuilt as (a kernel argument -- the job table itself, known to be global) + a byte
// distance, the distance hidden from the optimiser (which folds X + (Y - X) back into Y) behind an empty asm; a cast through
// address_space(1) and back, and an assumption "neither shared nor private", were both folded away (tools/: none needed --
// `grep -c flat_` on the kernel's ISA).  Costs four scalar instructions per pointer, once per workgroup.  Job pointers are
// uniform (indexed by blockIdx): the distance lives in scalar registers.
template <class T, class A> __device__ __forceinline__ T *as_global(T *p, const A *anchor) {
    const long d0 = (long)p - (long)anchor;
    int lo = __builtin_amdgcn_readfirstlane((int)d0), hi = __builtin_amdgcn_readfirstlane((int)(d0 >> 32));  // (a large job is copied through VGPRs)
    asm volatile("" : "+s"(lo), "+s"(hi));
    const long d = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
    return (T *)((const char *)anchor + d);
}
template <class A> __device__ __forceinline__ DevGenome genome_as_global(DevGenome g, const A *anchor) {
    g.packed = as_global(g.packed, anchor); g.mask = as_global(g.mask, anchor); g.ascii = as_global(g.ascii, anchor);
    return g;
}

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

// grid = 2 (stream 0: CPython random, stream 1: numpy), block = 320: four wavefronts produce words 0..622 (three
// per lane), the fifth only word 623 -- its chain of old words is as long as the others' work, so it runs beside
// them instead of behind one of them (measured: 1.08 s vs 1.31 s per 2.1e9 words with lane 169 doing both).
// Every word of the next block is written in terms of the OLD block only, so a block costs one barrier:
// with F(k) = twist(o[k], o[k+1]),
//   n[k]       = o[k+397] ^ F(k)                               k < 227
//   n[227 + j] = n[j] ^ F(227 + j) = o[j+397] ^ F(j) ^ F(227 + j)        j < 227
//   n[454 + j] = n[227 + j] ^ F(454 + j)                                  j < 169
//   n[623]     = n[396] ^ twist(o[623], n[0])
constexpr int FILL_THREADS = 320;
// (one stream: its state, where its next n_blocks blocks of 624 tempered words go)
__device__ __forceinline__ void mt_fill_body(MtState *state, uint32_t *out, uint32_t n_blocks) {
    __shared__ uint32_t buf[2][624];
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += FILL_THREADS) buf[0][k] = state->mt[k];
    __syncthreads();
    int cur = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t *o = buf[cur];
        uint32_t *n = buf[cur ^ 1];
        uint32_t *ob = out + (size_t)b * 624;
        if (tid < 227) {
            const uint32_t a = o[tid + 397] ^ mt_twist(o[tid], o[tid + 1]);          // n[tid]
            const uint32_t c = a ^ mt_twist(o[227 + tid], o[228 + tid]);             // n[227 + tid]
            n[tid] = a;
            n[227 + tid] = c;
            ob[tid] = mt_temper(a);
            ob[227 + tid] = mt_temper(c);
            if (tid < 169) {
                const uint32_t e = c ^ mt_twist(o[454 + tid], o[455 + tid]);         // n[454 + tid]
                n[454 + tid] = e;
                ob[454 + tid] = mt_temper(e);
            }
        } else if (tid == 256) {
            const uint32_t n0 = o[397] ^ mt_twist(o[0], o[1]);
            const uint32_t n169 = o[566] ^ mt_twist(o[169], o[170]);
            const uint32_t n396 = n169 ^ mt_twist(o[396], o[397]);
            const uint32_t e = n396 ^ mt_twist(o[623], n0);
            n[623] = e;
            ob[623] = mt_temper(e);
        }
        cur ^= 1;
        lds_barrier();
    }
    for (int k = tid; k < 624; k += FILL_THREADS) state->mt[k] = buf[cur][k];
}
__global__ __launch_bounds__(FILL_THREADS) void k_mt_fill(MtState *states, uint32_t *out0, uint32_t *out1, uint32_t blocks0,
                                                          uint32_t blocks1) {
    const int s = blockIdx.x;
    mt_fill_body(states + s, s ? out1 : out0, s ? blocks1 : blocks0);
}

// ---- W workers per launch (round 5).  The reference's own parallelism is N independent workers, each with its two streams
// seeded seed + cpu_number (iss/generator.py:234-236, iss/app.py:99-106): a worker is one chain, W workers are W chains --
// one workgroup each per kernel of the path, a job table in global memory instead of kernel arguments.  Every *_w kernel
// runs exactly the single-worker kernel's code on its worker's job; a job with nothing to do leaves at once.
struct MtFillJob {
    MtState *state;
    uint32_t *out;
    uint32_t n_blocks, pad;
};
__global__ __launch_bounds__(FILL_THREADS) void k_mt_fill_w(const MtFillJob *jobs) {
    const MtFillJob j = jobs[blockIdx.x];
    if (!j.n_blocks) return;  // (uniform)
    mt_fill_body(as_global(j.state, jobs), as_global(j.out, jobs), j.n_blocks);
}
// the unconsumed words of a stream move in front of the words produced ahead (the single-worker path: a device-to-device copy
// per stream and turn; 2 W of them would be 2 W launches)
struct MtMoveJob {
    const uint32_t *src;
    uint32_t *dst;
    uint32_t n, pad;
};
// (A worker whose resolver stopped early -- a pair for the walker -- leaves nearly a whole turn's words to move: tens of MB
//  for ONE job.  64 workgroups per job, eight independent loads per lane in flight: measured 0.9 ms -> ~0.1 ms for that tail.)
constexpr int MOVE_BLOCKS = 64;  // workgroups per job (grid.y)
constexpr int MOVE_UNROLL = 8;
__global__ __launch_bounds__(256) void k_mt_move_w(const MtMoveJob *jobs) {
    const MtMoveJob j = jobs[blockIdx.x];
    const uint32_t *__restrict__ src = as_global(j.src, jobs);
    uint32_t *__restrict__ dst = as_global(j.dst, jobs);
    const uint32_t span = 256u * (uint32_t)MOVE_UNROLL;
    for (uint32_t base = blockIdx.y * span; base < j.n; base += span * (uint32_t)MOVE_BLOCKS) {  // (uniform per workgroup)
        uint32_t v[MOVE_UNROLL];
#pragma unroll
        for (int u = 0; u < MOVE_UNROLL; ++u) {
            const uint32_t k = base + (uint32_t)u * 256u + threadIdx.x;
            v[u] = k < j.n ? src[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < MOVE_UNROLL; ++u) {
            const uint32_t k = base + (uint32_t)u * 256u + threadIdx.x;
            if (k < j.n) dst[k] = v[u];
        }
    }
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
    int32_t n_amb;              // need_host == 2 (BasicErrorModel): phred scores too close to a rounding boundary for the
    int32_t pad2;               // device's log / log10: MtWalkArgs::amb[0 .. n_amb) lists them, the host evaluates them (libm)
};

// a BasicErrorModel phred the host has to round (need_host == 2), and the host's answer on the relaunch
struct MtPhredAmb {
    int32_t mate, pos, cached, q;  // cached: the value is the second one (f * x1) of its polar pair; q: the answer
    double x1, x2;                 // the accepted polar candidate
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
    int32_t use_rows;  // the 16-bit digit rows (DevModel::mt_rows) are staged in LDS
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
    MtPhredAmb *amb;            // out: ambiguous phreds of the pair that stopped the walk (capacity MT_AMB_CAP)
    const MtPhredAmb *ovq;      // in: the host's answers for the FIRST pair of this launch
    int32_t n_ovq;
};
constexpr int MT_AMB_CAP = 512;

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// worst-case stream words one attempt at a pair can consume (randrange bounded at 64 words each)
// (a randrange round: 64 candidates of one word, of two words once the bound passes 2^32 -- records of 2^32 bases and more)
__host__ __device__ inline uint32_t mt_py_need(int RL) { return 256u + 2u * (10u * (uint32_t)(RL - 1) + 2u * (uint32_t)RL); }
__host__ __device__ inline uint32_t mt_np_need(int RL, bool basic = false) {
    // basic: the phreds of a mate are RL gaussians = RL/2 accepted polar candidates of 4 words (acceptance pi/4, expected
    // 2.55 * RL words); 8 * RL + 1024 is tens of standard deviations above that (k_mt_walk also checks every round)
    return 64u /* polar loop */ + 2u + 2u * (2u + 4u * (uint32_t)RL) + 2u + (basic ? 2u * (8u * (uint32_t)RL + 1024u) : 0u);
}
__host__ __device__ inline size_t mt_walk_fixed_lds_bytes(int RL) {
    const size_t rlp = (size_t)((RL + 63) & ~63);
    return 64 * 8 /* mut_thr */ + (rlp + 64) /* tmpl */ + 3 * rlp /* read, qual, stack */ + (size_t)10 * RL * 4 /* window */ +
           (size_t)8 * RL * 8 /* the slow indel path's thresholds of one mate */;
}

// CPython _randbelow_with_getrandbits on the stream: 64 candidate words per round, first one < n wins
// getrandbits(k) for k > 32 (_randommodule.c): 32-bit words from the least significant on, the top one shifted -- records of 2^32
// bases and more (round 5: MT mode takes records up to MAX_RECORD like the Philox path; the reference spills them to a memmap and
// carries on, iss/generator.py:313-331); a candidate is then two words
__device__ __forceinline__ uint64_t mt_randbelow(const uint32_t *py, uint32_t &opy, uint64_t n, int lane) {
    const int k = 64 - __clzll((long long)n);
    for (;;) {
        uint64_t r;
        if (k <= 32) r = py[opy + lane] >> (32 - k);
        else r = (uint64_t)py[opy + 2 * lane] | ((uint64_t)(py[opy + 2 * lane + 1] >> (64 - k)) << 32);
        const unsigned long long ok = __ballot(r < n);
        const uint32_t per = k <= 32 ? 1u : 2u;
        if (ok) {
            const int t = __ffsll(ok) - 1;
            opy += per * ((uint32_t)t + 1u);
            return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(r >> 32), t) << 32) | (uint32_t)__shfl((int)(uint32_t)r, t);
        }
        opy += per * 64u;
    }
}

// One phred: #(cdf < u) of position p's quality CDF (kde.py:72-85) through the 16-bit digit rows (DevModel::mt_rows; `drows`
// may be a copy in LDS): the leading digits first -- every 7th key, then the segment: two rounds of independent reads --, the full
// 53-bit thresholds only among those that share the draw's digit.
__device__ __forceinline__ int mt_phred_of(const DevModel &M, const uint16_t *drows, uint32_t row_h, int o, int slot, int bin, int p, uint64_t m) {
    const uint32_t h = (uint32_t)(m >> 37);
    const uint16_t *row = drows + ((size_t)(o * M.NB + slot) * M.RL + p) * row_h;
    const int nq = M.n_q;
    int c1 = 0;
    bool tie = false;
#pragma unroll
    for (int jk = 0; jk < 9; ++jk) {  // keys 6, 13, ... (n_q <= 60); indices past n_q read the 0xffff padding
        const uint32_t dgt = row[min(6 + 7 * jk, nq)];
        c1 += dgt < h ? 1 : 0;
        tie |= dgt == h;
    }
    int q = 7 * c1;
#pragma unroll
    for (int jk = 0; jk < 6; ++jk) {
        const uint32_t dgt = row[min(7 * c1 + jk, nq)];
        q += dgt < h ? 1 : 0;
        tie |= dgt == h;
    }
    if (tie) {
        const uint64_t *full = M.q_thr + ((size_t)(o * 4 + bin) * M.RL + p) * nq;
        while (q < nq && full[q] < m) ++q;
    }
    return q;
}

__device__ __forceinline__ void mt_walk_body(const DevModel &M, const DevGenome &g, const MtWalkArgs &A, PairDesc *desc) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const int RL = M.RL;
    const int rlp = (RL + 63) & ~63;
    // LDS carve: [rows (optional)] [mut_thr u64 x 64] [tmpl rlp+64] [read rlp] [qual rlp] [stk rlp] [win 10 RL] [thr u64 x 8 RL]
    uint32_t *rows = lds;
    const uint32_t rows_words = (uint32_t)(2 * M.NB * RL * M.mt_row_w);
    uint64_t *mut_thr = reinterpret_cast<uint64_t *>(lds + (A.use_rows ? ((rows_words + 1u) & ~1u) : 0u));
    uint8_t *tmpl = reinterpret_cast<uint8_t *>(mut_thr + 64);
    uint8_t *rd = tmpl + rlp + 64;
    uint8_t *ql = rd + rlp;
    uint8_t *stk = ql + rlp;
    uint32_t *win = reinterpret_cast<uint32_t *>(stk + rlp);
    uint64_t *thr_l = reinterpret_cast<uint64_t *>(win + 10 * RL);  // per loop step: ins_thr x 4, del_thr x 4 (8-byte aligned: all of the above are)
    if (A.use_rows) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(M.mt_rows);
        for (uint32_t i = lane; i < rows_words; i += 64) rows[i] = src[i];
    }
    const uint16_t *drows = A.use_rows ? reinterpret_cast<const uint16_t *>(rows) : M.mt_rows;
    const uint32_t row_h = (uint32_t)M.mt_row_w * 2u;  // u16 entries per row
    for (int i = lane; i <= M.n_q; i += 64) mut_thr[i] = M.mut_thr[i];
    __syncthreads();
    const uint32_t py_need = mt_py_need(RL), np_need = mt_np_need(RL, M.quality_mode == 1);
    const uint32_t *py = A.py, *np = A.np;
    uint32_t opy = 0, onp = 0;
    const int64_t L = g.L;
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
    int n_amb = 0, abort_pair = 0;  // wave-uniform
    // BasicErrorModel.gen_phred_scores (basic.py:40-54): RL values of np.random.normal(mean, sd) -- numpy's legacy polar
    // Box-Muller: candidates (x1, x2) of two doubles each until 0 < r2 < 1; an accepted candidate yields f * x2, then
    // (cached for the next call, across mates and pairs) f * x1 -- capped at 0.9999, then
    // prob_to_phred = int(round(-10 * log10(1 - p))) (util.py:44).  64 candidates per round, one per lane; ranks by
    // ballot.  The device's log / log10 are not libm's: a value within `guard` of a rounding boundary is not decided
    // here but listed for the host (returns 2; the relaunch brings the answers).  Returns 1 when the stream runs dry.
    auto basic_phreds = [&](int o, bool first_of_launch) -> int {
        const double mean = M.basic_mean, sd = M.basic_sd, cap = M.basic_cap;
        int amb_here = 0;
        auto phred_of = [&](double gval, double x1, double x2, int cached, int pos, bool active) {
            double p = __dadd_rn(mean, __dmul_rn(sd, gval));
            if (p > cap) p = cap;  // min(q, 0.9999)
            const double x = __dmul_rn(-10.0, log10(__dadd_rn(1.0, -p)));
            const double r = rint(x);
            int q = (int)r;
            bool amb = active && !(fabs(x - r) < 0.5 - A.guard);
            if (amb && first_of_launch)
                for (int k = 0; k < A.n_ovq; ++k)
                    if (A.ovq[k].mate == o && A.ovq[k].pos == pos) { q = A.ovq[k].q; amb = false; }
            const unsigned long long am = __ballot(amb);
            if (amb) {
                const int at = n_amb + amb_here + (int)__popcll(am & ((1ull << lane) - 1ull));
                if (at < MT_AMB_CAP) {
                    MtPhredAmb e;
                    e.mate = o; e.pos = pos; e.cached = cached; e.q = 0; e.x1 = x1; e.x2 = x2;
                    A.amb[at] = e;
                }
            }
            amb_here += (int)__popcll(am);
            if (active) ql[pos] = (uint8_t)q;
        };
        int j = 0;
        if (gs.has_gauss) {  // the cached second value of an earlier candidate comes first
            phred_of(gs.gauss, gs.x1, gs.x2, 1, 0, lane == 0);
            gs.has_gauss = 0;
            j = 1;
        }
        while (j < RL) {
            if (onp + 256u > A.np_avail) return 1;
            const uint32_t *w = np + onp + 4u * (uint32_t)lane;
            const double x1 = __dadd_rn(__dmul_rn(2.0, (double)mk53(w[0], w[1]) * (1.0 / 9007199254740992.0)), -1.0);
            const double x2 = __dadd_rn(__dmul_rn(2.0, (double)mk53(w[2], w[3]) * (1.0 / 9007199254740992.0)), -1.0);
            const double r2 = __dadd_rn(__dmul_rn(x1, x1), __dmul_rn(x2, x2));
            const bool ok = !(r2 >= 1.0 || r2 == 0.0);
            const unsigned long long m = __ballot(ok);
            const int k = (int)__popcll(m & ((1ull << lane) - 1ull));  // rank among the accepted candidates
            const int want = (RL - j + 1) >> 1;                           // accepted candidates still needed
            const int acc = (int)__popcll(m);
            const bool use = ok && k < want;
            const double f = use ? sqrt(__ddiv_rn(__dmul_rn(-2.0, log(r2)), r2)) : 0.0;
            const double v0 = __dmul_rn(f, x2), v1 = __dmul_rn(f, x1);
            phred_of(v0, x1, x2, 0, j + 2 * k, use && j + 2 * k < RL);
            phred_of(v1, x1, x2, 1, j + 2 * k + 1, use && j + 2 * k + 1 < RL);
            if (acc >= want) {
                const int last = __ffsll((unsigned long long)__ballot(ok && k == want - 1)) - 1;  // lane of the last one used
                onp += 4u * (uint32_t)(last + 1);
                if ((RL - j) & 1) {  // its second value is left over: cached for the next call
                    gs.has_gauss = 1;
                    gs.gauss = __shfl(v1, last);
                    gs.x1 = __shfl(x1, last);
                    gs.x2 = __shfl(x2, last);
                }
                j = RL;
            } else {
                onp += 256u;
                j += 2 * acc;
            }
        }
        n_amb += amb_here;
        return amb_here ? 2 : 0;
    };
    while (i < A.n_pairs) {
        const int64_t mut_mark = n_mut;
        if (opy + py_need > A.py_avail || onp + np_need > A.np_avail) { starved = 1; break; }
        n_amb = 0;
        abort_pair = 0;
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
        } else if (M.quality_mode == 1) {
            isz = M.basic_insert_size;  // BasicErrorModel.random_insert_size: a constant, no draw (basic.py:56-63)
            frag = isz + 2 * (int64_t)RL;
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
            fs = (int64_t)mt_randbelow(py, opy, (uint64_t)(width > 0 ? width : L - RL), lane);
        } else {
            fs = 0;
        }
        const int64_t fe = fs + RL;
        d.fs = (int32_t)(uint32_t)fs;  // (low 32 bits; bits 32-35 of both coordinates go into meta with the reverse end below)
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
                if (re > L) { re = RL + (int64_t)mt_randbelow(py, opy, (uint64_t)(L - RL), lane); rs = re - RL; }
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
                // the thresholds of this mate's loop steps too: the walk below is ONE chain of RL - 1 dependent steps, and a step
                // that waits for two loads from HBM / the L2 (an insertion row, then the deletion entry of the token it found)
                // took 0.3 us by itself and 2 us beside a set's fill kernels -- 0.56 ms per walked pair (round 5, profiles/
                // r05_mtset_kernel_stats.csv); from LDS a step is two LDS round trips
                for (int k = lane; k < 4 * (RL - 1); k += 64) {
                    thr_l[(k >> 2) * 8 + (k & 3)] = M.ins_thr[(size_t)o * RL * 4 + k];
                    thr_l[(k >> 2) * 8 + 4 + (k & 3)] = M.del_thr[(size_t)o * RL * 4 + k];
                }
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
                    const uint64_t *tl = thr_l + n * 8;
                    for (int x = 0; x < 4; ++x) {
                        const uint64_t m = mk53(win[pos], win[pos + 1]);
                        pos += 2;
                        if (m < tl[x]) {
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
                    if (m < tl[4 + bi]) {  // next token slides in
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
            if (M.quality_mode == 1) {
                abort_pair = basic_phreds(o, opy0 == 0u && onp0 == 0u);
                if (!abort_pair && onp + 2u * (uint32_t)RL + 4u > A.np_avail) abort_pair = 1;  // room for the substitution picks
                if (abort_pair) break;
                __syncthreads();
            } else {
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
                const int q = mt_phred_of(M, drows, row_h, o, slot, bin, p, m);
                ql[p] = (uint8_t)q;
            }
            onp += 2u * (uint32_t)RL;
            __syncthreads();
            }
            // ---- mut_sequence: one py double per position, one np double per substitution event, in order
            uint8_t *ob = A.out[2 * o] + (size_t)i * M.row;
            uint8_t *oq = A.out[2 * o + 1] + (size_t)i * M.row;
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
                    rec = p >= t_len || alt != (int)tmpl[p];  // only if it differs from the ORIGINAL read (:98); past a cut-short
                                                              // template the reference raises IndexError: the row is kept
                    ref_ch = ch;
                    ch = alt;
                }
                const unsigned long long recm = __ballot(rec);
                if (rec) put_mut(o, 0, p, ref_ch, ch, q, n_mut + (int64_t)__popcll(recm & ((1ull << lane) - 1ull)));
                n_mut += (int64_t)__popcll(recm);
                nev += (uint32_t)__popcll(evm);
                if (p < RL) { ob[xp(p)] = (uint8_t)ch; oq[xp(p)] = (uint8_t)q; }
            }
            opy += 2u * (uint32_t)RL;
            onp += 2u * nev;
            __syncthreads();
        }
        if (abort_pair) {  // BasicErrorModel: out of stream words (1) or a phred the host has to round (2)
            if (abort_pair == 1) starved = 1; else need_host = 2;
            opy = opy0; onp = onp0; gs = gs0; n_mut = mut_mark;
            break;
        }
        d.re = (int32_t)(uint32_t)re;
        d.meta |= desc_hi_bits(fs, re);
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
        A.res->n_amb = n_amb;
        *A.gauss = gs;
    }
}
__global__ __launch_bounds__(64) void k_mt_walk(DevModel M, DevGenome g, MtWalkArgs A, PairDesc *desc) { mt_walk_body(M, g, A, desc); }
struct MtWalkJob {
    MtWalkArgs A;
    DevGenome g;
    PairDesc *desc;
};
__global__ __launch_bounds__(64) void k_mt_walk_w(DevModel M, const MtWalkJob *jobs) {
    MtWalkJob j = jobs[blockIdx.x];
    if (j.A.n_pairs <= 0) return;  // (uniform: this worker has no turn of the walker)
    j.A.py = as_global(j.A.py, jobs); j.A.np = as_global(j.A.np, jobs); j.A.res = as_global(j.A.res, jobs); j.A.gauss = as_global(j.A.gauss, jobs);
#pragma unroll
    for (int k = 0; k < 4; ++k) j.A.out[k] = as_global(j.A.out[k], jobs);
    j.A.mut = as_global(j.A.mut, jobs); j.A.amb = as_global(j.A.amb, jobs); j.A.ovq = as_global(j.A.ovq, jobs);
    mt_walk_body(M, genome_as_global(j.g, jobs), j.A, as_global(j.desc, jobs));
}

// ====================================================================== resolver + emitter
// The stream offsets are the only thing that chains the pairs of a worker: once a pair's offsets are known
// the pair itself is ordinary parallel work.  k_mt_resolve walks the pairs computing ONLY the offsets: per
// mate, one lane per position inverts the quality CDF, runs the substitution test and looks for indel
// candidates, all from LDS (stream words in two-half rings refilled through registers one half ahead,
// digit rows, thresholds); one workgroup barrier per mate sums the substitution events, which is all the
// next mate's offsets need.  It stops (need_generic) at the first pair that is not plain -- an indel
// candidate, a letter outside ACGT in a template, a randrange that needs a second round of words -- and
// the host runs k_mt_walk for exactly that pair.  k_mt_emit then builds the reads of the resolved pairs,
// one wavefront per mate, from the recorded offsets.
struct MtPairRec {
    uint32_t opy_err[2];  // py words: first substitution-test double of mate o
    uint32_t onp_bin[2];  // np words: bin draw of mate o (then RL phred doubles, then the substitution picks)
};

struct MtResolveArgs {
    const uint32_t *py_base, *np_base;  // stream buffers (16-byte aligned)
    uint32_t py_off, np_off;            // consumption point (words from the base)
    uint32_t py_fill, np_fill;          // valid words
    uint32_t py_cap, np_cap;            // allocated words
    int64_t n_pairs;
    int32_t sequence_type, gc_bias;
    uint64_t gc_thr;
    MtWalkResult *res;
    MtPairRec *rec;
    int32_t has_frag;           // custom fragment length: int(np.random.normal(mu, sd)) instead of the insert-size CDF
    double frag_mu, frag_sd, guard;
    MtGauss *gauss;             // numpy's cached second gaussian, shared with k_mt_walk
};

constexpr int RES_THREADS = 512;  // two groups of four wavefronts: the mates of a pair are worked on side by side
// the rings must show a whole pair at once (both mates are read concurrently)
__host__ __device__ inline uint32_t mt_res_need_py(int RL) { return 256u + 2u * (10u * (uint32_t)(RL - 1) + 2u * (uint32_t)RL); }
__host__ __device__ inline uint32_t mt_res_need_np(int RL) { return 16u + 64u /* polar loop */ + 8u * (uint32_t)RL; }
__host__ __device__ inline size_t mt_res_lds_bytes(const DevModel &M, int pyv, int npv, bool rows_lds) {
    size_t b = (size_t)(2 * pyv + 2 * npv) * 1024 * 4 + 2 * 16 * 4;  // rings + their 16-word mirrors
    b += (size_t)(64 + 8 + M.n_isize) * 8;                           // mut_thr, bin_thr, isize_thr
    b += (size_t)2 * M.RL * 5 * 4 + 16 * 4;                          // indel limits, scratch
    if (rows_lds) b += (size_t)2 * M.NB * M.RL * M.mt_row_w * 4;     // digit rows
    return b;
}

template <int PYV, int NPV, bool ROWS_LDS>
__device__ __forceinline__ void mt_resolve_body(const DevModel &M, const DevGenome &g, const MtResolveArgs &A, PairDesc *desc) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t HPY = PYV * 1024u, WPY = 2u * HPY, HNP = NPV * 1024u, WNP = 2u * HNP;
    constexpr uint32_t CHUNK = RES_THREADS * 4u;  // words one load / store instruction of the workgroup moves
    static_assert(HPY % CHUNK == 0 && HNP % CHUNK == 0, "ring halves are whole chunks");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gw = wave & 3, tid_g = tid & 255;  // mate group, wavefront and lane index inside it
    const int RL = M.RL, nq = M.n_q;
    // ring[W .. W + 16) mirrors ring[0 .. 16): a run of <= 16 words starting anywhere needs no wrap-around
    uint32_t *ring_py = lds;
    uint32_t *ring_np = ring_py + WPY + 16;
    uint64_t *mut_thr = reinterpret_cast<uint64_t *>(ring_np + WNP + 16);
    uint64_t *bin_thr = mut_thr + 64;
    uint64_t *isz_thr = bin_thr + 8;
    uint32_t *lim = reinterpret_cast<uint32_t *>(isz_thr + M.n_isize);
    uint32_t *scratch = lim + 2 * RL * 5;
    uint16_t *rows_l = reinterpret_cast<uint16_t *>(scratch + 16);
    const uint32_t row_h = (uint32_t)M.mt_row_w * 2u;  // u16 entries per row
    for (int i = tid; i <= nq; i += RES_THREADS) mut_thr[i] = M.mut_thr[i];
    for (int i = tid; i < 8; i += RES_THREADS) bin_thr[i] = M.bin_thr[i];
    for (int i = tid; i < M.n_isize; i += RES_THREADS) isz_thr[i] = M.isize_thr[i];
    for (int i = tid; i < 2 * RL * 5; i += RES_THREADS) lim[i] = M.mt_lim[i];
    if (ROWS_LDS) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(M.mt_rows);
        uint32_t *dst = reinterpret_cast<uint32_t *>(rows_l);
        for (int i = tid; i < 2 * M.NB * RL * M.mt_row_w; i += RES_THREADS) dst[i] = src[i];
    }
    // ---- stream rings: halves kpy, kpy + 1 are in the ring, half kpy + 2 is on its way in registers
    constexpr int PYL = (int)(HPY / CHUNK), NPL = (int)(HNP / CHUNK);
    uint4 pv[PYL], nv[NPL];
    auto fetch_py = [&](uint32_t half) {
#pragma unroll
        for (int v = 0; v < PYL; ++v) {
            const uint32_t idx = half * HPY + (uint32_t)v * CHUNK + (uint32_t)tid * 4u;
            pv[v] = idx + 4u <= A.py_cap ? *reinterpret_cast<const uint4 *>(A.py_base + idx) : make_uint4(0, 0, 0, 0);
        }
    };
    auto fetch_np = [&](uint32_t half) {
#pragma unroll
        for (int v = 0; v < NPL; ++v) {
            const uint32_t idx = half * HNP + (uint32_t)v * CHUNK + (uint32_t)tid * 4u;
            nv[v] = idx + 4u <= A.np_cap ? *reinterpret_cast<const uint4 *>(A.np_base + idx) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_py = [&](uint32_t half) {
#pragma unroll
        for (int v = 0; v < PYL; ++v)
            *reinterpret_cast<uint4 *>(ring_py + (half & 1u) * HPY + (uint32_t)v * CHUNK + (uint32_t)tid * 4u) = pv[v];
        if (!(half & 1u) && tid < 4) *reinterpret_cast<uint4 *>(ring_py + WPY + (uint32_t)tid * 4u) = pv[0];
    };
    auto store_np = [&](uint32_t half) {
#pragma unroll
        for (int v = 0; v < NPL; ++v)
            *reinterpret_cast<uint4 *>(ring_np + (half & 1u) * HNP + (uint32_t)v * CHUNK + (uint32_t)tid * 4u) = nv[v];
        if (!(half & 1u) && tid < 4) *reinterpret_cast<uint4 *>(ring_np + WNP + (uint32_t)tid * 4u) = nv[0];
    };
    uint32_t opy = A.py_off, onp = A.np_off;
    uint32_t kpy = opy / HPY, knp = onp / HNP;
    fetch_py(kpy); store_py(kpy);
    fetch_py(kpy + 1); store_py(kpy + 1);
    fetch_np(knp); store_np(knp);
    fetch_np(knp + 1); store_np(knp + 1);
    fetch_py(kpy + 2);
    fetch_np(knp + 2);
    __syncthreads();
    auto pyr = [&](uint32_t x) { return ring_py[x & (WPY - 1u)]; };
    auto npr = [&](uint32_t x) { return ring_np[x & (WNP - 1u)]; };
    const uint32_t py_need = mt_py_need(RL), np_need = mt_np_need(RL);
    const int64_t L = g.L;
    uint32_t sl = 0;  // scratch slot (alternates per barrier)
    int64_t i = 0;
    int starved = 0, need_generic = 0;
    MtGauss gs = *A.gauss;  // wave-uniform copy; written back at the end

    uint32_t slots = 0;  // bin -> slot, 2 bits each, [o][bin]
    for (int k = 0; k < 8; ++k) slots |= ((uint32_t)M.bin_slot[k] & 3u) << (2 * k);
    const bool spare_wave = RL <= 192;  // a group's last wavefront has no position: it checks the indel draws
    const uint32_t C_MATE = 10u * (uint32_t)(RL - 1) + 2u * (uint32_t)RL;  // py words of a plain mate

    // One mate on the fast path, worked on by the four wavefronts of group o: every wavefront leaves
    // (substitution events | bin slot << 16 | not-plain << 31) of its lanes in the scratch slot.
    auto mate_body = [&](int o, uint32_t opy_m, uint32_t onp_m) {
        const uint64_t mb = mk53(npr(onp_m), npr(onp_m + 1));
        const uint64_t *bt = bin_thr + 4 * o;  // np.random.choice: #(cdf <= u), kde.py:74
        int bin = (bt[0] <= mb ? 1 : 0) + (bt[1] <= mb ? 1 : 0) + (bt[2] <= mb ? 1 : 0) + (bt[3] <= mb ? 1 : 0);
        bin = bin > 3 ? 3 : bin;
        const int slot = (int)((slots >> (2 * (o * 4 + bin))) & 3u);
        const uint32_t opy_err = opy_m + 10u * (uint32_t)(RL - 1);
        uint32_t nev = 0;
        bool cand = false;
        auto indel_step = [&](int p) {
            const uint32_t *lm = lim + ((size_t)o * RL + p) * 5;
            const uint32_t *w = ring_py + ((opy_m + 10u * (uint32_t)p) & (WPY - 1u));
#pragma unroll
            for (int x = 0; x < 5; ++x) cand |= (w[2 * x] >> 5) < lm[x];  // (lm = ceil(thr / 2^26): 0 for a test that never fires)
        };
        if (spare_wave && gw == 3)
            for (int p = lane; p < RL - 1; p += 64) indel_step(p);
        for (int p = tid_g; p < RL; p += 256) {
            const uint32_t *wq = ring_np + ((onp_m + 2u + 2u * (uint32_t)p) & (WNP - 1u));
            const uint32_t *we = ring_py + ((opy_err + 2u * (uint32_t)p) & (WPY - 1u));
            const uint64_t mq = mk53(wq[0], wq[1]);
            const uint64_t me = mk53(we[0], we[1]);
            const uint32_t h = (uint32_t)(mq >> 37);
            const size_t roff = ((size_t)(o * M.NB + slot) * RL + p) * row_h;
            const uint16_t *row = ROWS_LDS ? rows_l + roff : M.mt_rows + roff;
            int q;
            bool tie = false;
            if (nq == 41) {  // every shipped model: keys at fixed offsets, no clamping
                const uint32_t d0 = row[6], d1 = row[13], d2 = row[20], d3 = row[27], d4 = row[34];
                const int c1 = (d0 < h) + (d1 < h) + (d2 < h) + (d3 < h) + (d4 < h);
                tie = d0 == h || d1 == h || d2 == h || d3 == h || d4 == h;
                const uint16_t *seg = row + 7 * c1;
                q = 7 * c1;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const uint32_t dgt = seg[j];
                    q += dgt < h ? 1 : 0;
                    tie |= dgt == h;
                }
            } else {
                int c1 = 0;
#pragma unroll
                for (int j = 0; j < 9; ++j) {  // keys 6, 13, ... (n_q <= 60); indices past n_q read the 0xffff padding
                    const int k = min(6 + 7 * j, nq);
                    const uint32_t dgt = row[k];
                    c1 += dgt < h ? 1 : 0;
                    tie |= dgt == h;
                }
                q = 7 * c1;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int k = min(7 * c1 + j, nq);
                    const uint32_t dgt = row[k];
                    q += dgt < h ? 1 : 0;
                    tie |= dgt == h;
                }
            }
            if (tie) {  // thresholds sharing the leading digit: compare in full, from the first of them on
                const uint64_t *full = M.q_thr + ((size_t)(o * 4 + bin) * RL + p) * nq;
                while (q < nq && full[q] < mq) ++q;
            }
            const bool err = me > mut_thr[q];
            nev += (uint32_t)__popcll(__ballot(err));
            if (!spare_wave && p < RL - 1) indel_step(p);
        }
        const bool hit = __ballot(cand) != 0ull;
        if (lane == 0) scratch[sl * 8 + wave] = nev | ((uint32_t)slot << 16) | (hit ? 0x80000000u : 0u);
    };
    // after the barrier: what group o's four wavefronts left (events summed, slot, any not-plain)
    auto mate_result = [&](int o, uint32_t &nev, uint32_t &slot, bool &hit) {
        nev = 0;
        uint32_t any = 0, v = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v = scratch[sl * 8 + 4 * o + w];
            nev += v & 0xffffu;
            any |= v >> 31;
        }
        slot = (v >> 16) & 3u;
        hit = any != 0u;
    };
    while (i < A.n_pairs) {
        if (opy + py_need > A.py_fill || onp + np_need > A.np_fill) { starved = 1; break; }
        const uint32_t opy0 = opy, onp0 = onp;
        if (opy >= (kpy + 1) * HPY) { store_py(kpy + 2); ++kpy; lds_barrier(); fetch_py(kpy + 2); }
        if (onp >= (knp + 1) * HNP) { store_np(knp + 2); ++knp; lds_barrier(); fetch_np(knp + 2); }
        const MtGauss gs0 = gs;
        bool odd = false;  // this pair goes to the sequential walker
        int64_t isz;
        if (A.has_frag) {
            // int(np.random.normal(mu, sd)), numpy's legacy polar Box-Muller with its cached second value (generator.py:122);
            // a draw too close to an integer for the device's log(), or more than 16 rejected candidates: the walker
            double gval;
            if (gs.has_gauss) {
                gval = gs.gauss;
                gs.has_gauss = 0;
            } else {
                double x1 = 0, x2 = 0, r2 = 2.0;
                for (int t = 0; t < 16 && (r2 >= 1.0 || r2 == 0.0); ++t) {
                    x1 = __dadd_rn(__dmul_rn(2.0, (double)mk53(npr(onp), npr(onp + 1)) * (1.0 / 9007199254740992.0)), -1.0);
                    x2 = __dadd_rn(__dmul_rn(2.0, (double)mk53(npr(onp + 2), npr(onp + 3)) * (1.0 / 9007199254740992.0)), -1.0);
                    onp += 4;
                    r2 = __dadd_rn(__dmul_rn(x1, x1), __dmul_rn(x2, x2));
                }
                if (r2 >= 1.0 || r2 == 0.0) { odd = true; r2 = 0.5; }
                const double f = sqrt(__ddiv_rn(__dmul_rn(-2.0, log(r2)), r2));
                gs.gauss = __dmul_rn(f, x1);
                gs.has_gauss = 1;
                gs.x1 = x1;
                gs.x2 = x2;
                gval = __dmul_rn(f, x2);
            }
            const double x = __dadd_rn(A.frag_mu, __dmul_rn(A.frag_sd, gval));
            if (!(fabs(x) < 1e15) || fabs(x - rint(x)) < A.guard) odd = true;
            isz = (odd ? 0 : (int64_t)x) - 2 * (int64_t)RL;
        } else {
            // ---- insert size: np.searchsorted(cdf, np.random.rand())  (kde.py:97), two-level count in LDS
            const uint64_t m = mk53(npr(onp), npr(onp + 1));
            onp += 2;
            const int n = M.n_isize, S = (n + 63) / 64;
            const int js = min((lane + 1) * S - 1, n - 1);
            const int b = __popcll(__ballot(isz_thr[js] < m));
            int cnt = n;
            if (b < 64) {
                int c2 = 0;
                for (int r = 0; r < S; r += 64) {
                    const int idx = b * S + r + lane;
                    c2 += __popcll(__ballot(r + lane < S && idx < n && isz_thr[idx] < m));
                }
                cnt = b * S + c2;
            }
            isz = cnt;
        }
        const int64_t frag = isz + 2 * (int64_t)RL;
        int64_t fs = 0;
        auto randbelow_at = [&](uint32_t &off, uint64_t n) -> uint64_t {  // one round of 64 candidates (one word each; two beyond 2^32) from py[off ..]
            const int k = 64 - __clzll((long long)n);
            uint64_t r;
            if (k <= 32) r = pyr(off + (uint32_t)lane) >> (32 - k);
            else r = (uint64_t)pyr(off + 2u * (uint32_t)lane) | ((uint64_t)(pyr(off + 2u * (uint32_t)lane + 1u) >> (64 - k)) << 32);
            const unsigned long long ok = __ballot(r < n);
            if (!ok) { odd = true; return 0u; }
            const int t = __ffsll(ok) - 1;
            off += (k <= 32 ? 1u : 2u) * ((uint32_t)t + 1u);
            return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(r >> 32), t) << 32) | (uint32_t)__shfl((int)(uint32_t)r, t);
        };
        if (A.sequence_type == 0 && !odd) {  // generator.py:134-135, 142-144
            const int64_t width = L - frag;
            fs = (int64_t)randbelow_at(opy, (uint64_t)(width > 0 ? width : L - RL));
        }
        const int64_t fe = fs + RL;
        auto exceptions_in = [&](int64_t lo, int64_t hi) -> bool {  // any letter outside ACGT in [lo, hi)
            if (!g.has_exceptions) return false;
            const int64_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
            bool any = false;
            for (int64_t w = w0 + lane; w <= w1; w += 64) {
                uint32_t bits = g.mask[w];
                if (w == w0) bits &= 0xffffffffu << (lo & 31);
                if (w == w1) bits &= 0xffffffffu >> (31 - ((hi - 1) & 31));
                any |= bits != 0u;
            }
            return __ballot(any) != 0ull;
        };
        // the py words of the second mate do not depend on the first mate's outcome: its start, incl. the
        // reverse-end fallback randrange drawn between the mates, is known now; only its np offset is not
        uint32_t opy1 = opy + C_MATE;
        int64_t rs, re;
        if (A.sequence_type == 0) { rs = fe + isz; re = rs + RL; }  // generator.py:164-177
        else { rs = L - RL; re = L; }
        if (!odd && re > L) { re = RL + (int64_t)randbelow_at(opy1, (uint64_t)(L - RL)); rs = re - RL; }
        if (fe > L || rs < 0) odd = true;  // templates cut by the genome ends (short / negative fragments): Python slice rules
        if (!odd) odd = exceptions_in(fs, fe) || exceptions_in(rs, re);
        PairDesc d;
        d.fs = (int32_t)(uint32_t)fs;
        d.re = (int32_t)(uint32_t)re;
        d.isz = (int32_t)isz;
        d.meta = 0;
        const uint32_t hi_bits = desc_hi_bits(fs, re);  // (bits 32-35 of the coordinates: records of 2^31 bases and more)
        MtPairRec rc;
        if (!odd) {
            // both mates at once: group 0 the first, group 1 the second ASSUMING the first has no substitution event
            const uint32_t onp1_guess = onp + 2u + 2u * (uint32_t)RL;
            if (grp == 0) mate_body(0, opy, onp);
            else mate_body(1, opy1, onp1_guess);
            lds_barrier();
            uint32_t nev0, slot0, nev1, slot1;
            bool hit0, hit1;
            mate_result(0, nev0, slot0, hit0);
            mate_result(1, nev1, slot1, hit1);
            sl ^= 1u;  // the next barrier uses the other slot (a fast wave may write before a slow one has read)
            const uint32_t onp1 = onp1_guess + 2u * nev0;
            if (!hit0 && nev0 != 0u) {  // the guess was wrong: the second mate again, from its true np offset
                if (grp == 1) mate_body(1, opy1, onp1);
                lds_barrier();
                mate_result(1, nev1, slot1, hit1);
                sl ^= 1u;
            }
            odd = hit0 || hit1;
            d.meta = slot0 | (slot1 << 2) | hi_bits;
            rc.opy_err[0] = opy + 10u * (uint32_t)(RL - 1);
            rc.onp_bin[0] = onp;
            rc.opy_err[1] = opy1 + 10u * (uint32_t)(RL - 1);
            rc.onp_bin[1] = onp1;
            opy = opy1 + C_MATE;
            onp = onp1 + 2u + 2u * (uint32_t)RL + 2u * nev1;
        }
        if (odd) { opy = opy0; onp = onp0; gs = gs0; need_generic = 1; break; }
        bool keep = true;
        if (A.gc_bias) {  // generator.py:82-92
            keep = mk53(npr(onp), npr(onp + 1)) < A.gc_thr;
            onp += 2;
        }
        if (keep) {
            if (tid == 0) { desc[i] = d; A.rec[i] = rc; }
            ++i;
        }
    }
    if (tid == 0) {
        A.res->py_used = opy - A.py_off;
        A.res->np_used = onp - A.np_off;
        A.res->n_done = i;
        A.res->starved = starved;
        A.res->pad = need_generic;
        A.res->n_mut = 0;
        A.res->need_host = 0;
        A.res->host_cached = 0;
        *A.gauss = gs;
    }
}
template <int PYV, int NPV, bool ROWS_LDS>
__global__ __launch_bounds__(RES_THREADS) void k_mt_resolve(DevModel M, DevGenome g, MtResolveArgs A, PairDesc *desc) {
    mt_resolve_body<PYV, NPV, ROWS_LDS>(M, g, A, desc);
}
struct MtResolveJob {
    MtResolveArgs A;
    DevGenome g;
    PairDesc *desc;
};
template <int PYV, int NPV, bool ROWS_LDS>
__global__ __launch_bounds__(RES_THREADS) void k_mt_resolve_w(DevModel M, const MtResolveJob *jobs) {
    MtResolveJob j = jobs[blockIdx.x];
    if (j.A.n_pairs <= 0) return;  // (uniform: the worker is done, or this turn is its walker's)
    j.A.py_base = as_global(j.A.py_base, jobs); j.A.np_base = as_global(j.A.np_base, jobs); j.A.res = as_global(j.A.res, jobs);
    j.A.rec = as_global(j.A.rec, jobs); j.A.gauss = as_global(j.A.gauss, jobs);
    mt_resolve_body<PYV, NPV, ROWS_LDS>(M, genome_as_global(j.g, jobs), j.A, as_global(j.desc, jobs));
}

// reads of the pairs k_mt_resolve resolved: one wavefront per (pair, mate); the read is the template
// (no indel, only ACGT), phred scores and substitutions come from the recorded stream offsets
// --store_mutations: the rows of a (pair, mate) are its substituted positions in ascending order, and the file
// lists the mates in order.  Pass 1 (mut_cnt != NULL) only counts the rows of every mate; the host turns the counts
// into offsets; pass 2 (mut != NULL) writes each row at its final place mut_off[item] + rank.
struct MtEmitMut {
    int32_t *mut_cnt;         // pass 1: [2 * n_pairs] rows per (pair, mate)
    const int64_t *mut_off;   // pass 2: [2 * n_pairs] first row of (pair, mate) in `mut`
    MutRecord *mut;
    int64_t mut_cap;
    int64_t pair_base;        // pair index (within the call) of this launch's first pair
};

__device__ __forceinline__ void mt_emit_body(const DevModel &M, const DevGenome &g, const uint32_t *py, const uint32_t *np,
                                             int64_t n_pairs, const PairDesc *desc, const MtPairRec *rec, uint8_t *out0,
                                             uint8_t *out1, uint8_t *out2, uint8_t *out3, const MtEmitMut &E) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= 2 * n_pairs) return;
    const int64_t i = item >> 1;
    const int o = (int)(item & 1);
    const int RL = M.RL;
    const PairDesc d = desc[i];
    const MtPairRec r = rec[i];
    const uint32_t onp_q = r.onp_bin[o] + 2u, onp_s = onp_q + 2u * (uint32_t)RL, opy_e = r.opy_err[o];
    int bin = count_le(M.bin_thr + 4 * o, 4, mk53(np[r.onp_bin[o]], np[r.onp_bin[o] + 1]));
    bin = bin > 3 ? 3 : bin;
    // (the phreds through the 16-bit digit rows, as the walker and the resolver find them: a bisection of the 53-bit thresholds --
    //  six dependent 8-byte reads per base, every lane in a row of its own -- was 27 GB of L2 -> L1 lines per 117 k pairs and
    //  most of the emitter's 1.2 ms; a row of digits is one or two lines)
    const int slot = M.bin_slot[o * 4 + bin] & 3;
    uint8_t *ob = (o ? out2 : out0) + (size_t)i * M.row;
    uint8_t *oq = (o ? out3 : out1) + (size_t)i * M.row;
    uint32_t nev = 0, n_rows = 0;
    const int64_t row0 = E.mut ? E.mut_off[item] : 0;
    for (int p0 = 0; p0 < RL; p0 += 64) {
        const int p = p0 + lane;
        bool err = false;
        int ch = 0, q = 0, bi = -1, before = 0;
        if (p < RL) {
            ch = o == 0 ? fetch_ascii(g, desc_fs(d) + p) : complement_ascii(fetch_ascii(g, desc_re(d) - 1 - p));
            before = ch;
            const uint64_t mq = mk53(np[onp_q + 2u * (uint32_t)p], np[onp_q + 2u * (uint32_t)p + 1u]);
            q = mt_phred_of(M, M.mt_rows, (uint32_t)M.mt_row_w * 2u, o, slot, bin, p, mq);
            bi = base_index(ch);
            const uint64_t m = mk53(py[opy_e + 2u * (uint32_t)p], py[opy_e + 2u * (uint32_t)p + 1u]);
            err = m > M.mut_thr[q] && bi >= 0;
        }
        const unsigned long long evm = __ballot(err);
        if (err) {
            const uint32_t rank = nev + (uint32_t)__popcll(evm & ((1ull << lane) - 1ull));
            const uint64_t ms = mk53(np[onp_s + 2u * rank], np[onp_s + 2u * rank + 1u]);
            const size_t row = ((size_t)(o * RL + p) * 4 + bi) * 3;
            const int k = (ms >= M.subst_thr[row]) + (ms >= M.subst_thr[row + 1]);
            ch = M.subst_alt[row + k];
        }
        nev += (uint32_t)__popcll(evm);
        // a row only if the new letter differs from the original read, which here is the template (__init__.py:98)
        const bool rowp = err && ch != before;
        const unsigned long long rm = __ballot(rowp);
        if (rowp && E.mut) {
            const int64_t at = row0 + n_rows + (int64_t)__popcll(rm & ((1ull << lane) - 1ull));
            if (at < E.mut_cap) {
                MutRecord mr;
                mr.pair = (int32_t)(E.pair_base + i); mr.mate = (int8_t)o; mr.type = 0; mr.position = (int16_t)p;
                mr.ref = (uint8_t)before; mr.alt = (uint8_t)ch; mr.quality = (int16_t)q;
                E.mut[at] = mr;
            }
        }
        n_rows += (uint32_t)__popcll(rm);
        if (p < RL) { ob[xp(p)] = (uint8_t)ch; oq[xp(p)] = (uint8_t)q; }
    }
    if (E.mut_cnt && lane == 0) E.mut_cnt[item] = (int32_t)n_rows;
    for (int p = RL + lane; p < M.pitch; p += 64) { ob[xp(p)] = 0; oq[xp(p)] = 0; }
}
__global__ __launch_bounds__(256) void k_mt_emit(DevModel M, DevGenome g, const uint32_t *py, const uint32_t *np,
                                                 int64_t n_pairs, const PairDesc *desc, const MtPairRec *rec, uint8_t *out0,
                                                 uint8_t *out1, uint8_t *out2, uint8_t *out3, MtEmitMut E) {
    mt_emit_body(M, g, py, np, n_pairs, desc, rec, out0, out1, out2, out3, E);
}
struct MtEmitJob {  // (grid.y = worker; no --store_mutations rows on this path)
    const uint32_t *py, *np;
    int64_t n_pairs;
    const PairDesc *desc;
    const MtPairRec *rec;
    uint8_t *out[4];
    DevGenome g;
};
__global__ __launch_bounds__(256) void k_mt_emit_w(DevModel M, const MtEmitJob *jobs) {
    const MtEmitJob j = jobs[blockIdx.y];
    if ((int64_t)blockIdx.x * 4 >= 2 * j.n_pairs) return;
    const MtEmitMut none{};
    mt_emit_body(M, genome_as_global(j.g, jobs), as_global(j.py, jobs), as_global(j.np, jobs), j.n_pairs, as_global(j.desc, jobs), as_global(j.rec, jobs),
                 as_global(j.out[0], jobs), as_global(j.out[1], jobs), as_global(j.out[2], jobs), as_global(j.out[3], jobs), none);
}

}  // namespace iss
