// iss_kernels.hip.h -- gfx950 kernels of the read-generation path (included by iss_mi355x.hip).
//
// Reference semantics (InSilicoSeq v2.0.1), restated for a counter-based RNG:
//   simulate_read                      iss/generator.py:98-192
//   introduce_indels/adjust_seq_length iss/error_models/__init__.py:158-228, 114-156
//   gen_phred_scores/random_insert_size iss/error_models/kde.py:52-98
//   mut_sequence                       iss/error_models/__init__.py:69-112
//
// Kernel plan (one iss_generate / iss_generate_batch call; in a batch call the records stand side by side in one arena and the
// pair descriptors carry arena coordinates).  Launch order per chunk of a call:
//   k_setup        : 1 lane / pair -> PairDesc {forward_start, reverse_end, bin slots, attempt, insert}; models whose reads
//                    rarely have an indel ("light", DevModel::p_read_event): + the pair's two indel event processes, a read
//                    with an event goes straight to k_indel_fixup's list.  On the setup stream, beside the call before.
//   k_indel_scan   : heavy models only.  1 lane / read: the read's indel events (step, event mask), sampled by skipping from
//                    one firing test to the next; two segmented lists of the reads with an event (one event step / more).
//   k_indel_script : heavy models only, two launches side by side (reads with ONE event step: closed form; the others: the
//                    walk of the token transducer).  1 lane / listed read: introduce_indels + adjust_seq_length as an EDIT
//                    SCRIPT -- per 8-position piece a shift of the template or 8 explicit letters (SC_* below).
//   k_main         : persistent workgroups (1024 lanes, one per CU); the compressed per-position quality CDF rows of a
//                    position tile are staged ONCE per workgroup in LDS; 4 lanes / pair, a lane takes 8 consecutive
//                    positions of both mates per iteration: three Philox blocks give its 16 quality digits (16 bits) and
//                    16 error-test digits (8 bits); CDF inversion = LDS guide byte + two packed (threshold, phred, error
//                    threshold) entries; letters from the 2-bit genome through LDS letter tables; two 16-byte stores per
//                    lane, 64 contiguous bytes per pair and mate.  Digits that tie with a table entry, and positions whose
//                    error test fires, are queued in a per-wavefront LDS ring and settled exactly (no global loads) every
//                    few iterations.  <.., INDEL>: a scripted read's pieces come from the shifted genome window or the
//                    script's explicit letters -- the read is built once, mut_sequence sees the final letters.
//   k_indel_fixup  : 1 wavefront / listed read (irregular pairs, reads with more events than a script holds, IUPAC records,
//                    windows that leave the record; every read with an event of a light model): exact sequential indel
//                    semantics (lane 0 walks the token transducer) + re-mutation by all lanes, rewrites that read's letters.
// No MFMA anywhere: this is sampling/indexing.  All f64 comparisons of the reference are exact
// integer comparisons here (thresholds prepared on the host, see iss_mi355x.h / DESIGN.md); the indel tests are
// sampled as an event process with the same joint distribution (indel_events below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace iss {

// ---------------------------------------------------------------- RNG address map
enum : uint32_t {
    K_PAIR = 0, K_FS = 1, K_RS = 2, K_QM = 3, K_SUB = 4, K_QM_LO = 7, K_FRAG = 10, K_EV = 11  // (5, 6, 8, 9: retired)
};

struct u32x4 {
    uint32_t x, y, z, w;
};

// Philox4x32 with ROUNDS rounds (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).  Ten rounds
// is the generator's default; SEVEN is the fewest the authors found to pass every test of BigCrush ("Crush-resistant",
// their table 2; philox4x32_R(7, ...) of Random123) and what the HOT digit blocks (K_QM: three per 16 bases, all of k_main's
// 64-bit multiplies) use since round 4 -- measured: 1.32 -> 1.11 ms per 5 M-pair launch, the multiplies being the kernel's
// scarcest issue slots.  Every other draw keeps ten rounds.  The CPU oracle follows the same rule; both round counts are checked
// against Random123's known-answer vectors (tests/).
constexpr int PHILOX_HOT_ROUNDS = 7;
template <int ROUNDS = 10>
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);  // 3-input xor
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
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
// (`kind` is a constant at every call site: the test folds away)
__device__ __forceinline__ u32x4 draw_block(const Addr &a, uint32_t kind, uint32_t index, uint32_t sub) {
    if (kind == K_QM) return philox4x32<PHILOX_HOT_ROUNDS>(a.c0, a.c1, (kind << 24) | (index & 0xffffffu), sub, a.k0, a.k1);
    return philox4x32<10>(a.c0, a.c1, (kind << 24) | (index & 0xffffffu), sub, a.k0, a.k1);
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
// ---------------------------------------------------------------- indel events
// introduce_indels runs, per loop step n <= RL-2, four insertion tests `random() < p_ins[n][x]` and one deletion test
// `random() < p_del[n][base]` (__init__.py:193-196, :209): 5 (RL - 1) independent Bernoulli tests per read, all but a
// handful failing.  The position-addressable path samples the SAME joint distribution by skipping from one firing
// test to the next ("indel event process"; all integer arithmetic):
//   S[t] = prod over the slots of t's segment up to t of (1 - T / 2^53), 0.64 fixed point; one uniform r in (0, 1] per
//   draw (K_EV block j, sub = mate, words 0-1): the next firing test after slot `cur` is the first t of its segment
//   with S[t] <= r * S[cur]; none: the next segment, with the next draw.  A firing deletion slot fires for base b iff
//   floor(v * T_max / 2^53) < T_b, v from words 2-3 of the same block (the reference's one uniform for the four bases).
// emit(n, mask): step n, bits 0-3 insertion of letter slot x fires, bits 4-7 the deletion fires if the token is base b;
// steps come in ascending order (several calls for one step are possible).
constexpr uint64_t EV_ONE = 0xffffffffffffffffull;
// ONE draw of the event process (the CPU oracle holds its twin; exported through iss_ev_step, iss_units.hip.h).  State
// `cur`: the last slot decided (-1: none yet); m53 / v53: numerators of the draw's uniform and of the deletion sub-draw.
// Returns the new state; slot = the slot that fired (mask = its event mask) or -1 (nothing fires in the rest of cur's
// segment: the state is the segment's last slot and the next draw starts the next segment).
__device__ __forceinline__ int ev_step(const uint64_t *S, const uint16_t *E, const uint64_t *T, const uint64_t *del_thr /* [RL][4] of the mate */,
                                       int cur, uint64_t m53, uint64_t v53, int &slot, uint32_t &mask) {
    const int seg_last = E[cur + 1];
    uint64_t base = EV_ONE;
    if (cur >= 0 && E[cur] == seg_last) base = S[cur];
    const uint64_t rr = ((((uint64_t)1 << 53) - m53) << 11) - 1u;  // (1 - u) in 0.64 fixed point
    const uint64_t target = __umul64hi(rr, base);
    slot = -1;
    mask = 0;
    if (S[seg_last] > target) return seg_last;  // nothing fires in the rest of the segment
    int lo = cur + 1, hi = seg_last;  // first slot with S <= target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (S[mid] <= target) hi = mid; else lo = mid + 1;
    }
    const int n = lo / 5, k = lo - 5 * n;
    if (k < 4) {
        mask = 1u << k;
    } else {
        const uint64_t tm = T[lo];
        const uint64_t scaled = (__umul64hi(v53, tm) << 11) | ((v53 * tm) >> 53);  // floor(v * T_max / 2^53)
        for (int b = 0; b < 4; ++b) mask |= (scaled < del_thr[(size_t)n * 4 + b] ? 16u : 0u) << b;
    }
    slot = lo;
    return lo;
}
template <class Emit>
__device__ __forceinline__ void indel_events(const uint64_t *S, const uint16_t *E, const uint64_t *T, const uint64_t *del_thr /* [RL][4] of the mate */,
                                             int ns, const Addr &a, int o, Emit emit) {
    int cur = -1;
    uint32_t j = 0;
    while (cur < ns - 1) {
        const u32x4 w = draw_block(a, K_EV, j++, (uint32_t)o);
        int slot;
        uint32_t mask;
        cur = ev_step(S, E, T, del_thr, cur, mk53(w.x, w.y), mk53(w.z, w.w), slot, mask);
        if (slot >= 0) emit(slot / 5, mask);
    }
}

// Output rows.  A pair owns ONE row of M.row bytes (a multiple of 128).  The 32 read positions 32 l .. 32 l + 31 are the
// 128-byte line l of the row: bytes 0-63 the forward mate, 64-127 the reverse mate, each four 16-byte pieces
// [bases 8][phreds 8] of 8 positions -- the four lanes that work on a pair write 64 contiguous bytes per store.
// RunArgs::out[k] points at array k's first byte of the launch's first row (ROW_ARRAY_OFF): position p of array k of
// pair i is out[k][i * M.row + xp(p)].
__host__ __device__ __forceinline__ int xp(int p) { return ((p >> 5) << 7) + (((p >> 3) & 3) << 4) + (p & 7); }
__host__ __device__ __forceinline__ int row_array_off(int k) { return (k >> 1) * 64 + (k & 1) * 8; }

// ---------------------------------------------------------------- device-side tables
struct DevModel {
    int32_t RL, n_isize, n_q, G, pitch;  // pitch = 8 * ceil(RL / 8); G = pitch/4 = position groups per read
    int32_t row;         // bytes of a pair's output row: 128 * ceil(S / 4) (see xp())
    // compressed quality rows for k_main (built at upload, see iss_mi355x.hip: build_qrows)
    int32_t NB;          // bin slots per orientation (non-empty bins, compacted)
    int32_t GB;          // guide bits: a row starts with 1 << GB guide bytes (top GB bits of the digit)
    int32_t stride_w;    // u32 words per row: guide words + (S_max + 2) entries
    int32_t GS;          // words per position group = 4 * stride_w + 1 (odd: consecutive lanes hit distinct LDS banks)
    int32_t TG, TP;      // position groups / positions per tile (TG even)
    int32_t S, TS;       // superitems (8 positions) per read (pitch / 8) / per tile (TG / 2)
    int32_t n_tiles;
    int32_t tile_words;  // 2 * NB * TG * GS rounded up to a multiple of 4
    int8_t bin_slot[8];  // [o][bin] -> slot (or -1)
    int8_t slot_bin[8];  // [o][slot] -> bin
    const uint32_t *qrows;      // [n_tiles][2][NB][TG] groups of GS words (4 rows of stride_w + 1 pad)
    const uint64_t *isize_thr;  // [n_isize]
    const uint64_t *bin_thr;    // [2][4]
    const uint64_t *q_thr;      // [2][4][RL][n_q]   (exact tie resolution)
    const uint32_t *subst13;    // [n_tiles][2][TP][4]: t0_13 | t1_13 << 13 | (alt0 | alt1 << 2 | alt2 << 4) << 26: leading 13 bits of
                                // subst_thr, alternatives as indices into alt_letters
    uint32_t alt_letters;       // the (<= 4) distinct letters of subst_alt
    // edit scripts of reads with indel events (k_indel_script -> k_main<.., INDEL>; see SC_* below)
    int32_t sc_gpt;             // groups of 8 iterations per position tile of k_main: ceil(ceil(TS / 4) / 8)
    int32_t sc_stride;          // bytes per read: n_tiles * sc_gpt * 64
    int32_t ins_plain;          // every insertion letter of the model is one of A/C/G/T (else reads with events take k_indel_fixup)
    float p_read_event;         // probability that a read has an indel event (the larger of the two mates')
    float p_defer;              // expected share of bases that leave k_main's hot loop for the exact path (the host's choice of k_main_g)
    const uint64_t *subst_thr;  // [2][RL][4][3]
    const uint8_t *subst_alt;   // [2][RL][4][3]
    const uint64_t *ins_thr;    // [2][RL][4]
    const uint8_t *ins_letter;  // [2][RL][4]
    const uint64_t *del_thr;    // [2][RL][4]
    const uint64_t *mut_thr;      // [n_q+1]
    // indel events (k_indel_scan; see indel_events()): per mate the 5 * (RL - 1) test slots 5 n + k (k = 0..3 insertion of letter
    // slot k at loop step n, k = 4 the deletion with the largest of its four thresholds)
    const uint64_t *ev_S;         // [2][ev_ns] survival inside the slot's segment, 0.64 fixed point
    const uint16_t *ev_E;         // [2][ev_ns] last slot of the slot's segment
    const uint64_t *ev_T;         // [2][ev_ns] threshold of the slot
    int32_t ev_ns;
    int32_t n_scan;               // 1: some indel probability is non-zero (the indel pass runs), 0: none
    // reference-compatible MT mode, k_mt_resolve (iss_mt_compat.hip.h)
    int32_t mt_row_w;             // 32-bit words per row of mt_rows (odd)
    const uint16_t *mt_rows;      // [2][NB][RL] rows of n_q leading digits min(q_thr >> 37, 0xffff) (no merging: index == phred)
    const uint32_t *mt_lim;       // [2][RL][5] ceil(thr / 2^26) of the 4 insertion thresholds and the largest deletion threshold    // BasicErrorModel (iss/error_models/basic.py)
    int32_t quality_mode;         // 0 KDE tables; 1 basic: phred = round(-10 log10(1 - min(N(basic_mean, basic_sd), basic_cap)))
    int32_t basic_insert_size;    // basic.py:21, :56-63 (no draw)
    double basic_mean, basic_sd, basic_cap;
};

struct DevGenome {
    const uint32_t *packed;  // 2-bit codes, 16 bases / word (A,T,C,G = 0..3; exceptions 0); words -1 and
                             // ceil(L/16)..+1 are readable padding
    const uint32_t *mask;    // 1 bit / base: 1 = read the ASCII copy (IUPAC or lower case); same padding
    const uint8_t *ascii;
    int64_t L;
    int32_t has_exceptions;  // any letter that is not plain A/C/G/T
};

struct PairDesc {
    int32_t fs;     // forward_start, low 32 bits
    int32_t re;     // reverse_end (reverse_start = re - RL), low 32 bits
    uint32_t meta;  // bits 0-1 bin slot fwd, 2-3 bin slot rev, 4 / 5 fwd / rev window has IUPAC or lower-case letters,
                    // 6 irregular geometry (custom fragment lengths only), 8-11 / 12-15 bits 32-35 of fs / re (signed: round 4,
                    // records of 2^31 - 1 bases and more -- the reference spills them to a memmap, generator.py:313-331),
                    // 16-31 attempt
    int32_t isz;    // insert size
};
// coordinates of a pair: 36-bit signed (a record holds up to MAX_RECORD bases; a custom fragment length can make re negative)
constexpr int64_t MAX_RECORD = ((int64_t)1 << 34) - 4096;  // word offsets into the packed genome (16 bases / word) stay below 2^32 bytes
// (k_main addresses a window word as a 32-bit BYTE offset, `word << 2`: the last word a lane can ask for lies behind the record's
//  last base by at most a read's padded length (8 bases x 128 superitems) + the windows' overhang -- with room to spare)
static_assert((((MAX_RECORD + 8 * 128 + 64) / 16 + 2) * 4) < ((int64_t)1 << 32) - 512, "k_main: 32-bit byte offsets into the packed genome");
__host__ __device__ __forceinline__ int64_t desc_fs(const PairDesc &d) {
    return ((int64_t)((int32_t)(d.meta << 20) >> 28) << 32) | (uint32_t)d.fs;
}
__host__ __device__ __forceinline__ int64_t desc_re(const PairDesc &d) {
    return ((int64_t)((int32_t)(d.meta << 16) >> 28) << 32) | (uint32_t)d.re;
}
__host__ __device__ __forceinline__ uint32_t desc_hi_bits(int64_t fs, int64_t re) {
    return (((uint32_t)(fs >> 32) & 15u) << 8) | (((uint32_t)(re >> 32) & 15u) << 12);
}

// one VCF row of --store_mutations (iss/error_models/__init__.py:98-108, 197-221; generator.py:598-620)
struct MutRecord {
    int32_t pair;      // pair index within the call (the read id's i); -1: unused slot
    int8_t mate;       // 0 forward, 1 reverse
    int8_t type;       // bits 0-1: 0 substitution, 1 insertion, 2 deletion.  Philox path only, stripped by the host:
                       // bits 2-4 order inside the loop step (insertion slot 0-3, deletion 4), bit 5 written by the fix-up
    int16_t position;  // 0-based
    uint8_t ref;       // ASCII
    uint8_t alt;       // ASCII: new base / inserted letter / '.'
    int16_t quality;   // phred for substitutions, -1 ('.') otherwise
};

// Philox path: the kernels append rows unordered; every wavefront reserves MUT_CHUNK slots at a time with one
// atomic and fills them (unused slots keep pair == -1); the host drops stale rows and sorts (iss_mutations_download).
constexpr uint32_t MUT_CHUNK = 256;
struct MutChunk {
    uint32_t base, used;  // wave-uniform
};

struct FragAmb {
    uint32_t pair, pad;
    double x1, x2;  // the accepted polar candidate
};

constexpr int MAX_TILES = 64;  // position tiles of k_main (a model whose tables need more is rejected at upload)

struct RunArgs {
    int64_t n_pairs;
    uint64_t first_ordinal;
    uint64_t seed;
    int32_t sequence_type;
    int32_t gc_bias;
    uint64_t gc_thr;  // ceil(0.90 * 2^53): accept iff m < gc_thr (generator.py:88)
    uint8_t *out[4];  // rows of this launch: R1 base, R1 qual, R2 base, R2 qual (out[k] = out[0] + row_array_off(k), see xp())
    // custom fragment length (generator.py:121-123): fragment = int(mu + sd * gaussian), per-pair polar Box-Muller
    int32_t has_frag;
    double frag_mu, frag_sd, frag_guard;
    struct FragAmb *amb_list;     // pairs whose value is too close to an integer for the device's log(): host decides
    uint32_t *amb_count;
    const uint32_t *ov_pairs;     // k_setup override pass: these pairs ...
    const int64_t *ov_frags;      // ... with these host-evaluated fragment lengths
    uint32_t n_ov;
    uint32_t *flags, *fix_list, *fix_count;  // irregular pairs (template shorter than the read, ...) go straight to the fix-up
    // indel events (k_indel_scan -> k_indel_script): reads (2 * pair + mate) with an event are listed once in read_list as
    // {read, first event, second event, number of steps with an event}, an event = step << 8 | event mask; a read with
    // more than two such steps keeps all of them in its EV_K words of ev_list.  ev_count[read] = min(steps with an
    // event, 15) | first such step << 4 (0: none) for every read of the launch.
    uint32_t *ev_count, *ev_list, *read_count;  // read_count[w] / [SCAN_MAX_WGS + w]: reads scan workgroup w listed in read_list / read_list1
    uint4 *read_list;    // reads with two or more steps with an event
    uint2 *read_list1;   // reads with ONE such step: {read, event} -- more than half of the listed reads of BASELINE configs[4]; their
                         // edit scripts are straight-line code (k_indel_script<.., SINGLE>), and no lane of a block waits for a longer walk
    uint32_t scan_wgs;  // workgroups of k_indel_scan: the read list is one segment per workgroup (no global counter)
    // Models whose reads often have indels: the edit scripts k_indel_script leaves for k_main (see SC_* below): M.sc_stride bytes
    // per READ (2 * pair + mate), valid iff ev_count[read] != 0 once k_indel_script has run
    uint8_t *script;
    PairDesc *desc_out;  // k_main copies the descriptors it works from here (the call's own set is double-buffered: k_setup of the
                         // next call may be rewriting it while the host asks for this call's coordinates)
    int32_t light;  // 1: reads with an indel are rare (DevModel::p_read_event): k_indel_scan hands every one of them to k_indel_fixup
                    // (no k_indel_scan / k_indel_script launches, k_main's plain variant)
    uint16_t tile_wg0[MAX_TILES + 2];  // k_main: workgroups [tile_wg0[t], tile_wg0[t + 1]) work on position tile t
    MutRecord *mut;               // --store_mutations rows (NULL: off)
    uint32_t *mut_count;          // slots reserved so far
    uint32_t mut_cap;
    int64_t pair_base;            // index of this launch's first pair within the iss_generate call
    // iss_generate_batch: several records in one launch.  The genomes stand in ONE arena (DevGenome of the launch) at
    // offsets `off` (bases, multiples of 32); pair i of the call belongs to the item k with item_first[k] <= i <
    // item_first[k + 1]; pair descriptors carry ARENA coordinates, so k_main does not know about records at all.
    const struct BatchItem *items;  // NULL: one record, the launch's DevGenome is that record
    const int64_t *item_first;      // [n_items + 1]
    int32_t n_items;
};

struct BatchItem {
    int64_t off;             // arena coordinate of the record's first base
    int64_t L;
    int32_t has_exceptions;
    int32_t pad;
};

// the item of pair `p` of a batch call (p counted from the call's first pair)
__device__ __forceinline__ int batch_item_of(const RunArgs &A, int64_t p) {
    int lo = 0, hi = A.n_items;  // largest k with item_first[k] <= p
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.item_first[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ DevGenome batch_genome(const DevGenome &arena, const BatchItem &it) {
    return DevGenome{arena.packed + it.off / 16, arena.mask + it.off / 32, arena.ascii + it.off, it.L, it.has_exceptions};
}


// reserve n <= 64 record slots for the active lanes of this wavefront; returns the first slot (wave-uniform) or
// 0xffffffff when the buffer is full (the host then reports the overflow)
__device__ __forceinline__ uint32_t mut_alloc(const RunArgs &A, MutChunk &c, uint32_t n) {
    if (c.used + n > MUT_CHUNK) {
        const int leader = __ffsll((unsigned long long)__ballot(1)) - 1;
        uint32_t b = 0;
        if ((int)(threadIdx.x & 63) == leader) b = atomicAdd(A.mut_count, MUT_CHUNK);
        c.base = (uint32_t)__shfl((int)b, leader);
        c.used = 0;
    }
    const uint32_t r = c.base + c.used;
    c.used += n;
    return r + n <= A.mut_cap ? r : 0xffffffffu;
}
// all active lanes call this; lanes with `have` write their row
__device__ __forceinline__ void mut_emit(const RunArgs &A, MutChunk &c, bool have, const MutRecord &r) {
    const unsigned long long m = __ballot(have);
    if (!m) return;
    const uint32_t at = mut_alloc(A, c, (uint32_t)__popcll(m));
    if (have && at != 0xffffffffu) A.mut[at + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = r;
}

// one row from one lane (the lanes of a k_indel_script wavefront walk different reads: no wave-uniform chunk there)
__device__ __forceinline__ void mut_emit1(const RunArgs &A, const MutRecord &r) {
    const uint32_t at = atomicAdd(A.mut_count, 1u);
    if (at < A.mut_cap) A.mut[at] = r;
}

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint8_t code_to_ascii(uint32_t code) { return (uint8_t)((0x47435441u >> (8 * code)) & 0xffu); }
// A,T,C,G (either case) -> 0..3, anything else (IUPAC ambiguity codes) -> -1.  Branch-free: bits 1-2 of the upper-case
// letter tell A, C, T, G apart (0, 1, 2, 3); the candidate is then checked against the letter itself.
__device__ __forceinline__ int base_index(int c) {
    const int u = c & ~0x20;
    const int i = (u >> 1) & 3;
    const int bi = ((i << 1) & 2) | (i >> 1);  // A 0, T 1, C 2, G 3 (code_to_ascii's order)
    return (int)code_to_ascii((uint32_t)bi) == u ? bi : -1;
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
        const int64_t pos = desc_fs(d) + k;
        return pos < g.L ? fetch_ascii(g, pos) : 'A';
    }
    const int64_t pos = desc_re(d) - 1 - k;
    return pos >= 0 ? complement_ascii(fetch_ascii(g, pos)) : 'A';
}

// General geometry of one mate (Python slice semantics + adjust_seq_length padding), needed once custom
// fragment lengths allow negative inserts / templates cut by the genome end (generator.py:146-180,
// __init__.py:141-155).  For the model's own insert sizes t_len == RL and this equals read_dir_base.
struct MateGeom {
    int64_t lo, hi;   // normalised slice [lo, hi) of the genome
    int64_t fe, rs;   // un-normalised forward end / reverse start (the padding rule uses them)
    int t_len;        // template length (<= RL)
};
__device__ __forceinline__ MateGeom mate_geom(int o, const PairDesc &d, int RL, int64_t L, int64_t coord_off = 0) {
    MateGeom m;
    const int64_t fs = desc_fs(d) - coord_off, re = desc_re(d) - coord_off;
    m.fe = fs + RL;
    m.rs = re - RL;
    if (o == 0) {
        m.lo = fs < L ? fs : L;
        m.hi = m.fe < L ? m.fe : L;
    } else {
        m.lo = m.rs; m.hi = re;
        if (m.lo < 0) { m.lo += L; if (m.lo < 0) m.lo = 0; } else if (m.lo > L) m.lo = L;
        if (m.hi < 0) { m.hi += L; if (m.hi < 0) m.hi = 0; } else if (m.hi > L) m.hi = L;
    }
    if (m.hi < m.lo) m.hi = m.lo;
    m.t_len = (int)(m.hi - m.lo);
    return m;
}
__device__ __forceinline__ int geom_base(const DevGenome &g, int o, const MateGeom &m, int k) {
    if (k < m.t_len) return o == 0 ? fetch_ascii(g, m.lo + k) : complement_ascii(fetch_ascii(g, m.hi - 1 - k));
    const int64_t i = k - m.t_len;
    if (o == 0) { const int64_t idx = m.fe + i; return idx >= g.L ? 'A' : fetch_ascii(g, idx); }
    const int64_t idx = m.rs - 1 - i;
    return idx < 0 ? 'A' : complement_ascii(fetch_ascii(g, idx < g.L ? idx : g.L - 1));
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

// CPython _randbelow_with_getrandbits(n), n >= 1, words from the K_FS / K_RS streams.  getrandbits(k) (_randommodule.c): k <= 32:
// one word >> (32 - k); else 32-bit words from the least significant on, the top one shifted (records of 2^32 bases and more)
__device__ __forceinline__ uint64_t randbelow(const Addr &a, uint32_t kind, uint64_t n) {
    const int k = 64 - __clzll((long long)n);
    u32x4 blk = {0, 0, 0, 0};
    for (uint32_t t = 0;;) {
        if ((t & 3u) == 0) blk = draw_block(a, kind, t >> 2, 0);
        uint64_t r;
        if (k <= 32) {
            r = word_of(blk, (int)(t & 3u)) >> (32 - k);
            ++t;
        } else {
            r = word_of(blk, (int)(t & 3u));
            ++t;
            if ((t & 3u) == 0) blk = draw_block(a, kind, t >> 2, 0);
            r |= (uint64_t)(word_of(blk, (int)(t & 3u)) >> (64 - k)) << 32;
            ++t;
        }
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
// the K_SUB block of base (p, mate o): its substitution choice (x, y) and the trailing bits of its error-test draw (z, w)
// np.random.choice(alternatives, p=...) for an erroneous, non-ambiguous base (__init__.py:95-97)
__device__ __forceinline__ int substitute(const DevModel &M, const u32x4 &sb, int o, int p, int base) {
    const int bi = base_index(base);
    if (bi < 0) return base;  // nucl.upper() in "RYWSMKHBVDN": left alone
    const uint64_t m = mk53(sb.x, sb.y);
    const size_t row = ((size_t)(o * M.RL + p) * 4 + bi) * 3;
    const int k = (m >= M.subst_thr[row]) + (m >= M.subst_thr[row + 1]);
    return M.subst_alt[row + k];
}

// ================================================================== k_pack_genome
// Genome ingest on the device: ASCII (already in HBM) -> 2-bit codes + exception mask; letters outside
// util.rev_comp's alphabet (iss/util.py:57-88) are reported through `status`.
// One lane per 32 bases: two packed words and one mask word.  status[0] = number of invalid letters,
// status[1] = offset of the first one (atomicMin), status[2] = number of exception letters.
__global__ __launch_bounds__(256) void k_pack_genome(const uint8_t *__restrict__ ascii, int64_t L, uint32_t *packed,
                                                     uint32_t *mask, unsigned long long *status) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // mask word index
    const int64_t base = w * 32;
    if (base >= L) return;
    uint32_t pk[2] = {0, 0}, mk = 0, bad = 0;
    int64_t first_bad = L;
    for (int i = 0; i < 32 && base + i < L; ++i) {
        const uint32_t c = ascii[base + i];
        uint32_t code = 0;
        switch (c) {
            case 'A': code = 0; break; case 'T': code = 1; break; case 'C': code = 2; break; case 'G': code = 3; break;
            default: {
                mk |= 1u << i;
                const uint32_t u = c & ~0x20u;
                const bool letter = (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
                const bool ok = letter && (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'Y' || u == 'R' || u == 'W' ||
                                           u == 'S' || u == 'K' || u == 'M' || u == 'N' || u == 'B' || u == 'V' ||
                                           u == 'D' || u == 'H');
                if (!ok) { if (!bad) first_bad = base + i; ++bad; }
            }
        }
        pk[i >> 4] |= code << ((i & 15) * 2);
    }
    packed[2 * w] = pk[0];
    packed[2 * w + 1] = pk[1];
    mask[w] = mk;
    if (mk) atomicAdd(&status[2], (unsigned long long)__popc(mk));
    if (bad) {
        atomicAdd(&status[0], (unsigned long long)bad);
        atomicMin(&status[1], (unsigned long long)first_bad);
    }
}

// ================================================================== k_unpack_genome
// iss_genome_upload_packed: 2-bit codes (16 bases / word, A,T,C,G = 0..3) -> the ASCII copy; one lane per word.  The
// codes of positions >= L in the last word are cleared.
__global__ __launch_bounds__(256) void k_unpack_genome(uint32_t *__restrict__ packed, int64_t L, uint8_t *__restrict__ ascii) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t base = w * 16;
    if (base >= L) return;
    uint32_t v = packed[w];
    const int n = (int)min((int64_t)16, L - base);
    if (n < 16) { v &= (1u << (2 * n)) - 1u; packed[w] = v; }
    for (int i = 0; i < n; ++i) ascii[base + i] = code_to_ascii((v >> (2 * i)) & 3u);
}

// ================================================================== k_rows_to_arrays
// iss_output_download: interleaved rows (xp) -> four plain arrays [n_pairs][8 * S]; one wavefront per pair, one lane per
// 8-byte piece (block = 64 x 4)
__global__ __launch_bounds__(256) void k_rows_to_arrays(const uint8_t *__restrict__ rows, uint8_t *__restrict__ arrays,
                                                        int64_t n_pairs, int32_t S, int32_t row) {
    const int64_t pair = (int64_t)blockIdx.x * 4 + threadIdx.y;
    if (pair >= n_pairs) return;
    for (uint32_t r = threadIdx.x; r < (uint32_t)S * 4u; r += 64u) {  // piece r: superitem r >> 2, array r & 3
        const uint32_t s = r >> 2, k = r & 3u;
        const uint2 v = *reinterpret_cast<const uint2 *>(rows + (size_t)pair * (size_t)row + (size_t)(row_array_off((int)k) + xp((int)s * 8)));
        *reinterpret_cast<uint2 *>(arrays + (((size_t)k * (size_t)n_pairs + (size_t)pair) * (size_t)S + s) * 8u) = v;
    }
}

// ================================================================== k_setup
// `ov_frag` != NULL: second pass for the few pairs whose fragment length the host evaluated.
// l_S / l_E (light models only, first pass): the indel event tables in LDS -- the pair's two reads are checked for an event
// right here (a draw each, usually) and handed to k_indel_fixup if they have one; there is no k_indel_scan launch then.
__device__ __forceinline__ void setup_pair(const DevModel &M, const DevGenome &g, const RunArgs &A, PairDesc *desc,
                                           const uint64_t *s_isize, int64_t i, const int64_t *ov_frag, int64_t coord_off = 0,
                                           const uint64_t *l_S = nullptr, const uint16_t *l_E = nullptr) {
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
    int64_t isz, frag;
    if (A.has_frag) {  // generator.py:121-123
        if (ov_frag) {
            frag = *ov_frag;
        } else {
            double x1, x2, r2;
            uint32_t t = 0;
            do {
                const u32x4 w = draw_block(a, K_FRAG, t++, 0);
                x1 = __dadd_rn(__dmul_rn(2.0, (double)mk53(w.x, w.y) * (1.0 / 9007199254740992.0)), -1.0);
                x2 = __dadd_rn(__dmul_rn(2.0, (double)mk53(w.z, w.w) * (1.0 / 9007199254740992.0)), -1.0);
                r2 = __dadd_rn(__dmul_rn(x1, x1), __dmul_rn(x2, x2));
            } while (r2 >= 1.0 || r2 == 0.0);
            const double f = sqrt(__ddiv_rn(__dmul_rn(-2.0, log(r2)), r2));
            const double x = __dadd_rn(A.frag_mu, __dmul_rn(A.frag_sd, __dmul_rn(f, x2)));
            if (!(fabs(x) < 1e15) || fabs(x - rint(x)) < A.frag_guard) {
                // the device's log() is not libm's: the host evaluates this one and k_setup_override redoes the pair
                FragAmb rec;
                rec.pair = (uint32_t)i; rec.pad = 0; rec.x1 = x1; rec.x2 = x2;
                A.amb_list[atomicAdd(A.amb_count, 1u)] = rec;
            }
            frag = fabs(x) < 1e15 ? (int64_t)x : 0;  // int(): truncation toward zero
        }
        isz = frag - 2 * (int64_t)RL;
    } else if (M.quality_mode == 1) {
        isz = M.basic_insert_size;  // BasicErrorModel.random_insert_size: a constant, no draw (basic.py:56-63)
        frag = isz + 2 * (int64_t)RL;
    } else {
        isz = count_lt(s_isize, M.n_isize, mk53(w0.x, w1.x));  // kde.py:97
        frag = isz + 2 * (int64_t)RL;
    }
    int bin_f = count_le(M.bin_thr, 4, mk53(w0.y, w1.y));      // kde.py:74
    int bin_r = count_le(M.bin_thr + 4, 4, mk53(w0.z, w1.z));
    bin_f = bin_f > 3 ? 3 : bin_f;  // kde.py:77-78
    bin_r = bin_r > 3 ? 3 : bin_r;
    int64_t fs, rs, re;
    if (A.sequence_type == 0) {
        const int64_t width = L - frag;                     // generator.py:135
        if (width > 0) fs = (int64_t)randbelow(a, K_FS, (uint64_t)width);
        else fs = (int64_t)randbelow(a, K_FS, (uint64_t)(L - RL));   // generator.py:144
        rs = fs + RL + isz;                                 // generator.py:165
        re = rs + RL;
    } else {
        fs = 0;                                             // generator.py:137
        rs = L - RL;                                        // generator.py:168
        re = L;
    }
    if (re > L) {                                           // generator.py:172-176
        re = RL + (int64_t)randbelow(a, K_RS, (uint64_t)(L - RL));
        rs = re - RL;
    }
    // a template cut by the genome end / a reverse start before the genome start: only with custom fragment
    // lengths; k_main skips such pairs' genome loads and the fix-up kernel builds both mates exactly
    const bool irregular = fs + RL > L || rs < 0;
    // does either template window (incl. the <= 3 padding bases k_main's last group touches) hold a
    // letter that is not plain A/C/G/T?  k_main reads the exception mask only for such pairs.
    uint32_t exc = 0;
    if (g.has_exceptions && !irregular) {
        uint32_t any = 0;
        for (int64_t w = fs >> 5; w <= (fs + RL + 2) >> 5; ++w) any |= g.mask[w];
        exc |= any ? 16u : 0u;
        any = 0;
        for (int64_t w = (rs - 3) >> 5; w <= (re - 1) >> 5; ++w) any |= g.mask[w];
        exc |= any ? 32u : 0u;
    }
    // first pass: the pair's indel / irregular flags start clean (no separate memset; the event counters are written by
    // k_indel_scan, every one of them)
    uint32_t fl = ov_frag ? A.flags[i] : 0u, fl_new = fl;
    if (irregular) fl_new |= 3u;
    if (l_S && !ov_frag) {  // light models: does a read of the pair have an indel event at all?
        const int ns = M.ev_ns;
        for (int o = 0; o < 2; ++o) {
            if ((fl_new >> o) & 1u) continue;
            int cur = -1, slot = -1;
            for (uint32_t j = 0; cur < ns - 1 && slot < 0; ++j) {
                const u32x4 w = draw_block(a, K_EV, j, (uint32_t)o);
                uint32_t mask;
                cur = ev_step(l_S + o * ns, l_E + o * ns, M.ev_T + (size_t)o * ns, M.del_thr + (size_t)o * RL * 4, cur, mk53(w.x, w.y), mk53(w.z, w.w), slot, mask);
            }
            if (slot >= 0) fl_new |= 1u << o;
        }
    }
    for (uint32_t o = 0; o < 2u; ++o)  // (one lane per pair: no atomics on the flags)
        if (((fl_new & ~fl) >> o) & 1u) A.fix_list[atomicAdd(A.fix_count, 1u)] = (uint32_t)i * 2u + o;
    if (fl_new != fl || !ov_frag) A.flags[i] = fl_new;
    PairDesc d;
    d.fs = (int32_t)(uint32_t)(fs + coord_off);  // (batch calls: arena coordinates)
    d.re = (int32_t)(uint32_t)(re + coord_off);
    d.meta = (uint32_t)(M.bin_slot[bin_f] & 3) | ((uint32_t)(M.bin_slot[4 + bin_r] & 3) << 2) | exc | (irregular ? 64u : 0u) |
             desc_hi_bits(fs + coord_off, re + coord_off) | (attempt << 16);
    d.isz = (int32_t)isz;
    desc[i] = d;
}

__host__ __device__ inline size_t setup_lds_bytes(int n_isize, int ev_ns, bool light) {
    return (size_t)n_isize * 8 + (light ? (size_t)2 * ev_ns * 8 + (((size_t)2 * ev_ns * 2 + 7) & ~(size_t)7) : 0);
}
// (persistent workgroups: the tables are staged once per workgroup, not once per 256 pairs)
__global__ __launch_bounds__(256) void k_setup(DevModel M, DevGenome g, RunArgs A, PairDesc *desc) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_isize[];  // insert-size thresholds (binary-searched per pair)
    const bool light = A.light && M.n_scan > 0;
    const int ns = M.ev_ns;
    // [2][ns] each (light models: the indel event tables; A.light == 2: too large for the LDS beside the insert sizes, read in place)
    const uint64_t *l_S = M.ev_S;
    const uint16_t *l_E = M.ev_E;
    for (int k = threadIdx.x; k < M.n_isize; k += blockDim.x) s_isize[k] = M.isize_thr[k];
    if (light && A.light == 1) {
        uint64_t *s_S = s_isize + M.n_isize;
        uint16_t *s_E = reinterpret_cast<uint16_t *>(s_S + 2 * ns);
        for (int k = threadIdx.x; k < 2 * ns; k += blockDim.x) { s_S[k] = M.ev_S[k]; s_E[k] = M.ev_E[k]; }
        l_S = s_S;
        l_E = s_E;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        if (A.items) {  // the record of this pair, then as for a single record
            const BatchItem it = A.items[batch_item_of(A, A.pair_base + i)];
            setup_pair(M, batch_genome(g, it), A, desc, s_isize, i, nullptr, it.off, light ? l_S : nullptr, l_E);
        } else {
            setup_pair(M, g, A, desc, s_isize, i, nullptr, 0, light ? l_S : nullptr, l_E);
        }
    }
}

__global__ __launch_bounds__(64) void k_setup_override(DevModel M, DevGenome g, RunArgs A, PairDesc *desc) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n_ov) return;
    const int64_t i = (int64_t)A.ov_pairs[j];
    if (A.items) {  // (as k_setup)
        const BatchItem it = A.items[batch_item_of(A, A.pair_base + i)];
        setup_pair(M, batch_genome(g, it), A, desc, nullptr, i, A.ov_frags + j, it.off);
        return;
    }
    setup_pair(M, g, A, desc, nullptr, i, A.ov_frags + j);
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

// The hot digits of superitem s (8 read positions, both mates) are three Philox blocks (K_QM, s, sub 0..2):
//   sub 0 / 2: quality digits (16 bits) of positions 0-3 / 4-7: word mate * 2 + (cc >> 1), half cc & 1
//   sub 1    : error-test digits (8 bits): word half * 2 + mate, byte cc            (c = p & 7, half = c >> 2, cc = c & 3)
// The quality draw is m = h16 << 37 | l37 (l37 from K_QM_LO), the error-test draw m = e8 << 45 | l45 (l45 from the
// K_SUB block of the base, which also holds its substitution choice): trailing bits are drawn on a tie only.
// word i of a block, i a per-lane value: two levels of selects (no branches around the block's last round)
__device__ __forceinline__ uint32_t word_sel(const u32x4 &v, int i) {
    const uint32_t lo = (i & 1) ? v.y : v.x, hi = (i & 1) ? v.w : v.z;
    return (i & 2) ? hi : lo;
}
__device__ __forceinline__ uint32_t hot_h16(const u32x4 &blk, int o, int cc) {
    return (word_sel(blk, o * 2 + (cc >> 1)) >> (16 * (cc & 1))) & 0xffffu;
}
__device__ __forceinline__ uint32_t hot_e8(const u32x4 &blk1, int half, int o, int cc) {
    return (word_sel(blk1, half * 2 + o) >> (8 * cc)) & 0xffu;
}
__device__ __forceinline__ uint64_t error_test_draw(uint32_t e8, const u32x4 &sub_blk) {
    return ((uint64_t)e8 << 45) | ((uint64_t)(sub_blk.z & 0x1fffu) << 32) | sub_blk.w;
}

// ---- edit scripts (k_indel_script -> k_main<.., INDEL>)
// Per READ with an indel event DevModel::sc_stride bytes: for every position tile of k_main and every group of 8 of the tile's
// iterations four 16-byte rows, one per lane j4 of the pair.  Bytes 0-7 of a row: one byte per iteration `it` of the group, for
// the piece (8 read positions) s = tile * TS + 4 * it + j4:
//   64 + sh (0..127): the piece is the template shifted by sh tokens (read position j <-> template token j + sh);
//   128 + 16 k      : the piece holds inserted letters or a run boundary: its 8 letters are the k-th 16-bit word (k < SC_ROW_CODES)
//                     of bytes 8-15 -- 2-bit codes in the format of k_main's genome windows (forward: bit pair c = read
//                     position c; reverse: bit pair c = read position 7 - c, complemented, i.e. genome orientation).
// A read without a script is all SC_IDLE bytes (shift 0).  The four lanes of a pair load 64 contiguous bytes per mate and pass.
constexpr uint32_t SC_IDLE = 0x40404040u;
constexpr int SC_ROW_CODES = 4;

constexpr int MAIN_THREADS = 1024;
constexpr int MAIN_PAIRS = MAIN_THREADS / 4;  // pairs of one workgroup pass: four lanes per pair
constexpr int MAIN_MUT_WORDS = 128;  // LDS words of the substitution-test thresholds (n_q <= 60)
constexpr int SLOW_RING = 128;  // entries (three words) of a wavefront's private ring of deferred lane-items (LDS): a
                                // wavefront pushes <= 64 per iteration and drains a round of 64 as soon as it has one
constexpr int MAIN_LUT_WORDS = 512;  // LDS words [0, 512): 4 two-bit codes (a byte of the packed genome) -> 4 letters, for the forward mate
                                     // and -- complemented, in reverse order -- for the reverse mate (at LDS offset 0: the lookups'
                                     // addresses are one SDWA shift of the window byte, the table base an immediate offset)
// Dynamic LDS of k_main (32-bit words), after the MAIN_LUT_WORDS of the letter tables:
//   [0, tile_words)           compressed quality rows of the position tile: per (mate, bin slot, group)
//                             GS words = 4 rows of stride_w words + 1 pad word; a row = guide bytes
//                             (1 << GB of them, each the BYTE offset 4 * j of an entry) then packed entries
//                             (t16 << 16 | phred << 8 | te8), ascending, closed by two sentinels; te8 = leading
//                             8 bits of the phred's substitution-test threshold
//   [mut, +128)               the substitution-test thresholds (u64) of phreds 0 .. n_q (exact path)
//   [subst, +2 * TP * 4)      per (mate, position of the tile, template base): leading 13 bits of the two substitution
//                             thresholds | alternatives (2 bits each, indices into DevModel::alt_letters) << 26
//   [rings]                   MAIN_THREADS / 64 private rings of SLOW_RING deferred lane-items, three words each
struct MainTile {  // per-workgroup constants of k_main
    int s0;        // first superitem (8 positions) of the tile
    uint32_t ts;   // superitems in the tile
};

// k_main's tables of position tile `tile` into LDS (the layout above), by the whole workgroup; the caller synchronises
__device__ __forceinline__ void main_stage_tables(const DevModel &M, int tile, uint32_t *lds_all) {
    uint32_t *const lds = lds_all + MAIN_LUT_WORDS;
    const uint4 *src = reinterpret_cast<const uint4 *>(M.qrows + (size_t)tile * M.tile_words);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < M.tile_words / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i <= M.n_q; i += blockDim.x) reinterpret_cast<uint64_t *>(lds + M.tile_words)[i] = M.mut_thr[i];
    const uint32_t *ssrc = M.subst13 + (size_t)tile * 2 * M.TP * 4;
    for (int i = threadIdx.x; i < 2 * M.TP * 4; i += blockDim.x) lds[M.tile_words + MAIN_MUT_WORDS + i] = ssrc[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {  // the letter tables: byte k of entry i = the letter of code k of byte i
        uint32_t f = 0, r = 0;
        for (int k = 0; k < 4; ++k) {
            f |= (uint32_t)code_to_ascii(((uint32_t)i >> (2 * k)) & 3u) << (8 * k);
            r |= (uint32_t)code_to_ascii((((uint32_t)i >> (2 * (3 - k))) & 3u) ^ 1u) << (8 * k);  // complement: code ^ 1
        }
        lds_all[i] = f;
        lds_all[256 + i] = r;
    }
}

// The rare work of ONE base, done exactly (one lane per flagged base, so the pass is dense): base s = mate*4 + cc of
// half `half` of superitem `sl` of the tile hit a rare condition in the hot loop (leading-digit tie, more than two
// thresholds in its guide bucket, or the substitution test fired / tied).  Everything about the base is recomputed
// from its uniforms; its phred / base BYTES are patched in place (two lanes may patch different bytes of one dword,
// hence byte stores).  A round is bound by latency, not by instructions, and a global load behind this kernel's write
// stream takes microseconds: so the common case touches NO global memory except for the two byte stores -- the queue
// entry carries the lane-item's two 8-base genome windows and the pair's bin slots, the thresholds sit in LDS.
// (Records with IUPAC / lower-case letters, gc_bias and digit ties do load: descriptor, ASCII genome, full thresholds.)
// Returns 0 (no substitution), 1 (substituted, by the letter the ORIGINAL read had there) or 3 (by a different letter: a
// --store_mutations row); `rec` is filled whenever a substitution was applied.
// INDEL: `scripted` != 0 <=> the mate was built from an edit script (k_indel_script): its windows hold the FINAL letters
// (shifted template / inserted letters, all of them plain A/C/G/T), which is what mut_sequence sees (generator.py:152-154:
// indels first, then the phreds, then the substitutions).
// HOLD (k_main_g): the byte patches are not stored but handed back in `held` -- the base's row still waits in its owner's
// registers, the patches follow it out (HeldPatch).
struct HeldPatch {
    uint64_t byte_off;  // of the base inside the launch's rows (both arrays of a mate: RunArgs::out[2 o] / out[2 o + 1])
    uint32_t w;         // bits 0-7 phred, 8-15 letter, 16 the phred is to be stored, 17 the letter is, 18 mate
};
template <bool PLAIN, bool INDEL, bool STORE_MUT, bool HOLD = false>
__device__ __forceinline__ int main_slow_base(const DevModel &M, const DevGenome &g, const RunArgs &A,
                                               const PairDesc *__restrict__ desc, const uint32_t *lds, const MainTile &T,
                                               uint32_t pair, uint32_t sl, int half, int s, uint32_t slots, uint32_t windows,
                                               uint32_t scripted, MutRecord &rec, HeldPatch *held = nullptr) {
    const int o = s >> 2, cc = s & 3, c = half * 4 + cc;
    const uint32_t s_abs = (uint32_t)T.s0 + sl;
    const int p = (int)s_abs * 8 + c;
    if (p >= M.RL) return 0;
    uint32_t attempt = 0;
    if (A.gc_bias) attempt = desc[pair].meta >> 16;  // (the only use of the attempt number)
    const Addr a = make_addr(A.seed, A.first_ordinal + pair, attempt);
    const uint32_t h = hot_h16(draw_block(a, K_QM, s_abs, half ? 2u : 0u), o, cc);
    const uint32_t e8 = hot_e8(draw_block(a, K_QM, s_abs, 1u), half, o, cc);
    const size_t byte_off = (size_t)pair * (size_t)(uint32_t)M.row + (size_t)xp(p);  // (a launch's rows may span more than 2^32 bytes: MiSeq, 5 M pairs)
    const uint32_t slot = (slots >> (2 * o)) & 3u;
    // quality: full search of the LDS row, exact thresholds on a tie
    const uint32_t gwords = (1u << M.GB) / 4;
    const uint32_t row = __umul24(__umul24((uint32_t)(o * M.NB) + slot, (uint32_t)M.TG) + 2u * sl + (uint32_t)half, (uint32_t)M.GS) +
                         __umul24((uint32_t)cc, (uint32_t)M.stride_w);
    uint32_t j = reinterpret_cast<const uint8_t *>(lds)[row * 4 + (h >> (16 - M.GB))] >> 2;  // guide bytes hold 4 * index
    uint32_t e = lds[row + gwords + j];
    const uint32_t e_next = lds[row + gwords + j + 1u];
    const uint32_t q_hot = (((e >> 16) < h ? e_next : e) >> 8) & 0xffu;  // the phred the hot loop stored (hot_lookup: the first of two entries)
    while ((e >> 16) < h) e = lds[row + gwords + (++j)];
    uint32_t q = (e >> 8) & 0xffu;
    if ((e >> 16) == h) q = (uint32_t)quality_exact(M, a, o, (int)slot, p, h);
    // (most bases come here for their substitution test, their phred stands: a byte store into a line that has left the L2
    //  is a read-modify-write in HBM -- round 4's ablations, interleaved on one box: without the two byte stores of this
    //  function 1.19 -> 1.10 ms for NovaSeq and 1.28 -> 1.02 ms for HiSeq; with them redirected into a 2 MB window that stays
    //  in the L2: 1.12 / 1.05 -- the patches that MISS are what costs, and rounds of 32 instead of 64 did not land them
    //  sooner for less: 1.21 / 1.30)
    if (HOLD) {
        held->byte_off = (uint64_t)byte_off;
        held->w = (uint32_t)o << 18;
        if (q != q_hot) held->w |= q | (1u << 16);
    } else if (q != q_hot) A.out[2 * o + 1][byte_off] = (uint8_t)q;
    // substitution test (__init__.py:94)
    const uint64_t thr = reinterpret_cast<const uint64_t *>(lds + M.tile_words)[q];
    const uint32_t t8 = (uint32_t)(thr >> 45);
    if (e8 < t8) return 0;
    const u32x4 sb = draw_block(a, K_SUB, (uint32_t)p, (uint32_t)o);
    if (e8 == t8 && !(error_test_draw(e8, sb) > thr)) return 0;
    // the template base: forward window bit pair c; reverse window (already complemented) bit pair 7 - c
    const uint32_t code = o ? ((windows >> (16 + 2 * (7 - c))) & 3u) : ((windows >> (2 * c)) & 3u);
    int base = code_to_ascii(code), bi = (int)code;
    if (!PLAIN && !(INDEL && scripted)) {  // the letter may be IUPAC / lower case, the pair irregular (custom fragment lengths)
        const PairDesc d = desc[pair];
        if (A.has_frag && (d.meta & 64u)) return 0;  // irregular pair: the fix-up kernel builds its bases
        base = fetch_ascii(g, o ? desc_re(d) - 1 - p : desc_fs(d) + p);
        if (o) base = complement_ascii(base);
        bi = base_index(base);
        if (bi < 0) return 0;  // nucl.upper() in "RYWSMKHBVDN": left alone
    }
    const uint64_t m = mk53(sb.x, sb.y);
    // leading 13 bits of the two thresholds + the three alternatives (two bits each; model_alt: their letters)
    const uint32_t sd = lds[M.tile_words + MAIN_MUT_WORDS + ((uint32_t)(o * M.TP) + 8u * sl + (uint32_t)c) * 4u + (uint32_t)bi];
    const uint32_t hs = (uint32_t)(m >> 40), t0 = sd & 0x1fffu, t1 = (sd >> 13) & 0x1fffu;
    int k = (hs > t0) + (hs > t1);
    if (hs == t0 || hs == t1) {  // tie of a leading digit: exact thresholds
        const size_t srow = ((size_t)(o * M.RL + p) * 4 + bi) * 3;
        k = (m >= M.subst_thr[srow]) + (m >= M.subst_thr[srow + 1]);
    }
    const uint32_t nb = (M.alt_letters >> (8 * ((sd >> (26 + 2 * k)) & 3u))) & 0xffu;
    if (HOLD) held->w |= (nb << 8) | (1u << 17);
    else A.out[2 * o][byte_off] = (uint8_t)nb;
    // without indels the original read equals the template, i.e. the base just replaced (__init__.py:98); a scripted
    // mate's original letter at this index is the UN-shifted template's
    int orig = base;
    if (INDEL && STORE_MUT && scripted) {
        const PairDesc d = desc[pair];
        orig = fetch_ascii(g, o ? desc_re(d) - 1 - p : desc_fs(d) + p);
        if (o) orig = complement_ascii(orig);
    }
    rec.pair = (int32_t)(A.pair_base + pair); rec.mate = (int8_t)o; rec.type = 0; rec.position = (int16_t)p;
    rec.ref = (uint8_t)base; rec.alt = (uint8_t)nb; rec.quality = (int16_t)q;
    return (int)nb != orig ? 3 : 1;
}

// One CDF inversion + substitution test of the hot loop, loop-free: guide byte -> two consecutive entries -> select.
// Entry = t16 << 16 | phred << 8 | te8.  `wq` holds the base's quality digit in its low (HI = 0) or high half, `we`
// its error-test digit in byte BYTE; row_g / row_e = byte offsets of the row's guide / entries.  Returns the selected
// entry; `flag`: lane mask of the bases that need the exact path (digit tie, > 2 thresholds of the guide bucket below the digit,
// substitution test fires or ties).
// (LDS is addressed by ABSOLUTE byte offsets -- row_g / row_e include the address of the kernel's dynamic LDS array -- through
//  address-space-3 pointers made from integers: with a generic pointer the compiler adds the array's address, a link-time zero,
//  to every row offset again: eight additions of 0 per lane-item)
typedef const __attribute__((address_space(3))) uint8_t lds_u8_t;
typedef const __attribute__((address_space(3))) uint32_t lds_u32_t;
template <int HI, int BYTE>
__device__ __forceinline__ uint32_t hot_lookup(uint32_t row_g, uint32_t row_e, uint32_t off, uint32_t wq,
                                               uint32_t we, uint32_t gsh, uint32_t gb, unsigned long long &flag) {
    const uint32_t h = HI ? (wq >> 16) : (wq & 0xffffu);
    const uint32_t gi = HI ? (wq >> (gsh + 16u)) : __builtin_amdgcn_ubfe(wq, gsh, gb);
    const uint32_t j = *reinterpret_cast<lds_u8_t *>(row_g + off + gi);
    lds_u32_t *ent = reinterpret_cast<lds_u32_t *>(row_e + off + j);  // guide bytes hold 4 * index
    const uint32_t e0 = ent[0], e1 = ent[1];
    const uint32_t sel = (e0 >> 16) < h ? e1 : e0;   // first entry with t16 >= h (if among the two)
    const uint32_t e8 = (we >> (8 * BYTE)) & 0xffu;
    // (one ballot per compare: each folds into its v_cmp, the masks are combined by the scalar unit)
    flag = __builtin_amdgcn_ballot_w64((sel >> 16) <= h) | __builtin_amdgcn_ballot_w64((sel & 0xffu) <= e8);
    return sel;
}

// word (byte BYTE of w) of the letter table at byte offset OFF of the LDS: the address is ONE instruction (byte select and
// "* 4" in an SDWA multiply), the table base the load's immediate offset
template <int BYTE, int OFF>
__device__ __forceinline__ uint32_t lut_at(uint32_t lds0, uint32_t w) {
    uint32_t a;
    const uint32_t four = 4u;
    if (BYTE == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(a) : "v"(w), "v"(four));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a) : "v"(w), "v"(four));
    // (the kernel's dynamic LDS array is the only LDS it has and stands at address 0 -- k_main / k_main_g check that once --, so the
    //  table's place is the load's immediate offset; with `lds0 + OFF + a` every lookup paid an addition of a link-time zero)
    (void)lds0;
    __builtin_assume(a <= 1020u);
    return *reinterpret_cast<lds_u32_t *>((uint32_t)OFF + a);
}

// r = r << 1 | flag in one instruction: the flag's lane mask is the carry-in of v_addc
__device__ __forceinline__ uint32_t shift_in(uint32_t r, unsigned long long flag_mask) {
    asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(r) : "s"(flag_mask) : "vcc");
    return r;
}

// STORE_MUT: --store_mutations variant (keeps the row bookkeeping out of the common kernel's register budget)
// PLAIN: the record holds nothing but A/C/G/T and there is no custom fragment length (no irregular pairs) -- the
// common case runs without the tests, masks and zero-initialisations of the other two.
//
// Work layout: a wavefront holds 16 pairs, four lanes each; per iteration lane j of a pair takes superitem 4 * i + j
// (8 read positions of both mates = 16 bases: three Philox blocks, two 8-base genome windows, 16 table lookups,
// four 8-byte stores -- the four lanes of a pair write 32 contiguous bytes of each output row).  Everything that
// belongs to the pair (descriptor, Philox address, row offsets of its bin slots) is loaded once and stays in registers
// for the pair's iterations.
#ifndef ISS_MAIN_OCC
#define ISS_MAIN_OCC 4   // wavefronts per SIMD the register budget is cut for: one 1024-lane workgroup per CU, 128 VGPRs (measured against 8 / 64: -12 % time)
#endif
// INDEL: models whose reads often have indels (k_indel_scan and k_indel_script ran in front of this launch).  A read with an
// event has an edit script (SC_* below): per 8-position piece either "the template, shifted by sh tokens" -- the lane then
// takes its 8-base window from the shifted genome position, the same funnel shift as ever -- or 8 explicit letters as
// 2-bit codes in the window's own format.  The letters mut_sequence tests are the final ones (generator.py:152-154), so no
// piece is written twice and nothing is listed for later.
template <bool STORE_MUT, bool PLAIN, bool INDEL>
__global__ __launch_bounds__(MAIN_THREADS, ISS_MAIN_OCC) void k_main(DevModel M, DevGenome g, RunArgs A,
                                                       const PairDesc *__restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    uint32_t *const lds = lds_all + MAIN_LUT_WORDS;  // (the tables; the letter tables stand in front of them)
    // workgroups are dealt to the position tiles in proportion to the tiles' sizes (the last tile may be short)
    int tile = 0;
    while (tile + 1 < M.n_tiles && blockIdx.x >= A.tile_wg0[tile + 1]) ++tile;
    const uint32_t wg = blockIdx.x - A.tile_wg0[tile], n_wg = (uint32_t)A.tile_wg0[tile + 1] - A.tile_wg0[tile];
    MainTile T;
    T.s0 = tile * M.TS;
    T.ts = (uint32_t)min(M.TS, M.S - T.s0);
    // Deferred lane-items (a base needs the exact path): a private ring per wavefront -- no atomics, no barriers.
    // Entry = {(pass << it_bits | iteration) << 19 | lane << 13 | bin slots << 8 | (INDEL) "mate was built from an edit script" bits 0-1,
    //          forward window | complemented reverse window << 16 (2-bit codes of the lane-item's 8 + 8 template bases),
    //          base mask: bit 15 - (8 * half + s) <=> base s = mate * 4 + cc of that half of the superitem};
    // head / tail are wave-uniform.
    uint32_t *ring = lds + M.tile_words + MAIN_MUT_WORDS + 2 * M.TP * 4 + (threadIdx.x >> 6) * (SLOW_RING * 3);
    uint32_t q_head = 0, q_tail = 0;
    const uint32_t lane = threadIdx.x & 63u;
    main_stage_tables(M, tile, lds_all);  // this tile's tables into LDS (once per workgroup)
    __syncthreads();
    // (row offsets below are absolute LDS addresses: ds_read2's offsets are too narrow to skip the letter tables)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_all;
    if (lds0 != 0u) __builtin_trap();  // (lut_at: the letter tables are addressed from LDS address 0)
    auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t n_iter = sgpr((T.ts + 3u) >> 2);
    const uint32_t it_max = ((uint32_t)M.TS + 3u) / 4u - 1u;  // largest iteration number of a pass
    const uint32_t it_bits = sgpr(it_max ? 32u - (uint32_t)__clz(it_max) : 0u);  // (13 bits hold pass and iteration)
    const uint32_t n_pass = sgpr(((uint32_t)A.n_pairs + MAIN_PAIRS - 1) / MAIN_PAIRS);
    // a workgroup's passes (256 pairs each): every n_wg-th block of the launch (the chip's concurrent writes stay in one moving
    // window of rows; contiguous ranges per workgroup were measured 15 % slower)
    const uint32_t blk_first = sgpr(wg), blk_step = sgpr(n_wg), blk_end = n_pass;
    const uint32_t gsh = 16u - (uint32_t)M.GB, gb = (uint32_t)M.GB;
    const uint32_t stride_b = (uint32_t)M.stride_w * 4u, gbytes = 1u << M.GB;
    const uint32_t gs_b = (uint32_t)M.GS * 4u;
    const uint32_t slot_b = (uint32_t)M.TG * gs_b;  // bytes per (mate, bin slot)
    // byte offsets of the rows of the 8 positions of a superitem (two position groups)
    uint32_t off_g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) off_g[c] = sgpr((uint32_t)(c >> 2) * gs_b + (uint32_t)(c & 3) * stride_b);
    const char *const packed_b = reinterpret_cast<const char *>(g.packed - 1);  // the leading padding word: offsets >= 0
    MutChunk mchunk = {0u, MUT_CHUNK};  // --store_mutations: forces a reservation at first use
    const uint32_t wave_pair0 = (threadIdx.x >> 6) * 16u;
    // one round of the exact path: lane k takes ONE base of the k-th pending entry of this wavefront (n <= 64 of them);
    // an entry with more bases (noisy models: NextSeq, MiSeq) goes back into the ring, so every round runs full
    // instead of looping until the lane with the most bases is done (at most 64 come back for the 64 taken out)
    auto drain_round = [&](uint32_t n) {
        // (the byte patches below follow this wavefront's own stores of the same lines: vector memory instructions of
        //  one wavefront reach a given address in issue order, no wait is needed)
        uint32_t rest_m = 0u, ent_x = 0u, ent_y = 0u;
        int subst = 0;
        MutRecord rec;
        rec.position = 0; rec.mate = 0; rec.ref = 0;
        if (lane < n) {
            const uint32_t *ep = ring + ((q_head + lane) & (SLOW_RING - 1)) * 3;
            ent_x = ep[0]; ent_y = ep[1];
            const uint32_t e_pass = ent_x >> (19u + it_bits), e_it = (ent_x >> 19) & ((1u << it_bits) - 1u), e_lane = (ent_x >> 13) & 63u;
            const uint32_t e_pair = __umul24(blk_first + __umul24(e_pass, blk_step), (uint32_t)MAIN_PAIRS) + wave_pair0 + (e_lane >> 2);
            const uint32_t mask = ep[2];  // never empty
            const int bit = 31 - __clz(mask);
            rest_m = mask & ~(1u << bit);
            subst = main_slow_base<PLAIN, INDEL, STORE_MUT>(M, g, A, desc, lds, T, e_pair, 4u * e_it + (e_lane & 3u), (15 - bit) >> 3,
                                                             (15 - bit) & 7, (ent_x >> 8) & 15u, ent_y,
                                                             INDEL ? (ent_x >> (((15 - bit) & 7) >> 2)) & 1u : 0u, rec);
            if (STORE_MUT) mut_emit(A, mchunk, subst == 3, rec);
        }
        q_head += n;
        const unsigned long long again = __ballot(rest_m != 0u);
        if (again) {
            if (rest_m) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(again >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)again, 0u));
                uint32_t *ep = ring + ((q_tail + rank) & (SLOW_RING - 1)) * 3;
                ep[0] = ent_x; ep[1] = ent_y; ep[2] = rest_m;
            }
            q_tail += (uint32_t)__popcll(again);
        }
    };
    // tag: (pass << it_bits | iteration) << 19 | lane << 13 | bin slots << 8; rare: the base mask of the lane-item's 16 bases
    auto push = [&](uint32_t rare, uint32_t tag, uint32_t windows) {
        const unsigned long long rm = __ballot(rare != 0u);
        if (rm) {
            if (rare) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
                uint32_t *ep = ring + ((q_tail + rank) & (SLOW_RING - 1)) * 3;
                ep[0] = tag; ep[1] = windows; ep[2] = rare;
            }
            q_tail += (uint32_t)__popcll(rm);
            while (q_tail - q_head >= 64u) drain_round(64u);
        }
    };
    const uint32_t j4 = lane & 3u;
    // (the descriptor of the NEXT pass is requested at the start of a pass: its latency hides behind the pass)
    PairDesc d_next = {0, 0, 0u, 0};
    // INDEL: the event counters of a pair's two reads (!= 0 <=> the read has an edit script) are requested two passes ahead,
    // this lane's script rows one pass ahead: both have a whole pass to arrive
    uint2 ec_ahead = {0u, 0u};
    uint4 scn_f = make_uint4(SC_IDLE, SC_IDLE, 0u, 0u), scn_r = scn_f;
    uint32_t hv_next = 0u;  // bit mate: that read of the next pass has a script
    auto sc_rows = [&](uint32_t pair_x, uint2 ec, uint32_t grp, uint4 &rf, uint4 &rr) {
        const uint8_t *row = A.script + (size_t)(2u * pair_x) * (size_t)(uint32_t)M.sc_stride +
                             (((uint32_t)tile * (uint32_t)M.sc_gpt + grp) * 4u + j4) * 16u;
        if (ec.x) rf = *reinterpret_cast<const uint4 *>(row);
        if (ec.y) rr = *reinterpret_cast<const uint4 *>(row + M.sc_stride);
    };
    {
        const uint32_t pair0 = blk_first * MAIN_PAIRS + wave_pair0 + (lane >> 2);
        if (blk_first < blk_end && pair0 < (uint32_t)A.n_pairs) {
            d_next = desc[pair0];
            if (INDEL) {  // (the first pass of a workgroup waits for this chain once)
                const uint2 ec0 = *reinterpret_cast<const uint2 *>(A.ev_count + 2u * pair0);
                const uint32_t pair1 = pair0 + blk_step * MAIN_PAIRS;
                if (blk_first + blk_step < blk_end && pair1 < (uint32_t)A.n_pairs) ec_ahead = *reinterpret_cast<const uint2 *>(A.ev_count + 2u * pair1);
                hv_next = (ec0.x ? 1u : 0u) | (ec0.y ? 2u : 0u);
                sc_rows(pair0, ec0, 0u, scn_f, scn_r);
            }
        }
    }
    for (uint32_t pass = 0, blk = blk_first; blk < blk_end; ++pass, blk += blk_step) {
        const uint32_t pair = blk * MAIN_PAIRS + wave_pair0 + (lane >> 2);
        const bool valid = pair < (uint32_t)A.n_pairs;
        const PairDesc d = d_next;
        const uint32_t has_ev = INDEL ? hv_next : 0u;  // bit mate: that read is built from its edit script
        uint4 sc_f = scn_f, sc_r = scn_r;
        if (tile == 0 && j4 == 0u && valid) A.desc_out[pair] = d;
        // The requests for the NEXT pass (descriptor; INDEL: counters two passes ahead, script rows).  They are issued in the
        // pass's first iteration BEHIND its genome windows: vmcnt counts in issue order, so the wait for the windows -- hits of
        // the L2 -- at the end of that iteration would otherwise also be a wait for these cold lines from HBM.
        auto request_next = [&]() {
            const uint32_t pair_n = pair + blk_step * MAIN_PAIRS;
            const bool valid_n = blk + blk_step < blk_end && pair_n < (uint32_t)A.n_pairs;
            if (valid_n) d_next = desc[pair_n];
            if (INDEL) {
                const uint2 ec = ec_ahead;  // the next pass's counters (zero beyond the launch's pairs)
                ec_ahead = make_uint2(0u, 0u);
                const uint32_t pair_nn = pair_n + blk_step * MAIN_PAIRS;
                if (blk + 2u * blk_step < blk_end && pair_nn < (uint32_t)A.n_pairs) ec_ahead = *reinterpret_cast<const uint2 *>(A.ev_count + 2u * pair_nn);
                hv_next = (ec.x ? 1u : 0u) | (ec.y ? 2u : 0u);
                scn_f = make_uint4(SC_IDLE, SC_IDLE, 0u, 0u);
                scn_r = scn_f;
                if (valid_n) sc_rows(pair_n, ec, 0u, scn_f, scn_r);
            }
        };
        const Addr a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
        // LDS byte offsets of the pair's rows (its bin slots) at this lane's first superitem; one iteration = 8 groups on
        const uint32_t lane_row = j4 * 2u * gs_b + (uint32_t)MAIN_LUT_WORDS * 4u + lds0;
        uint32_t rowf = __umul24(d.meta & 3u, slot_b) + lane_row, rowf_e = rowf + gbytes;
        uint32_t rowr = __umul24((uint32_t)M.NB + ((d.meta >> 2) & 3u), slot_b) + lane_row, rowr_e = rowr + gbytes;
        const uint32_t s_lane = (uint32_t)T.s0 + j4;
        // genome position of the lane's first forward base / lowest genome position of its 8 reverse bases, as (word of the packed
        // genome incl. its leading padding word, base within the word): positions need up to 35 bits (records of 2^31 bases and
        // more), word numbers 32 -- and the base within the word is the same for every iteration of the pass
        const int64_t pf64 = desc_fs(d) + (int64_t)(s_lane * 8u), pr64 = desc_re(d) - 8 - (int64_t)(s_lane * 8u);
        uint32_t pfw = (uint32_t)((pf64 >> 4) + 1), prw = (uint32_t)((pr64 >> 4) + 1);
        const uint32_t pfs = (uint32_t)pf64 & 15u, prs = (uint32_t)pr64 & 15u;
        // rows: a 64-bit scalar base per pass (the row of the block's first pair) + a 32-bit lane offset below 256 rows -- a launch's
        // rows may span more than 2^32 bytes (round 5: 5 M MiSeq pairs are ONE launch), and the stores keep their scalar-base form
        uint8_t *const out_pass = A.out[0] + (size_t)blk * (size_t)MAIN_PAIRS * (size_t)(uint32_t)M.row;
        uint32_t out_b = (wave_pair0 + (lane >> 2)) * (uint32_t)M.row + (s_lane >> 2) * 128u + (s_lane & 3u) * 16u;  // (tiles start at multiples of 4 superitems: whole lines)
        const bool regular = PLAIN || !(A.has_frag && (d.meta & 64u));  // irregular pairs are built by the fix-up kernel
        const uint32_t tag0 = (pass << (19u + it_bits)) | (lane << 13) | ((d.meta & 15u) << 8) | has_ev;
        for (uint32_t it = 0; it < n_iter; ++it) {
            uint32_t rare0 = 0, windows = 0;
            if (valid && 4u * it + j4 < T.ts) {
                const uint32_t s_abs = s_lane + 4u * it;
                // ---- the two 8-base windows of the 2-bit genome: forward g[pf .. pf+7]; reverse comp(g[pr+7 .. pr]); loaded
                //      first, used last (the wait for them would otherwise also be a wait for the previous stores)
                uint2 gf = {0u, 0u}, gr = {0u, 0u};
                // INDEL: this piece of each mate per its script row -- shifted template (the window moves) or explicit codes
                uint32_t pfw_e = pfw, prw_e = prw, pfs_e = pfs, prs_e = prs;
                uint32_t code_f = 0u, code_r = 0u;
                bool ex_f = false, ex_r = false;
                if (INDEL) {
                    if ((it & 7u) == 0u && it != 0u) {  // long tiles: the rows of the next 8 iterations (a wait; no shipped model)
                        sc_f = make_uint4(SC_IDLE, SC_IDLE, 0u, 0u);
                        sc_r = sc_f;
                        sc_rows(pair, make_uint2(has_ev & 1u, has_ev & 2u), it >> 3, sc_f, sc_r);
                    }
                    const uint32_t bf = sc_f.x & 0xffu, br = sc_r.x & 0xffu;
                    ex_f = bf >= 128u;
                    ex_r = br >= 128u;
                    const int32_t tf = (int32_t)pfs + (ex_f ? 0 : (int32_t)bf - 64), tr = (int32_t)prs - (ex_r ? 0 : (int32_t)br - 64);
                    pfw_e = pfw + (uint32_t)(tf >> 4); pfs_e = (uint32_t)tf & 15u;
                    prw_e = prw + (uint32_t)(tr >> 4); prs_e = (uint32_t)tr & 15u;
                    code_f = (uint32_t)((((uint64_t)sc_f.w << 32) | sc_f.z) >> (bf & 63u));  // (128 + 16 k) & 63 = 16 k
                    code_r = (uint32_t)((((uint64_t)sc_r.w << 32) | sc_r.z) >> (br & 63u));
                    sc_f.x = __builtin_amdgcn_alignbit(sc_f.y, sc_f.x, 8u); sc_f.y >>= 8;
                    sc_r.x = __builtin_amdgcn_alignbit(sc_r.y, sc_r.x, 8u); sc_r.y >>= 8;
                }
                if (regular) {
                    gf = *reinterpret_cast<const uint2 *>(packed_b + (size_t)(pfw_e << 2));
                    gr = *reinterpret_cast<const uint2 *>(packed_b + (size_t)(prw_e << 2));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (it == 0u) {  // (every lane with work in this tile has an iteration 0; the others need no descriptor)
                    request_next();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- hot digits: 16 quality digits (16 bits) + 16 error-test digits (8 bits)
                const u32x4 q0 = draw_block(a, K_QM, s_abs, 0);
                const u32x4 ee = draw_block(a, K_QM, s_abs, 1);
                const u32x4 q1 = draw_block(a, K_QM, s_abs, 2);
                // ---- phred scores + substitution test, loop-free (hot_lookup); 16 independent lookups
                uint32_t sel[16];
                unsigned long long fl;
#define ISS_LOOKUP(K, ROW, C, HI, BYTE, WQ, WE, RARE)                                                                \
                sel[K] = hot_lookup<HI, BYTE>(ROW, ROW##_e, off_g[C], WQ, WE, gsh, gb, fl);                            \
                RARE = shift_in(RARE, fl);
                // half 0: positions 0-3; bit 15 - (8 * half + s) of rare0 <=> base s = mate * 4 + cc of that half
                ISS_LOOKUP(0, rowf, 0, 0, 0, q0.x, ee.x, rare0)
                ISS_LOOKUP(1, rowf, 1, 1, 1, q0.x, ee.x, rare0)
                ISS_LOOKUP(2, rowf, 2, 0, 2, q0.y, ee.x, rare0)
                ISS_LOOKUP(3, rowf, 3, 1, 3, q0.y, ee.x, rare0)
                ISS_LOOKUP(4, rowr, 0, 0, 0, q0.z, ee.y, rare0)
                ISS_LOOKUP(5, rowr, 1, 1, 1, q0.z, ee.y, rare0)
                ISS_LOOKUP(6, rowr, 2, 0, 2, q0.w, ee.y, rare0)
                ISS_LOOKUP(7, rowr, 3, 1, 3, q0.w, ee.y, rare0)
                // half 1: positions 4-7
                ISS_LOOKUP(8, rowf, 4, 0, 0, q1.x, ee.z, rare0)
                ISS_LOOKUP(9, rowf, 5, 1, 1, q1.x, ee.z, rare0)
                ISS_LOOKUP(10, rowf, 6, 0, 2, q1.y, ee.z, rare0)
                ISS_LOOKUP(11, rowf, 7, 1, 3, q1.y, ee.z, rare0)
                ISS_LOOKUP(12, rowr, 4, 0, 0, q1.z, ee.w, rare0)
                ISS_LOOKUP(13, rowr, 5, 1, 1, q1.z, ee.w, rare0)
                ISS_LOOKUP(14, rowr, 6, 0, 2, q1.w, ee.w, rare0)
                ISS_LOOKUP(15, rowr, 7, 1, 3, q1.w, ee.w, rare0)
#undef ISS_LOOKUP
                // phred bytes: byte 1 of each selected entry
                auto quals = [&](int k) {
                    return __builtin_amdgcn_perm(sel[k + 1], sel[k], 0x0c0c0501u) | __builtin_amdgcn_perm(sel[k + 3], sel[k + 2], 0x05010c0cu);
                };
                uint2 qual_f = {quals(0), quals(8)}, qual_r = {quals(4), quals(12)};
                // ---- template bases
                uint32_t fm = 0, rm = 0;
                uint32_t fb = funnel_r(gf.x, gf.y, pfs_e * 2u);
                uint32_t rbr = funnel_r(gr.x, gr.y, prs_e * 2u);
                if (INDEL) { fb = ex_f ? code_f : fb; rbr = ex_r ? code_r : rbr; }
                windows = __builtin_amdgcn_perm(rbr ^ 0x5555u, fb, 0x05040100u);  // fb[15:0] | complemented (code ^ 1) rb[15:0] << 16
                if (!PLAIN && regular && (d.meta & 0x30u)) {  // only pairs whose windows hold IUPAC / lower-case letters (k_setup)
                    // (a scripted mate lies in a record of plain A/C/G/T -- k_indel_script -- so pf_e / pr_e differ from pf / pr
                    //  only where the mask is clear)
                    // (genome position = 16 * (word - 1) + base within the word; mask word = position >> 5)
                    const uint32_t *mw = g.mask + ((int32_t)(pfw_e - 1u) >> 1);
                    fm = funnel_r(mw[0], mw[1], ((pfw_e - 1u) & 1u) * 16u + pfs_e) & 0xffu;
                    const uint32_t *nw = g.mask + ((int32_t)(prw_e - 1u) >> 1);
                    rm = funnel_r(nw[0], nw[1], ((prw_e - 1u) & 1u) * 16u + prs_e) & 0xffu;
                }
                // (letters from the LDS tables: forward a byte of codes as it stands; reverse mate: read position c <-> genome
                //  position pr + 7 - c, complemented)
                uint2 base_f = {lut_at<0, 0>(lds0, fb), lut_at<1, 0>(lds0, fb)};
                uint2 base_r = {lut_at<1, 1024>(lds0, rbr), lut_at<0, 1024>(lds0, rbr)};
                if (!PLAIN && (fm | rm)) {  // IUPAC / lower-case letters: patch from the ASCII copy
                    for (int c = 0; c < 8; ++c) {
                        if ((fm >> c) & 1u) {
                            const uint32_t ch = g.ascii[(int64_t)(int32_t)(pfw_e - 1u) * 16 + pfs_e + c];
                            uint32_t &w = c < 4 ? base_f.x : base_f.y;
                            w = (w & ~(0xffu << (8 * (c & 3)))) | (ch << (8 * (c & 3)));
                        }
                        if ((rm >> (7 - c)) & 1u) {
                            const uint32_t ch = (uint32_t)complement_ascii(g.ascii[(int64_t)(int32_t)(prw_e - 1u) * 16 + prs_e + 7 - c]);
                            uint32_t &w = c < 4 ? base_r.x : base_r.y;
                            w = (w & ~(0xffu << (8 * (c & 3)))) | (ch << (8 * (c & 3)));
                        }
                    }
                }
                // (the padding bytes of the last superitem hold the clamped last row's values; nothing reads them)
                // (Whole 128-byte lines per store instruction -- lanes l / l + 32 exchanging a forward for a reverse piece with
                //  v_permlane32_swap, so that one store writes both halves of eight pairs' lines -- measured in round 4: 1.273 /
                //  1.248 against 1.275 / 1.235 ms, interleaved on one box: nothing.  tools/store_bench.hip: the write stream
                //  alone takes 0.56 ms per 5 M pairs in either form.)
                uint4 *dst = reinterpret_cast<uint4 *>(out_pass + (size_t)out_b);  // two 16-byte pieces of the pair's 128-byte line
                dst[0] = make_uint4(base_f.x, base_f.y, qual_f.x, qual_f.y);
                dst[4] = make_uint4(base_r.x, base_r.y, qual_r.x, qual_r.y);
            }
            // bit 15 - (8 * half + s) <=> base s of that half needs the exact path (~0.9 % of bases: mostly substitution tests that tie)
            push(rare0, tag0 | (it << 19), windows);
            rowf += 8u * gs_b; rowf_e += 8u * gs_b;
            rowr += 8u * gs_b; rowr_e += 8u * gs_b;
            pfw += 2u;
            prw -= 2u;
            out_b += 128u;
        }
    }
    while (q_tail != q_head) drain_round(min(64u, q_tail - q_head));
}

// ================================================================== k_main_g
// k_main with the exact path's byte patches IN TIME (round 6; plain launches: no --store_mutations, no edit scripts).
//
// What a late patch costs (rounds 3-5): a full line leaves the L2 within microseconds of being written; a
// byte stored into it later is a masked write of a line the L2 no longer holds -- a read-modify-write behind the L2, ~200 bytes
// of HBM time each: 10 % of a NovaSeq launch, 17-20 % of a HiSeq launch, 1.76 x the algorithmic bytes for MiSeq.  A byte stored
// RIGHT BEHIND its row is free (tools/store_bench.hip: "delay 0").  So the rows wait: a workgroup's iterations are taken in
// GROUPS of G = NP passes x NI iterations (the loops unrolled, the rows of a group -- 8 registers per iteration -- held in
// registers), the wavefront's pending lane-items are settled at the END of the group in one round of the exact path whose
// patches are handed back (main_slow_base<.., HOLD>), then the group's rows are stored and the patches follow them out, from
// whichever lane computed them: a wavefront's vector memory operations reach an address in issue order.  Entries beyond the 64
// of that round, and the second base of a lane-item with two, are settled behind the rows as before (their patches are late or
// nearly in time, never early).  The ring holds 128 entries and a slot pushes up to 64: when the next slot might not fit, the
// rows computed so far are stored and rounds run at once (models that defer most of their bases; every test configuration).
// min_round: a group's closing round runs only with at least that many entries pending (fewer: they wait for the next group,
// their patches will be late -- 64 gives full rounds only).
// Everything else -- work layout, Philox addresses, lookups, ring entries -- is k_main's.
template <bool PLAIN, int NI, int NP>
__global__ __launch_bounds__(MAIN_THREADS, ISS_MAIN_OCC) void k_main_g(DevModel M, DevGenome g, RunArgs A,
                                                                       const PairDesc *__restrict__ desc, uint32_t min_round) {
    constexpr int G = NI * NP;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    uint32_t *const lds = lds_all + MAIN_LUT_WORDS;
    int tile = 0;
    while (tile + 1 < M.n_tiles && blockIdx.x >= A.tile_wg0[tile + 1]) ++tile;
    const uint32_t wg = blockIdx.x - A.tile_wg0[tile], n_wg = (uint32_t)A.tile_wg0[tile + 1] - A.tile_wg0[tile];
    MainTile T;
    T.s0 = tile * M.TS;
    T.ts = (uint32_t)min(M.TS, M.S - T.s0);
    uint32_t *ring = lds + M.tile_words + MAIN_MUT_WORDS + 2 * M.TP * 4 + (threadIdx.x >> 6) * (SLOW_RING * 3);
    uint32_t q_head = 0, q_tail = 0;
    const uint32_t lane = threadIdx.x & 63u;
    main_stage_tables(M, tile, lds_all);
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_all;
    if (lds0 != 0u) __builtin_trap();  // (lut_at: the letter tables are addressed from LDS address 0)
    auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t it_max = ((uint32_t)M.TS + 3u) / 4u - 1u;  // (== NI - 1: the host picks the instantiation)
    const uint32_t it_bits = sgpr(it_max ? 32u - (uint32_t)__clz(it_max) : 0u);
    const uint32_t n_pass = sgpr(((uint32_t)A.n_pairs + MAIN_PAIRS - 1) / MAIN_PAIRS);
    const uint32_t blk_first = sgpr(wg), blk_step = sgpr(n_wg), blk_end = n_pass;
    const uint32_t gsh = 16u - (uint32_t)M.GB, gb = (uint32_t)M.GB;
    const uint32_t stride_b = (uint32_t)M.stride_w * 4u, gbytes = 1u << M.GB;
    const uint32_t gs_b = (uint32_t)M.GS * 4u;
    const uint32_t slot_b = (uint32_t)M.TG * gs_b;
    uint32_t off_g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) off_g[c] = sgpr((uint32_t)(c >> 2) * gs_b + (uint32_t)(c & 3) * stride_b);
    const char *const packed_b = reinterpret_cast<const char *>(g.packed - 1);
    const uint32_t wave_pair0 = (threadIdx.x >> 6) * 16u;
    const uint32_t j4 = lane & 3u;
    const uint32_t s_lane = (uint32_t)T.s0 + j4;
    const uint32_t pass_bytes = sgpr(blk_step * (uint32_t)MAIN_PAIRS * (uint32_t)M.row);  // rows of one pass of the grid (the host keeps NP of them below 2^31)
    // one round of the exact path (k_main's); HOLD: the patch of this lane's base comes back in `held` instead of being stored
    auto drain_round = [&](uint32_t n, auto hold, HeldPatch &held) __attribute__((always_inline)) {
        constexpr bool HOLD = decltype(hold)::value;
        uint32_t rest_m = 0u, ent_x = 0u, ent_y = 0u;
        MutRecord rec;
        rec.position = 0; rec.mate = 0; rec.ref = 0;
        if (lane < n) {
            const uint32_t *ep = ring + ((q_head + lane) & (SLOW_RING - 1)) * 3;
            ent_x = ep[0]; ent_y = ep[1];
            const uint32_t e_pass = ent_x >> (19u + it_bits), e_it = (ent_x >> 19) & ((1u << it_bits) - 1u), e_lane = (ent_x >> 13) & 63u;
            const uint32_t e_pair = __umul24(blk_first + __umul24(e_pass, blk_step), (uint32_t)MAIN_PAIRS) + wave_pair0 + (e_lane >> 2);
            const uint32_t mask = ep[2];  // never empty
            const int bit = 31 - __clz(mask);
            rest_m = mask & ~(1u << bit);
            (void)main_slow_base<PLAIN, false, false, HOLD>(M, g, A, desc, lds, T, e_pair, 4u * e_it + (e_lane & 3u), (15 - bit) >> 3,
                                                            (15 - bit) & 7, (ent_x >> 8) & 15u, ent_y, 0u, rec, &held);
        }
        q_head += n;
        const unsigned long long again = __ballot(rest_m != 0u);
        if (again) {
            if (rest_m) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(again >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)again, 0u));
                uint32_t *ep = ring + ((q_tail + rank) & (SLOW_RING - 1)) * 3;
                ep[0] = ent_x; ep[1] = ent_y; ep[2] = rest_m;
            }
            q_tail += (uint32_t)__popcll(again);
        }
    };
    auto push = [&](uint32_t rare, uint32_t tag, uint32_t windows) __attribute__((always_inline)) {  // (no round in here: the rows of the group wait)
        const unsigned long long rm = __ballot(rare != 0u);
        if (rm) {
            if (rare) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
                uint32_t *ep = ring + ((q_tail + rank) & (SLOW_RING - 1)) * 3;
                ep[0] = tag; ep[1] = windows; ep[2] = rare;
            }
            q_tail += (uint32_t)__popcll(rm);
        }
    };
    PairDesc d_next[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        d_next[pp] = PairDesc{0, 0, 0u, 0};
        const uint32_t blk_p = blk_first + (uint32_t)pp * blk_step;
        const uint32_t pair0 = blk_p * MAIN_PAIRS + wave_pair0 + (lane >> 2);
        if (blk_p < blk_end && pair0 < (uint32_t)A.n_pairs) d_next[pp] = desc[pair0];
    }
    struct Sub {  // a pass of the group, per lane
        uint32_t meta;  // PairDesc::meta
        Addr a;
        uint32_t rowf, rowr, pfw, prw, pfs, prs, out_b, tag0, s_lane;
        bool valid, regular;
    };
    for (uint32_t pass = 0, blk = blk_first; blk < blk_end; pass += NP, blk += (uint32_t)NP * blk_step) {
        Sub sub[NP];
        uint32_t rows[G][8];  // per slot: forward letters (2), forward phreds (2), reverse letters (2), reverse phreds (2)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            Sub &S = sub[pp];
            const uint32_t blk_p = blk + (uint32_t)pp * blk_step;
            const uint32_t pair = blk_p * MAIN_PAIRS + wave_pair0 + (lane >> 2);
            S.valid = blk_p < blk_end && pair < (uint32_t)A.n_pairs;
            const PairDesc d = d_next[pp];
            S.meta = d.meta;
            if (tile == 0 && j4 == 0u && S.valid) A.desc_out[pair] = d;
            S.a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
            const uint32_t lane_row = j4 * 2u * gs_b + (uint32_t)MAIN_LUT_WORDS * 4u + lds0;
            S.rowf = __umul24(d.meta & 3u, slot_b) + lane_row;
            S.rowr = __umul24((uint32_t)M.NB + ((d.meta >> 2) & 3u), slot_b) + lane_row;
            const int64_t pf64 = desc_fs(d) + (int64_t)(s_lane * 8u), pr64 = desc_re(d) - 8 - (int64_t)(s_lane * 8u);
            S.pfw = (uint32_t)((pf64 >> 4) + 1); S.prw = (uint32_t)((pr64 >> 4) + 1);
            S.pfs = (uint32_t)pf64 & 15u; S.prs = (uint32_t)pr64 & 15u;
            S.out_b = (wave_pair0 + (lane >> 2)) * (uint32_t)M.row + (s_lane >> 2) * 128u + (s_lane & 3u) * 16u + (uint32_t)pp * pass_bytes;
            S.regular = PLAIN || !(A.has_frag && (d.meta & 64u));
            S.tag0 = ((pass + (uint32_t)pp) << (19u + it_bits)) | (lane << 13) | ((d.meta & 15u) << 8);
            S.s_lane = s_lane;
        }
        uint8_t *const out_pass = A.out[0] + (size_t)blk * (size_t)MAIN_PAIRS * (size_t)(uint32_t)M.row;
        auto request_next = [&]() __attribute__((always_inline)) {  // the descriptors of the next group: behind the first slot's genome windows (k_main)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const uint32_t blk_n = blk + (uint32_t)(NP + pp) * blk_step;
                const uint32_t pair_n = blk_n * MAIN_PAIRS + wave_pair0 + (lane >> 2);
                if (blk_n < blk_end && pair_n < (uint32_t)A.n_pairs) d_next[pp] = desc[pair_n];
            }
        };
        auto slot = [&](const int j) __attribute__((always_inline)) {  // iteration it of pass pp of the group: k_main's hot block, the row kept
            const int pp = j / NI;
            const uint32_t it = (uint32_t)(j % NI);
            const Sub &S = sub[pp];
            uint32_t rare0 = 0, windows = 0;
            if (S.valid && 4u * it + j4 < T.ts) {
                const uint32_t s_abs = S.s_lane + 4u * it, pfw_e = S.pfw + 2u * it, prw_e = S.prw - 2u * it;
                uint32_t rowf = S.rowf, rowr = S.rowr;
                uint2 gf = {0u, 0u}, gr = {0u, 0u};
                if (S.regular) {
                    gf = *reinterpret_cast<const uint2 *>(packed_b + (size_t)(pfw_e << 2));
                    gr = *reinterpret_cast<const uint2 *>(packed_b + (size_t)(prw_e << 2));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) {
                    request_next();
                    __builtin_amdgcn_sched_barrier(0);
                }
                const u32x4 q0 = draw_block(S.a, K_QM, s_abs, 0);
                const u32x4 ee = draw_block(S.a, K_QM, s_abs, 1);
                const u32x4 q1 = draw_block(S.a, K_QM, s_abs, 2);
                rowf += it * 8u * gs_b;
                rowr += it * 8u * gs_b;
                const uint32_t rowf_e = rowf + gbytes, rowr_e = rowr + gbytes;
                uint32_t sel[16];
                unsigned long long fl;
#define ISS_LOOKUP(K, ROW, C, HI, BYTE, WQ, WE, RARE)                                                                \
                sel[K] = hot_lookup<HI, BYTE>(ROW, ROW##_e, off_g[C], WQ, WE, gsh, gb, fl);                            \
                RARE = shift_in(RARE, fl);
                ISS_LOOKUP(0, rowf, 0, 0, 0, q0.x, ee.x, rare0)
                ISS_LOOKUP(1, rowf, 1, 1, 1, q0.x, ee.x, rare0)
                ISS_LOOKUP(2, rowf, 2, 0, 2, q0.y, ee.x, rare0)
                ISS_LOOKUP(3, rowf, 3, 1, 3, q0.y, ee.x, rare0)
                ISS_LOOKUP(4, rowr, 0, 0, 0, q0.z, ee.y, rare0)
                ISS_LOOKUP(5, rowr, 1, 1, 1, q0.z, ee.y, rare0)
                ISS_LOOKUP(6, rowr, 2, 0, 2, q0.w, ee.y, rare0)
                ISS_LOOKUP(7, rowr, 3, 1, 3, q0.w, ee.y, rare0)
                ISS_LOOKUP(8, rowf, 4, 0, 0, q1.x, ee.z, rare0)
                ISS_LOOKUP(9, rowf, 5, 1, 1, q1.x, ee.z, rare0)
                ISS_LOOKUP(10, rowf, 6, 0, 2, q1.y, ee.z, rare0)
                ISS_LOOKUP(11, rowf, 7, 1, 3, q1.y, ee.z, rare0)
                ISS_LOOKUP(12, rowr, 4, 0, 0, q1.z, ee.w, rare0)
                ISS_LOOKUP(13, rowr, 5, 1, 1, q1.z, ee.w, rare0)
                ISS_LOOKUP(14, rowr, 6, 0, 2, q1.w, ee.w, rare0)
                ISS_LOOKUP(15, rowr, 7, 1, 3, q1.w, ee.w, rare0)
#undef ISS_LOOKUP
                auto quals = [&](int k) {
                    return __builtin_amdgcn_perm(sel[k + 1], sel[k], 0x0c0c0501u) | __builtin_amdgcn_perm(sel[k + 3], sel[k + 2], 0x05010c0cu);
                };
                uint32_t fm = 0, rm = 0;
                const uint32_t fb = funnel_r(gf.x, gf.y, S.pfs * 2u);
                const uint32_t rbr = funnel_r(gr.x, gr.y, S.prs * 2u);
                windows = __builtin_amdgcn_perm(rbr ^ 0x5555u, fb, 0x05040100u);
                if (!PLAIN && S.regular && (S.meta & 0x30u)) {
                    const uint32_t *mw = g.mask + ((int32_t)(pfw_e - 1u) >> 1);
                    fm = funnel_r(mw[0], mw[1], ((pfw_e - 1u) & 1u) * 16u + S.pfs) & 0xffu;
                    const uint32_t *nw = g.mask + ((int32_t)(prw_e - 1u) >> 1);
                    rm = funnel_r(nw[0], nw[1], ((prw_e - 1u) & 1u) * 16u + S.prs) & 0xffu;
                }
                uint2 base_f = {lut_at<0, 0>(lds0, fb), lut_at<1, 0>(lds0, fb)};
                uint2 base_r = {lut_at<1, 1024>(lds0, rbr), lut_at<0, 1024>(lds0, rbr)};
                if (!PLAIN && (fm | rm)) {
                    for (int c = 0; c < 8; ++c) {
                        if ((fm >> c) & 1u) {
                            const uint32_t ch = g.ascii[(int64_t)(int32_t)(pfw_e - 1u) * 16 + S.pfs + c];
                            uint32_t &w = c < 4 ? base_f.x : base_f.y;
                            w = (w & ~(0xffu << (8 * (c & 3)))) | (ch << (8 * (c & 3)));
                        }
                        if ((rm >> (7 - c)) & 1u) {
                            const uint32_t ch = (uint32_t)complement_ascii(g.ascii[(int64_t)(int32_t)(prw_e - 1u) * 16 + S.prs + 7 - c]);
                            uint32_t &w = c < 4 ? base_r.x : base_r.y;
                            w = (w & ~(0xffu << (8 * (c & 3)))) | (ch << (8 * (c & 3)));
                        }
                    }
                }
                rows[j][0] = base_f.x; rows[j][1] = base_f.y; rows[j][2] = quals(0); rows[j][3] = quals(8);
                rows[j][4] = base_r.x; rows[j][5] = base_r.y; rows[j][6] = quals(4); rows[j][7] = quals(12);
            }
            push(rare0, S.tag0 | (it << 19), windows);
        };
        auto store_slot = [&](const int j) __attribute__((always_inline)) {
            const int pp = j / NI;
            const uint32_t it = (uint32_t)(j % NI);
            const Sub &S = sub[pp];
            if (S.valid && 4u * it + j4 < T.ts) {
                uint4 *dst = reinterpret_cast<uint4 *>(out_pass + (size_t)(S.out_b + it * 128u));
                dst[0] = make_uint4(rows[j][0], rows[j][1], rows[j][2], rows[j][3]);
                dst[4] = make_uint4(rows[j][4], rows[j][5], rows[j][6], rows[j][7]);
            }
        };
        uint32_t k = 0, first = 0;
        for (;;) {
            // (opaque re-definitions: what a slot derives from the pass's values and its own number -- the first Philox round,
            //  row offsets, window words, addresses -- is invariant in this loop; hoisted out of it, it costs five to eight
            //  registers per slot for the whole group and spills)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
                asm volatile("" : "+v"(sub[pp].s_lane), "+v"(sub[pp].rowf), "+v"(sub[pp].rowr), "+v"(sub[pp].pfw), "+v"(sub[pp].prw), "+v"(sub[pp].out_b));
            bool stop = false;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (k == (uint32_t)j && !stop) {
                    slot(j);
                    k = (uint32_t)j + 1u;
                    stop = j + 1 < G && q_tail - q_head > (uint32_t)SLOW_RING - 64u;  // the next slot's pushes might not fit
                }
            }
            const bool end = k == (uint32_t)G;
            HeldPatch held = {0ull, 0u};
            if (end) {  // the group's closing round: its patches wait for the rows
                const uint32_t pending = q_tail - q_head;
                if (pending && pending >= min_round) drain_round(min(64u, pending), std::true_type{}, held);
            }
#pragma unroll
            for (int j = 0; j < G; ++j)
                if ((uint32_t)j >= first && (uint32_t)j < k) store_slot(j);
            first = k;
            if (held.w & (3u << 16)) {
                uint8_t *const at = A.out[0] + held.byte_off + ((held.w >> 18) & 1u) * 64u;  // (row_array_off: mate 64 bytes on, phreds 8)
                if (held.w & (1u << 16)) at[8] = (uint8_t)held.w;
                if (held.w & (1u << 17)) at[0] = (uint8_t)(held.w >> 8);
            }
            HeldPatch none;
            while (q_tail - q_head >= 64u) drain_round(64u, std::false_type{}, none);
            if (end) break;
        }
    }
    HeldPatch none;
    while (q_tail != q_head) drain_round(min(64u, q_tail - q_head), std::false_type{}, none);
}

// ================================================================== k_indel_scan
// One lane per READ: the read's indel events (indel_events above: a draw per firing test + one per segment of the
// survival table, i.e. one for a read without events) in step order into its event list (EV_K words, step << 8 | event
// mask); no event => provably no indel, the read is its template.  k_indel_script turns the lists into edit scripts.  A read with more than
// EV_K events goes to the wavefront-per-read kernel instead (k_indel_fixup: extreme models only).  The reads with events
// are collected per wavefront in LDS and reach the read list >= 64 at a time.  The list is SEGMENTED: a workgroup's
// reads go to read_list[first read of its range ...], the place reserved by an LDS atomic of the workgroup -- a global
// counter everybody adds to takes ~10 ns per add, 0.2-0.5 ms per 5 M pairs of configs[4] -- and the segments' lengths
// to read_count[workgroup]; k_indel_script walks the segments through a prefix sum of their 64-read blocks.  Consecutive
// lanes take consecutive reads, so the list is in pair order, more or less, and k_indel_script's wavefronts share cache
// lines, DRAM pages and TLB entries.
constexpr int SCAN_THREADS = 1024;  // two workgroups per CU: 8 wavefronts / SIMD (the kernel is bound by the latency of its dependent LDS reads)
constexpr int SCAN_LISTN = 96;    // listed reads (two or more event steps) a wavefront collects before they go to the global list
constexpr int SCAN_LIST1 = 128;   // ... reads with one event step
constexpr int EV_K = 8;           // events kept per read
constexpr uint32_t FLAG_LISTED = 16u;  // RunArgs::flags: bits 0-1 mate goes to k_indel_fixup, bits 2-3 unused since round 4,
                                       // bits 4-5 mate is in read_list
constexpr int SCAN_MAX_WGS = 512; // (k_indel_script keeps the segment table in LDS)
__host__ __device__ inline uint32_t scan_per_wg(uint32_t n_reads, uint32_t wgs) {  // reads of a workgroup's contiguous range
    return ((n_reads + wgs - 1) / wgs + SCAN_THREADS - 1) / SCAN_THREADS * SCAN_THREADS;
}
constexpr int SCAN_WIN = 512;     // window of a wavefront's range whose event counts are staged in LDS: they leave in whole lines, half a window at a time
__host__ __device__ inline size_t scan_lds_bytes(int ev_ns) {
    return (size_t)2 * ev_ns * 8 + (((size_t)2 * ev_ns * 2 + 15) & ~(size_t)15) + (size_t)(SCAN_THREADS / 64) * (SCAN_LISTN * 16 + SCAN_LIST1 * 8 + SCAN_WIN * 2);
}

__global__ __launch_bounds__(SCAN_THREADS, 8) void k_indel_scan(DevModel M, RunArgs A, const PairDesc *__restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) uint64_t scan_lds[];
    const int ns = M.ev_ns;
    uint64_t *l_S = scan_lds;                                         // [2][ns]
    uint16_t *l_E = reinterpret_cast<uint16_t *>(l_S + 2 * ns);       // [2][ns]
    uint8_t *l_rest = reinterpret_cast<uint8_t *>(scan_lds) + (size_t)2 * ns * 8 + (((size_t)2 * ns * 2 + 15) & ~(size_t)15);
    uint4 *l_list = reinterpret_cast<uint4 *>(l_rest) + (threadIdx.x >> 6) * SCAN_LISTN;  // this wavefront's listed reads (>= 2 event steps)
    uint2 *l_list1 = reinterpret_cast<uint2 *>(l_rest + (size_t)(SCAN_THREADS / 64) * SCAN_LISTN * 16) + (threadIdx.x >> 6) * SCAN_LIST1;  // ... (one)
    uint16_t *l_cnt = reinterpret_cast<uint16_t *>(l_rest + (size_t)(SCAN_THREADS / 64) * (SCAN_LISTN * 16 + SCAN_LIST1 * 8)) + (threadIdx.x >> 6) * SCAN_WIN;  // ... event counts
    __shared__ uint32_t l_seg[2];  // reads of this workgroup listed so far
    for (int i = threadIdx.x; i < 2 * ns; i += blockDim.x) { l_S[i] = M.ev_S[i]; l_E[i] = M.ev_E[i]; }
    if (threadIdx.x < 2) l_seg[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_reads = 2u * (uint32_t)A.n_pairs;
    const uint32_t per_wg = scan_per_wg(n_reads, gridDim.x);  // contiguous ranges (a wavefront's: a multiple of 64 reads)
    uint4 *const seg_list = A.read_list + (size_t)blockIdx.x * per_wg;  // (a range lists at most its own reads)
    uint2 *const seg_list1 = A.read_list1 + (size_t)blockIdx.x * per_wg;
    uint32_t n_listed = 0, n_listed1 = 0;  // wave-uniform
    auto flush_list = [&]() {  // this wavefront's listed reads -> the workgroup's segment of the list
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&l_seg[0], n_listed);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t i = lane; i < n_listed; i += 64u) seg_list[base + i] = l_list[i];
        n_listed = 0;
    };
    auto flush_list1 = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&l_seg[1], n_listed1);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t i = lane; i < n_listed1; i += 64u) seg_list1[base + i] = l_list1[i];
        n_listed1 = 0;
    };
    const uint32_t per_wave = per_wg / (blockDim.x >> 6);
    const uint32_t w_first = min(n_reads, blockIdx.x * per_wg + (threadIdx.x >> 6) * per_wave), w_last = min(n_reads, w_first + per_wave);
    // A read needs one draw per event and per segment of the survival table -- one for most reads, half a dozen for the
    // unluckiest of 64: the lanes of a wavefront therefore do not walk the range in lockstep.  Every iteration is ONE draw
    // of every lane's current read; a lane whose read is finished takes the next read of the wavefront's range.
    // The event counts of reads [wb, wb + SCAN_WIN) are staged (ring l_cnt, index = read - w_first mod SCAN_WIN): reads are
    // handed out in order, so a window's first half is complete once no lane still holds one of its reads.
    uint32_t next = w_first, wb = w_first;  // wave-uniform
    auto flush_counts = [&](uint32_t n) {  // counts of reads [wb, wb + n) -> global memory, whole lines
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (uint32_t i = lane; i < n; i += 64u) A.ev_count[wb + i] = l_cnt[(wb - w_first + i) & (uint32_t)(SCAN_WIN - 1)];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the slots are reused)
    };
    bool busy = false;
    uint32_t rd = 0, cnt = 0, prev = 0, j = 0, e0 = 0, e1 = 0;  // (e0, e1: the read's first two events; prev: its last)
    int cur = -1, o = 0;
    Addr a = make_addr(A.seed, A.first_ordinal, 0u);
    for (;;) {
        const unsigned long long need = __ballot(!busy);
        bool room = true;
        if (need && next < w_last && next + 64u > wb + (uint32_t)SCAN_WIN) {  // the window is full: retire its first half ...
            if (!__ballot(busy && rd - wb < (uint32_t)(SCAN_WIN / 2))) {
                flush_counts((uint32_t)(SCAN_WIN / 2));
                wb += (uint32_t)(SCAN_WIN / 2);
            } else {
                room = false;  // ... once the lanes still working on it are done (no new reads until then)
            }
        }
        if (need && next < w_last && room) {
            const uint32_t cand = next + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (!busy && cand < w_last) {
                rd = cand; cnt = 0; prev = 0; j = 0; cur = -1; busy = true; e0 = 0; e1 = 0;
                o = (int)(rd & 1u);
                // the attempt number is 0 unless gc_bias re-drew the pair: no descriptor load in the common case
                a = make_addr(A.seed, A.first_ordinal + (rd >> 1), A.gc_bias ? desc[rd >> 1].meta >> 16 : 0u);
                // (the mates of irregular pairs -- custom fragment lengths -- are the fix-up kernel's already: k_setup)
                if (A.has_frag && ((A.flags[rd >> 1] >> (rd & 1u)) & 1u)) cur = ns;
            }
            next = min(w_last, next + (uint32_t)__popcll(need));
        }
        if (!__ballot(busy)) break;
        bool fin = false;
        if (busy) {
            if (cur < ns - 1) {
                const u32x4 w = draw_block(a, K_EV, j++, (uint32_t)o);
                int slot;
                uint32_t mask;
                cur = ev_step(l_S + o * ns, l_E + o * ns, M.ev_T + (size_t)o * ns, M.del_thr + (size_t)o * M.RL * 4, cur, mk53(w.x, w.y), mk53(w.z, w.w), slot, mask);
                if (slot >= 0) {  // (one event per step: a step's tests come one after the other)
                    uint32_t *list = A.ev_list + (size_t)rd * EV_K;  // (written only for reads with more than two events)
                    const int n = slot / 5;
                    if (cnt && (int)(prev >> 8) == n) {
                        prev |= mask;
                    } else {
                        prev = ((uint32_t)n << 8) | mask;
                        ++cnt;
                        if (cnt == 3u) { list[0] = e0; list[1] = e1; }
                    }
                    if (cnt == 1u) e0 = prev; else if (cnt == 2u) e1 = prev; else if (cnt <= (uint32_t)EV_K) list[cnt - 1] = prev;
                }
            }
            if (cur >= ns - 1) {  // the read is done
                if (cnt > (uint32_t)EV_K || (A.light && cnt)) {  // too many events for the list (or few such reads at all): the wavefront-per-read kernel takes the read
                    if (!(atomicOr(&A.flags[rd >> 1], 1u << o) & (1u << o))) A.fix_list[atomicAdd(A.fix_count, 1u)] = rd;
                    cnt = 0;
                }
                // (every read's counter is written: nothing else initialises them -- staged, so that they leave in whole lines)
                const uint32_t evc = cnt ? min(cnt, 15u) | ((e0 >> 8) << 4) : 0u;  // steps with an event (capped) | the first of them << 4
                l_cnt[(rd - w_first) & (uint32_t)(SCAN_WIN - 1)] = (uint16_t)evc;
                fin = true;
                busy = false;
            }
        }
        const unsigned long long m = __ballot(fin && cnt > 1u), m1 = __ballot(fin && cnt == 1u);
        if (m) {
            if (fin && cnt > 1u) l_list[n_listed + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = make_uint4(rd, e0, e1, cnt);
            n_listed += (uint32_t)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (n_listed > (uint32_t)SCAN_LISTN - 64u) flush_list();  // (room for the next 64)
        }
        if (m1) {
            if (fin && cnt == 1u) l_list1[n_listed1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = make_uint2(rd, e0);
            n_listed1 += (uint32_t)__popcll(m1);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (n_listed1 > (uint32_t)SCAN_LIST1 - 64u) flush_list1();
        }
    }
    if (n_listed) flush_list();
    if (n_listed1) flush_list1();
    flush_counts(w_last - wb);
    __syncthreads();
    if (threadIdx.x < 2) A.read_count[threadIdx.x * SCAN_MAX_WGS + blockIdx.x] = l_seg[threadIdx.x];
}

// ================================================================== k_indel_script
// The reads k_indel_scan listed (a mate with at least one event), 64 per wavefront block, one lane per read:
// introduce_indels + adjust_seq_length (__init__.py:158-228, 114-156) from the read's event list, exactly, as the token
// transducer of k_indel_fixup (below) -- the list prefix [0, n) is final when step n starts; the not-yet-visited suffix is
// (stack of freshly inserted letters, LIFO) ++ E(k), E(k+1), ... -- but instead of rebuilding the read it leaves the EDIT
// SCRIPT k_main<.., INDEL> builds the read from (SC_* above).  Round 4: round 3 rewrote the letters behind k_main
// (a pass behind k_main: every rewritten piece crossed HBM twice more, 3.1 GB per 5 M pairs of BASELINE configs[4], and the
// substitutions k_main had applied were listed and re-applied to the letters standing there afterwards); now the walk runs in
// FRONT of k_main, which takes a piece's window from the shifted genome position, and nothing is written twice.
//   * descriptor, events (registers), the read's window of the 2-bit genome (LDS), the walk over the steps with an event:
//     runs (from step s on, token = step + shift) and explicit letters (event steps, stack drains), as 2-bit codes;
//   * per piece of 8 positions one byte: the run's shift, or -- a piece with an explicit letter or a run boundary -- the index
//     of its 8 letters, merged here from the runs it touches and the explicit letters, in k_main's window format;
//   * the rows go out as whole 64-byte sectors (four 16-byte rows of a tile's group of 8 iterations).
// Reads this cannot take -- records with IUPAC / lower-case letters, a window that leaves the record (the reference pads with
// 'A' there), more explicit letters or stacked insertions than the script holds, more than SC_ROW_CODES explicit pieces in one
// row, read lengths beyond AP_MAX_PITCH, insertion letters outside A/C/G/T -- join the irregular pairs and the reads with more
// than EV_K events in k_indel_fixup's list (one wavefront per read, exact, slow) and have their event counter cleared:
// ev_count[read] != 0 <=> the read has a valid script.
constexpr int SC_WAVES = 8;             // wavefronts per workgroup
constexpr int SC_WAVES1 = 16;           // ... of the SINGLE launch (a lane's record is its window: 8 wavefronts / SIMD hide the loads' latency)
constexpr int SC_WGS_PER_CU = 2;        // workgroups per CU (the lanes' records: 2 x 8 x 8.4 KB of LDS for read lengths up to 168)
#ifndef ISS_SCRIPT_OCC
#define ISS_SCRIPT_OCC 4                // wavefronts per SIMD the register budget is cut for
#endif
constexpr int AP_RUNS = EV_K + 1;       // runs of a read: the initial one + one per step with an event
constexpr int AP_LETTERS = 12;          // explicit letters kept per read (16 bits each: position << 2 | code)
constexpr int AP_ITEMS = 128;           // batch calls: the table of up to this many work items is cached in LDS (beyond: global loads)
constexpr int AP_MAX_PITCH = 384;       // longer reads: k_indel_fixup
constexpr int AP_STACK = 8;             // inserted letters waiting to surface (a 64-bit register)

__host__ __device__ inline int ap_pitch(int pitch) { return pitch < AP_MAX_PITCH ? pitch : AP_MAX_PITCH; }
__host__ __device__ inline int ap_win(int pitch) { return ap_pitch(pitch) + 8; }  // template positions a read of <= EV_K events reaches
__host__ __device__ inline int ap_ww(int pitch) { return (ap_win(pitch) + 15) / 16 + 1; }  // window words (16 bases each)
// a lane's LDS record: [runs AP_RUNS][window words][letters AP_LETTERS / 2][pad][header SC_HDR, at the END of the record] (odd:
// no bank conflicts between lanes; the window's neighbours are the lane's own words: the piece builder may look one word
// past either end, masked out afterwards)
constexpr int SC_HDR = 5;               // read, geometry, piece mask (2), explicit pieces of the block in front of this read's
__host__ __device__ inline int sc_rec_words(int pitch) { return (AP_RUNS + ap_ww(pitch) + AP_LETTERS / 2 + 1 + SC_HDR) | 1; }
__host__ __device__ inline size_t ap_items_bytes() { return (AP_ITEMS + 2) * 4 + AP_ITEMS * sizeof(BatchItem); }
__host__ __device__ inline size_t ap_seg_bytes() { return (size_t)(2 * SCAN_MAX_WGS + 4) * 4; }  // the read list's segments: 64-read blocks in front of each, lengths
// [ins_letter 2*RL*4 u8, padded][segments][item_first AP_ITEMS+2 u32][items AP_ITEMS][per wave: 64 records]
__host__ __device__ inline size_t ap_tab_bytes(int RL) { return (((size_t)2 * RL * 4 + 15) & ~(size_t)15) + ap_seg_bytes() + ap_items_bytes(); }
__host__ __device__ inline int sc_rec_words1(int pitch) { return (1 + ap_ww(pitch) + 2) | 1; }  // SINGLE: [pad][window][pad]
__host__ __device__ inline size_t sc_wave_bytes(int pitch, bool single) { return (size_t)64 * (single ? sc_rec_words1(pitch) : sc_rec_words(pitch)) * 4; }
__host__ __device__ inline size_t script_lds_bytes(int RL, int pitch, bool single) {
    return ap_tab_bytes(RL) + (size_t)(single ? SC_WAVES1 : SC_WAVES) * sc_wave_bytes(pitch, single);
}

// WWM: window words a lane keeps in registers for the NEXT block's read (>= ap_ww(pitch): 12 for read lengths up to 168, else 26)
// SINGLE: the launch takes the reads with ONE step with an event (RunArgs::read_list1; more than half of the listed reads of
// BASELINE configs[4]) -- no walk, no lists: the event's effect is a closed form (at most five explicit letters in at most two
// pieces, one run behind them) and the rows follow from three numbers; the other launch takes read_list with the general code.
template <bool STORE_MUT, int WWM, bool SINGLE>
__global__ __launch_bounds__(SINGLE ? 64 * SC_WAVES1 : 64 * SC_WAVES, SINGLE ? 8 : ISS_SCRIPT_OCC) void k_indel_script(DevModel M, DevGenome g, RunArgs A,
                                                                               const PairDesc *__restrict__ desc, uint64_t *stats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ap_lds[];
    const int RL = M.RL, pitch = M.pitch, S = M.S, WW = ap_ww(pitch), RW = SINGLE ? sc_rec_words1(pitch) : sc_rec_words(pitch), WIN = ap_win(pitch);
    uint8_t *insl = reinterpret_cast<uint8_t *>(ap_lds);                  // [2][RL][4]
    uint32_t *ifirst = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ap_lds) + ap_tab_bytes(RL) - ap_items_bytes());  // (a call holds < 2^31 pairs)
    BatchItem *l_items = reinterpret_cast<BatchItem *>(ifirst + AP_ITEMS + 2);
    // the read list's segments (one per workgroup of k_indel_scan): seg_blk[s] = 64-read blocks in front of segment s
    uint32_t *seg_blk = ifirst - (2 * SCAN_MAX_WGS + 4), *seg_len = seg_blk + SCAN_MAX_WGS + 2;
    const uint32_t n_waves = blockDim.x >> 6;
    if (threadIdx.x < 64u) {
        constexpr uint32_t PER = SCAN_MAX_WGS / 64;
        uint32_t len[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t sg = threadIdx.x * PER + k;
            len[k] = sg < A.scan_wgs ? A.read_count[(SINGLE ? SCAN_MAX_WGS : 0) + sg] : 0u;
            sum += (len[k] + 63u) / 64u;
        }
        uint32_t inc = sum;
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, dlt);
            if ((int)threadIdx.x >= dlt) inc += t;
        }
        uint32_t run = inc - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            seg_blk[threadIdx.x * PER + k] = run;
            seg_len[threadIdx.x * PER + k] = len[k];
            run += (len[k] + 63u) / 64u;
        }
        if (threadIdx.x == 63u) seg_blk[SCAN_MAX_WGS] = run;
    }
    for (int i = threadIdx.x; i < 2 * RL * 4; i += blockDim.x) insl[i] = M.ins_letter[i];
    const bool items_cached = A.items && A.n_items <= AP_ITEMS;
    if (items_cached) {
        for (int i = threadIdx.x; i <= A.n_items; i += blockDim.x) ifirst[i] = (uint32_t)A.item_first[i];
        for (int i = threadIdx.x; i < A.n_items; i += blockDim.x) l_items[i] = A.items[i];
    }
    __syncthreads();
    const uint32_t n_blocks = seg_blk[SCAN_MAX_WGS];
    if (blockIdx.x * n_waves >= n_blocks) return;  // whole workgroup idle (uniform)
    const uint32_t seg_stride = scan_per_wg(2u * (uint32_t)A.n_pairs, A.scan_wgs);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t *wave0 = ap_lds + ap_tab_bytes(RL) / 4 + (size_t)wv * (sc_wave_bytes(pitch, SINGLE) / 4);
    // (then every listed read is k_indel_fixup's)
    const bool all_to_fixup = pitch > AP_MAX_PITCH || !M.ins_plain;
    const uint32_t inv_ts = (65536u + (uint32_t)M.TS - 1u) / (uint32_t)M.TS;  // s / TS == (s * inv_ts) >> 16 for the s < 128 of a read
    const char *const packed_b = reinterpret_cast<const char *>(g.packed - 1);  // (the leading padding word: offsets >= 0)
    uint32_t n_scripted = 0;
    const uint32_t NO_READ = 0xffffffffu, stride = gridDim.x * n_waves;
    const uint32_t blk0 = blockIdx.x * n_waves + wv;
    auto list_entry = [&](uint32_t b) {  // lane `lane` of block b of the segmented list
        if (b >= n_blocks) return make_uint4(NO_READ, 0u, 0u, 0u);
        uint32_t lo = 0, hi = SCAN_MAX_WGS;  // the segment: seg_blk[lo] <= b < seg_blk[lo + 1]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (seg_blk[mid] <= b) lo = mid; else hi = mid;
        }
        const uint32_t off = (b - seg_blk[lo]) * 64u + lane;
        if (off >= seg_len[lo]) return make_uint4(NO_READ, 0u, 0u, 0u);
        if (SINGLE) { const uint2 v = A.read_list1[(size_t)lo * seg_stride + off]; return make_uint4(v.x, v.y, 0u, 1u); }
        return A.read_list[(size_t)lo * seg_stride + off];
    };
    // software pipeline of the block loop, three deep: the list entry is requested three blocks ahead, what hangs on the read
    // number (the descriptor's two coordinates, the events of a read with more than two) two blocks ahead, the window of the
    // packed genome -- it hangs on the descriptor -- one block ahead: every load of the chain has a whole block to arrive
    struct Coords { int64_t fs, re; };
    auto coords_of = [&](const uint4 &rec) {
        const PairDesc dd = desc[(rec.x == NO_READ ? 0u : rec.x) >> 1];
        return Coords{desc_fs(dd), desc_re(dd)};
    };
    auto events_of = [&](const uint4 &rec, uint4 &ea, uint4 &eb) {
        ea = make_uint4(0u, 0u, 0u, 0u); eb = ea;
        if (!SINGLE && rec.x != NO_READ && rec.w > 2u) {
            ea = reinterpret_cast<const uint4 *>(A.ev_list + (size_t)rec.x * EV_K)[0];
            eb = reinterpret_cast<const uint4 *>(A.ev_list + (size_t)rec.x * EV_K)[1];
        }
    };
    uint32_t wn[WWM];
    auto request_window = [&](const uint4 &rec, const Coords &c) {
        const int64_t wl = (rec.x != NO_READ && (rec.x & 1u)) ? c.re - WIN : c.fs;
        // (16 bytes per load instruction, 4-byte aligned: a lane's window is 1-2 cache lines and every instruction brings 64
        //  lanes' lines through the L1 -- twelve single words per lane were measured at twice the kernel's time)
        const uint4 *src = reinterpret_cast<const uint4 *>(packed_b + (size_t)(((wl >> 4) + 1) << 2));
#pragma unroll
        for (int k = 0; k < WWM; k += 4) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (k < WW) v = src[k >> 2];
            wn[k] = v.x;
            if (k + 1 < WWM) wn[k + 1] = v.y;
            if (k + 2 < WWM) wn[k + 2] = v.z;
            if (k + 3 < WWM) wn[k + 3] = v.w;
        }
    };
    uint4 rec0 = list_entry(blk0), rec1 = list_entry(blk0 + stride), rec2 = list_entry(blk0 + 2u * stride);
    Coords c0 = coords_of(rec0), c1 = coords_of(rec1);
    uint4 ea0, eb0, ea1, eb1;
    events_of(rec0, ea0, eb0);
    events_of(rec1, ea1, eb1);
    request_window(rec0, c0);
    for (uint32_t blk = blk0; blk < n_blocks; blk += stride) {
        uint32_t wc[WWM];  // this block's windows
#pragma unroll
        for (int k = 0; k < WWM; ++k) wc[k] = wn[k];
        const bool listed = rec0.x != NO_READ;
        const uint32_t rd = listed ? rec0.x : 0u;
        const uint32_t pair = rd >> 1;
        const int o = (int)(rd & 1u);
        const int64_t d_fs = c0.fs, d_re = c0.re;
        const uint32_t cnt = listed ? rec0.w : 0u;
        uint32_t e[EV_K];
        e[0] = ea0.x; e[1] = ea0.y; e[2] = ea0.z; e[3] = ea0.w; e[4] = eb0.x; e[5] = eb0.y; e[6] = eb0.z; e[7] = eb0.w;
        if (cnt <= 2u) { e[0] = rec0.y; e[1] = rec0.z; }
        {   // the pipeline moves on: window of the next block, coordinates / events of the one after, list entry of the third
            rec0 = rec1; c0 = c1; ea0 = ea1; eb0 = eb1;
            request_window(rec0, c0);
            rec1 = rec2;
            c1 = coords_of(rec1);
            events_of(rec1, ea1, eb1);
            rec2 = list_entry(blk + 3u * stride);
        }
        // (a listed read is never k_indel_fixup's already: the scan lists neither the mates of irregular pairs nor reads
        //  with more events than a list holds)
        bool ok = cnt > 0u && cnt <= (uint32_t)EV_K;
        // the record of the pair: the launch's genome, or its slice of the arena (batch calls); descriptors hold arena coordinates
        int64_t rec_lo = 0, rec_hi = g.L;
        bool plain = !g.has_exceptions;
        if (A.items) {
            BatchItem it;
            if (items_cached) {
                const uint32_t p = (uint32_t)(A.pair_base + pair);
                int lo = 0, hi = A.n_items;  // largest k with item_first[k] <= p
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ifirst[mid] <= p) lo = mid; else hi = mid; }
                it = l_items[lo];
            } else {
                it = A.items[batch_item_of(A, A.pair_base + pair)];
            }
            rec_lo = it.off; rec_hi = it.off + it.L; plain = !it.has_exceptions;
        }
        // window: tokens 0 .. WIN - 1 = genome positions fs .. fs + WIN - 1 (forward) / re - 1 down to re - WIN (reverse).  Beyond
        // the record's ends the reference pads with 'A' (__init__.py:141-155), which k_main's windows cannot: such reads -- a few
        // in 10^5 -- are the fix-up kernel's
        const int64_t w_lo = o ? d_re - WIN : d_fs;
        auto to_fixup = [&]() {
            if (!(atomicOr(&A.flags[pair], 1u << o) & (1u << o))) A.fix_list[atomicAdd(A.fix_count, 1u)] = rd;
            A.ev_count[rd] = 0u;  // no script
        };
        if (ok && !(plain && !all_to_fixup && w_lo >= rec_lo && w_lo + WIN <= rec_hi)) { to_fixup(); ok = false; }
        uint32_t *R = wave0 + lane * RW, *runs = R, *win = R + (SINGLE ? 1 : AP_RUNS);
        uint16_t *let = reinterpret_cast<uint16_t *>(win + WW);
        const int64_t wpos = (w_lo >> 4) << 4;  // genome position of bit 0 of the window
        const int t0pos = (int)((o ? d_re - 1 : d_fs) - wpos);  // window position of token 0 (token t: t0pos +/- t)
        if (ok) {
#pragma unroll
            for (int k = 0; k < WWM; ++k) if (k < WW) win[k] = wc[k];  // (requested a block ago)
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // the walk over the steps with an event (k_indel_scan lists them in step order, one entry per step)
        auto code_at = [&](int tk) {  // 2-bit code of template token tk in read direction
            const int b = 2 * (o ? t0pos - tk : t0pos + tk);
            const uint32_t c = (win[b >> 5] >> (b & 31)) & 3u;
            return o ? c ^ 1u : c;  // complement: code ^ 1
        };
        if (SINGLE) {
            // ---- ONE step n with an event (mask m8), no walk: token n is the template's; the letters the step inserts wait on the
            //      stack; a deletion lets the next token slide in (the last inserted letter, else template token n + 1); the
            //      steps behind drain the stack; from step n + D + 1 on the read is the template again, shifted by sh1
            const int n = (int)(e[0] >> 8);
            const uint32_t m8 = e[0] & 0xffu;
            uint32_t stk = 0, rem = 0, Fn = 0;
            int D = 0, sh1 = 0;
            if (ok) {
                const uint32_t ch = code_at(n);
                MutRecord row;  // --store_mutations row being built
                row.pair = (int32_t)(A.pair_base + pair); row.mate = (int8_t)o; row.quality = -1;
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if ((m8 >> x) & 1u) {
                        const int letter = insl[((size_t)o * RL + n) * 4 + x];
                        stk |= (uint32_t)base_index(letter) << (2u * rem);
                        ++rem;
                        if (STORE_MUT) {  // ref = mutable_seq[position], alt = ref + letter (__init__.py:197-207)
                            row.type = (int8_t)(1 | (x << 2) | 32); row.position = (int16_t)n;
                            row.ref = code_to_ascii(ch); row.alt = (uint8_t)letter;
                            mut_emit1(A, row);
                        }
                    }
                int k = n + 1;
                Fn = ch;
                if ((m8 >> (4 + ch)) & 1u) {  // deleted: the next token slides in
                    if (rem > 0u) { --rem; Fn = (stk >> (2u * rem)) & 3u; }
                    else { Fn = code_at(n + 1); k = n + 2; }
                    if (STORE_MUT) {  // ref = mutable_seq[position] after the pop (__init__.py:211-221; it exists: n + 1 < RL)
                        row.type = (int8_t)(2 | (4 << 2) | 32); row.position = (int16_t)n;
                        row.ref = code_to_ascii(Fn); row.alt = '.';
                        mut_emit1(A, row);
                    }
                }
                D = min((int)rem, RL - 1 - n);  // steps n + 1 .. n + D drain the stack
                sh1 = k - (n + D + 1);
            }
            // explicit letters: position n holds Fn, position n + i (1 <= i <= D) the letter (stk >> 2 (rem - i)) & 3; they lie in
            // pieces p0 = n >> 3 and p1 = (n + D) >> 3 (the same or the next one)
            const int p0 = n >> 3, p1 = (n + D) >> 3;
            auto raw16 = [&](int t) {  // raw window bits of tokens t .. t + 7, the format of k_main's fb / rbr
                const int b = 2 * (o ? t0pos - t - 7 : t0pos + t);
                return funnel_r(win[b >> 5], win[(b >> 5) + 1], (uint32_t)b) & 0xffffu;
            };
            auto piece_codes = [&](int pc) {
                const int j0 = 8 * pc;
                const int lo_c = min(max(n - j0, 0), 8), hi_c = min(max(n + D + 1 - j0, 0), 8);  // [0, lo_c) template, [hi_c, 8) template shifted
                const uint32_t w0 = raw16(j0), w1 = raw16(j0 + sh1);
                const uint32_t m_lo = o ? 0xffffu & ~(0xffffu >> (2 * lo_c)) : (1u << (2 * lo_c)) - 1u;
                const uint32_t m_hi = o ? 0xffffu >> (2 * hi_c) : 0xffffu & ~((1u << (2 * hi_c)) - 1u);
                uint32_t nw = (w0 & m_lo) | (w1 & m_hi);
#pragma unroll
                for (int i = 0; i <= 4; ++i) {
                    const int c = n + i - j0;
                    if (i <= D && c >= 0 && c < 8) {
                        const uint32_t code = i == 0 ? Fn : (stk >> (2u * (rem - (uint32_t)i))) & 3u;
                        nw |= (o ? code ^ 1u : code) << (2u * (uint32_t)(o ? 7 - c : c));
                    }
                }
                return nw;
            };
            uint32_t code0 = 0, code1 = 0;
            if (ok) code0 = piece_codes(p0);
            if (__ballot(ok && p1 != p0)) { if (ok && p1 != p0) code1 = piece_codes(p1); }
            // rows: per tile and group of 8 iterations, row j4 holds the pieces s_first + 4 it + j4: those below p0 idle, p0 / p1
            // explicit (a row holds at most one of two neighbouring pieces), those above p1 shifted by sh1
            if (ok) {
                uint8_t *out = A.script + (size_t)rd * (size_t)(uint32_t)M.sc_stride;
                const int n_grp = M.n_tiles * M.sc_gpt;
                const uint64_t fill = (uint64_t)(((uint32_t)(64 + sh1) ^ 0x40u) & 0xffu) * 0x0101010101010101ull;
                int tile = 0, g_in_tile = 0;
                for (int gq = 0; gq < n_grp; ++gq) {
                    const int s_first = tile * M.TS + 32 * g_in_tile;
                    uint4 rowv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int a = p0 - s_first - j, b = p1 - s_first - j;
                        const int it_lo = a <= 0 ? 0 : min(8, (a + 3) >> 2), it_hi = b < 0 ? 0 : min(8, (b >> 2) + 1);
                        uint64_t r = ((uint64_t)SC_IDLE << 32) | SC_IDLE;
                        if (it_hi < 8) r ^= fill & (~(uint64_t)0 << (8 * it_hi));
                        uint32_t code = 0;
                        if (it_hi > it_lo) {
                            r ^= (uint64_t)(0x80u ^ 0x40u) << (8 * it_lo);
                            code = s_first + 4 * it_lo + j == p0 ? code0 : code1;
                        }
                        rowv[j] = make_uint4((uint32_t)r, (uint32_t)(r >> 32), code, 0u);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) reinterpret_cast<uint4 *>(out + (size_t)gq * 64)[j] = rowv[j];  // (a whole 64-byte sector)
                    if (++g_in_tile == M.sc_gpt) { g_in_tile = 0; ++tile; }
                }
                ++n_scripted;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        uint64_t evmask = 0;  // pieces with an explicit letter
        uint32_t n_runs = 1, n_let = 0;
        bool bad = false;
        // (--store_mutations: a dry walk first -- a read whose script overflows goes to k_indel_fixup, which writes ALL of its
        //  rows: none may come from here)
        auto walk = [&](bool emit_rows) {
            uint32_t ev[EV_K];
#pragma unroll
            for (int z = 0; z < EV_K; ++z) ev[z] = e[z];
            uint64_t stk = 0;  // inserted letters waiting to surface (LIFO, a byte each: their 2-bit codes)
            int sp = 0, k = 0, last = -1;
            evmask = 0; n_runs = 1; n_let = 0; bad = false;
            if (ok) runs[0] = 0u;
            MutRecord row;  // --store_mutations row being built
            row.pair = (int32_t)(A.pair_base + pair); row.mate = (int8_t)o; row.quality = -1;
            auto put_letter = [&](int pos, uint32_t code) {
                if (n_let < (uint32_t)AP_LETTERS) let[n_let] = (uint16_t)(((uint32_t)pos << 2) | code); else bad = true;
                ++n_let;
                evmask |= (uint64_t)1 << (pos >> 3);
            };
            for (uint32_t left = ok ? cnt : 0u; left > 0u; --left) {
                const int n = (int)(ev[0] >> 8);
                const uint32_t m8 = ev[0] & 0xffu;
#pragma unroll
                for (int z = 0; z + 1 < EV_K; ++z) ev[z] = ev[z + 1];
                const int next_n = left > 1u ? (int)(ev[0] >> 8) : RL;
                k += n - (last + 1);  // the settled run covers steps last+1 .. n-1 from the template
                uint32_t ch;  // 2-bit code (the record is plain A/C/G/T and so are the insertion letters: no ambiguous token, :190-192)
                bool visit;  // tok < len(template): else n >= len(seq), IndexError swallowed (:223): emitted unvisited
                if (sp > 0) { ch = (uint32_t)(stk & 0xffu); stk >>= 8; --sp; visit = true; }
                else { visit = k < RL; ch = code_at(k); ++k; }
                if (visit) {
                    for (int x = 0; x < 4; ++x)
                        if ((m8 >> x) & 1u) {
                            const int letter = insl[((size_t)o * RL + n) * 4 + x];
                            if (sp < AP_STACK) stk = (stk << 8) | (uint64_t)(uint32_t)base_index(letter); else bad = true;
                            ++sp;
                            if (STORE_MUT && emit_rows) {  // ref = mutable_seq[position], alt = ref + letter (__init__.py:197-207)
                                row.type = (int8_t)(1 | (x << 2) | 32); row.position = (int16_t)n;
                                row.ref = code_to_ascii(ch); row.alt = (uint8_t)letter;
                                mut_emit1(A, row);
                            }
                        }
                    if (bad) break;
                    if ((m8 >> (4 + ch)) & 1u) {  // deleted: the next token slides in
                        const bool exists = sp > 0 || k < RL;  // else mutable_seq[position] raises IndexError: no row
                        if (sp > 0) { ch = (uint32_t)(stk & 0xffu); stk >>= 8; --sp; }
                        else { ch = code_at(k); ++k; }
                        if (STORE_MUT && emit_rows && exists) {  // ref = mutable_seq[position] after the pop (__init__.py:211-221)
                            row.type = (int8_t)(2 | (4 << 2) | 32); row.position = (int16_t)n;
                            row.ref = code_to_ascii(ch); row.alt = '.';
                            mut_emit1(A, row);
                        }
                    }
                }
                put_letter(n, ch);
                last = n;
                // steps after n drain the insertion stack until it is empty or the next step with an event
                while (sp > 0 && last + 1 < RL && last + 1 != next_n) {
                    ++last;
                    put_letter(last, (uint32_t)(stk & 0xffu));
                    stk >>= 8; --sp;
                }
                // from step last + 1 on: token = step + (k - (last + 1)) until the next step with an event
                if (last + 1 < pitch) { runs[n_runs] = (uint32_t)(last + 1) | ((uint32_t)(k - (last + 1)) << 16); ++n_runs; }
            }
        };
        // more than SC_ROW_CODES explicit pieces in one 16-byte row of a group (the rows loop below finds the same overflow, but
        // only after walk(true) has emitted the read's rows: k_indel_fixup would then write them a second time)
        auto rows_over = [&]() {
            bool ov = false;
            int tile = 0, g_in_tile = 0;
            for (int gq = 0; gq < M.n_tiles * M.sc_gpt; ++gq) {
                const int s_first = tile * M.TS + 32 * g_in_tile, s_end = min(S, (tile + 1) * M.TS);
                if (++g_in_tile == M.sc_gpt) { g_in_tile = 0; ++tile; }
                if (s_first >= s_end) continue;
                const uint64_t below_end = ((uint64_t)1 << s_end) - 1u;  // (S <= AP_MAX_PITCH / 8 = 48 pieces)
                for (int j = 0; j < 4; ++j)
                    if (__popcll(evmask & below_end & ((uint64_t)0x11111111u << (s_first + j))) > SC_ROW_CODES) ov = true;
            }
            return ov;
        };
        if (STORE_MUT) { walk(false); if (ok && (bad || rows_over())) { to_fixup(); ok = false; } }
        walk(true);
        if (ok && bad) { to_fixup(); ok = false; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // ---- the rows, one lane per read: per tile and group of 8 iterations one byte per piece, in ascending order (the run
        //      pointer only moves forward) -- the run's shift, or the place of the piece's explicit codes in its row.  The four
        //      rows of a group leave as one whole 64-byte sector, the code halves still empty.
        if (__ballot(ok)) {
            int ri = 0, sh = 0, nstart = n_runs > 1 ? (int)(runs[1] & 0xffffu) : 0x7fffffff;
            uint8_t *out = A.script + (size_t)rd * (size_t)(uint32_t)M.sc_stride;
            const int n_grp = M.n_tiles * M.sc_gpt;
            bool over = false;
            int tile = 0, g_in_tile = 0;
            for (int gq = 0; gq < n_grp; ++gq) {
                const int s_first = tile * M.TS + 32 * g_in_tile, s_end = min(S, (tile + 1) * M.TS);
                if (++g_in_tile == M.sc_gpt) { g_in_tile = 0; ++tile; }
                uint32_t rx[4], ry[4], ne[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { rx[j] = SC_IDLE; ry[j] = SC_IDLE; ne[j] = 0; }
                for (int it8 = 0; it8 < 8; ++it8) {
                    if (!__ballot(ok && s_first + 4 * it8 < s_end)) break;  // (uniform)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int s = s_first + 4 * it8 + j, j0 = 8 * s;
                        if (!ok || s >= s_end) continue;
                        while (nstart <= j0) {  // the run position j0 lies in
                            ++ri;
                            sh = (int)runs[ri] >> 16;
                            nstart = ri + 1 < (int)n_runs ? (int)(runs[ri + 1] & 0xffffu) : 0x7fffffff;
                        }
                        uint32_t byte = (uint32_t)(64 + sh);
                        if ((evmask >> s) & 1u) {
                            if (ne[j] >= (uint32_t)SC_ROW_CODES) over = true;
                            byte = 128u + 16u * (ne[j] & 3u);
                            ++ne[j];
                        }
                        const uint32_t ins = (byte ^ 0x40u) << (8u * (uint32_t)(it8 & 3));  // (the rows start as SC_IDLE)
                        if (it8 < 4) rx[j] ^= ins; else ry[j] ^= ins;
                    }
                }
                if (ok)
#pragma unroll
                    for (int j = 0; j < 4; ++j) reinterpret_cast<uint4 *>(out + (size_t)gq * 64)[j] = make_uint4(rx[j], ry[j], 0u, 0u);
            }
            if (ok && over) { to_fixup(); ok = false; }
        }
        // ---- the explicit pieces, DENSELY: a read has one or two of them among its pitch / 8 pieces, so a lane-per-read loop
        //      over the pieces would run the (long) merge below with a tenth of its lanes.  The block's explicit pieces are
        //      numbered through (prefix sum of the reads' counts) and dealt out 64 at a time, one per lane: the lane finds the
        //      piece's read (bisection of the prefix sums), builds the piece's 8 codes from that read's record in LDS -- the run
        //      the piece starts in, the runs that start inside it merged from their first position on, the explicit letters --
        //      and stores them behind the row's bytes (two bytes into the sector the read's own lane has just written: the
        //      line is still in the L2).
        {
            const uint32_t n_exp = ok ? (uint32_t)__popcll(evmask) : 0u;
            uint32_t incl = n_exp;
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, dlt);
                if ((int)lane >= dlt) incl += t;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            R[RW - SC_HDR + 0] = rd;
            R[RW - SC_HDR + 1] = (uint32_t)t0pos | ((uint32_t)o << 16) | (n_runs << 20) | (n_let << 24);
            R[RW - SC_HDR + 2] = (uint32_t)evmask;
            R[RW - SC_HDR + 3] = (uint32_t)(evmask >> 32);
            R[RW - SC_HDR + 4] = incl - n_exp;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t item = base + lane;
                if (item < total) {
                    uint32_t lo = 0, hi = 64;  // the read of this piece: the largest l with (pieces in front of l) <= item
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (wave0[mid * RW + RW - SC_HDR + 4] <= item) lo = mid; else hi = mid;
                    }
                    const uint32_t *Ro = wave0 + lo * RW, *o_runs = Ro, *o_win = Ro + AP_RUNS, *H = Ro + RW - SC_HDR;
                    const uint16_t *o_let = reinterpret_cast<const uint16_t *>(o_win + WW);
                    const uint64_t em = (uint64_t)H[2] | ((uint64_t)H[3] << 32);
                    uint64_t m = em;
                    for (uint32_t i = item - H[4]; i > 0u; --i) m &= m - 1;
                    const int s = __ffsll((unsigned long long)m) - 1, j0 = 8 * s;
                    const uint32_t h1 = H[1];
                    const int p_t0 = (int)(h1 & 0xffffu), p_o = (int)((h1 >> 16) & 1u), p_runs = (int)((h1 >> 20) & 15u);
                    const uint32_t p_let = h1 >> 24;
                    // raw window bits of tokens t .. t + 7, the format of k_main's fb / rbr (reverse: genome orientation, uncomplemented)
                    auto raw16 = [&](int t) {
                        const int b = 2 * (p_o ? p_t0 - t - 7 : p_t0 + t);
                        return funnel_r(o_win[b >> 5], o_win[(b >> 5) + 1], (uint32_t)b) & 0xffffu;
                    };
                    int k0 = 0;
                    while (k0 + 1 < p_runs && (int)(o_runs[k0 + 1] & 0xffffu) <= j0) ++k0;
                    uint32_t nw = raw16(j0 + ((int)o_runs[k0] >> 16));
                    for (int k = k0 + 1; k < p_runs && (int)(o_runs[k] & 0xffffu) < j0 + 8; ++k) {
                        const int c = (int)(o_runs[k] & 0xffffu) - j0;
                        const uint32_t v = raw16(j0 + ((int)o_runs[k] >> 16));
                        const uint32_t mk = p_o ? 0xffffu >> (2 * c) : (0xffffu << (2 * c)) & 0xffffu;  // read positions c .. 7
                        nw = (v & mk) | (nw & ~mk);
                    }
                    for (uint32_t z = 0; z < p_let; ++z) {
                        const int c = (int)(o_let[z] >> 2) - j0;
                        if (c >= 0 && c < 8) {
                            const uint32_t code = o_let[z] & 3u, at = 2u * (uint32_t)(p_o ? 7 - c : c);
                            nw = (nw & ~(3u << at)) | ((p_o ? code ^ 1u : code) << at);
                        }
                    }
                    // its place: tile, group of 8 iterations, row j4, and the number of explicit pieces of the row in front of it
                    const int tile = (int)(((uint32_t)s * inv_ts) >> 16), it = (s - tile * M.TS) >> 2, grp = it >> 3, it8 = it & 7;
                    const uint64_t row_before = (0x1111111111111111ull << (s - 4 * it8)) & (((uint64_t)1 << s) - 1u) & em;
                    uint8_t *dst = A.script + (size_t)H[0] * (size_t)(uint32_t)M.sc_stride +
                                   (size_t)((((tile * M.sc_gpt + grp) * 4 + (s & 3)) * 16) + 8 + 2 * __popcll(row_before));
                    *reinterpret_cast<uint16_t *>(dst) = (uint16_t)nw;
                }
            }
        }
        if (ok) ++n_scripted;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned long long any = __ballot(n_scripted != 0u);
    if (any) {
        uint32_t tot = n_scripted;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
        if (lane == 0) atomicAdd((unsigned long long *)stats + 1, (unsigned long long)tot);
    }
}

// ================================================================== k_indel_fixup
// One wavefront per flagged read.  Exact introduce_indels + adjust_seq_length as a token transducer:
// the list prefix [0, n) is final when step n starts; the not-yet-visited suffix is
// (stack of freshly inserted letters, LIFO) ++ E(k), E(k+1), ...   (DESIGN.md "indel transducer").
//   phase 1: event mask per loop step n, independent of the token (bits 0-3 insertion of letter slot x fires,
//            bits 4-7 deletion fires if the token is base b) from the read's event process (indel_events, run
//            wave-uniformly); ballots of the steps with a non-empty mask; the template E(0 .. RL+63) is staged in LDS.
//   phase 2 (wave-uniform walk over the ACTIVE steps only): explicit map[] entries for active steps
//            and for the steps that drain the insertion stack; "from step n0 on, source index =
//            k0 + (n - n0)" records for everything in between.
//   phase 3 (all lanes): token -> base -> mut_sequence -> store.
constexpr int16_t FIX_NONE = 0x7fff;
constexpr int FIX_MAX_RL = 1024;  // read_length limit (checked at model upload)
constexpr int FIX_WAVES = 4;      // wavefronts (reads) per workgroup

// dynamic LDS of k_indel_fixup, in bytes: [fix table 2*RL*8 u32][mut8 64 u32][per wave: see below]
__host__ __device__ inline int fix_rlp(int RL) { return (RL + 63) & ~63; }
__host__ __device__ inline size_t fix_wave_bytes(int RL) {
    const size_t rlp = (size_t)fix_rlp(RL);
    return rlp * 4 /* ev, stk, qual, (pad) */ + (rlp + 64) /* tmpl */ + 3 * 2 * (rlp + 64) /* map, rec_n0, rec_k0 */ +
           2 * rlp /* dqm: the mate's error-test digits, one Philox block per lane */ +
           (FIX_MAX_RL / 64) * 8 /* act: the steps with an event, one 64-bit mask per chunk (wave-uniform, out of the registers) */;
}
// (+ the indel event tables [2][ns] of 8 + 2 bytes when they fit beside the waves' rows: the binary search of a draw is a chain
//  of dependent reads, microseconds each from global memory behind the chip's write stream)
__host__ __device__ inline size_t fix_ev_bytes(int RL) { const size_t ns = (size_t)5 * (RL - 1); return 2 * ns * 8 + ((2 * ns * 2 + 7) & ~(size_t)7); }
__host__ __device__ inline bool fix_ev_in_lds(int RL) { return 64 * 4 + FIX_WAVES * fix_wave_bytes(RL) + fix_ev_bytes(RL) <= (size_t)150 * 1024; }
__host__ __device__ inline size_t fix_lds_bytes(int RL) {
    return 64 * 4 + FIX_WAVES * fix_wave_bytes(RL) + (fix_ev_in_lds(RL) ? fix_ev_bytes(RL) : 0);
}

__global__ __launch_bounds__(64 * FIX_WAVES, 5) void k_indel_fixup(DevModel M, DevGenome g, RunArgs A,
                                                                const PairDesc *__restrict__ desc,
                                                                const uint32_t *__restrict__ fix_list,
                                                                const uint32_t *__restrict__ fix_count,
                                                                uint64_t *stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fix_lds[];
    const int RL = M.RL;
    const int rlp = fix_rlp(RL);
    uint32_t *mut8 = reinterpret_cast<uint32_t *>(fix_lds);           // [64] leading 8 bits of the substitution-test thresholds
    const uint32_t n_fix = *fix_count;
    if (blockIdx.x * FIX_WAVES >= n_fix) return;                       // whole workgroup idle (uniform)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long *)stats, (unsigned long long)n_fix);
    for (int i = threadIdx.x; i <= M.n_q; i += blockDim.x) mut8[i] = (uint32_t)(M.mut_thr[i] >> 45);
    const uint64_t *ev_S = M.ev_S;
    const uint16_t *ev_E = M.ev_E;
    if (fix_ev_in_lds(RL)) {
        uint64_t *s_S = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(mut8 + 64) + (size_t)FIX_WAVES * fix_wave_bytes(RL));
        uint16_t *s_E = reinterpret_cast<uint16_t *>(s_S + 2 * M.ev_ns);
        for (int i = threadIdx.x; i < 2 * M.ev_ns; i += blockDim.x) { s_S[i] = M.ev_S[i]; s_E[i] = M.ev_E[i]; }
        ev_S = s_S;
        ev_E = s_E;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t *wbase = reinterpret_cast<uint8_t *>(mut8 + 64) + (size_t)wv * fix_wave_bytes(RL);
    uint8_t *ev = wbase;
    uint8_t *stk = ev + rlp;
    uint8_t *qual = stk + rlp;
    uint8_t *tmpl = qual + 2 * rlp;  // rlp + 64 entries
    int16_t *map = reinterpret_cast<int16_t *>(tmpl + rlp + 64);
    int16_t *rec_n0 = map + (rlp + 64);
    int16_t *rec_k0 = rec_n0 + (rlp + 64);
    uint16_t *dqm = reinterpret_cast<uint16_t *>(rec_k0 + (rlp + 64));    // [rlp] error-test digit (8 bits) of position j (K_QM)
    uint64_t *act = reinterpret_cast<uint64_t *>(dqm + rlp);              // [FIX_MAX_RL / 64] steps with an event
    const int n_pre = RL + 64;
    const int n_chunks = rlp / 64;  // 64-step chunks covering indices 0 .. RL-1
    MutChunk mchunk = {0u, MUT_CHUNK};
    for (uint32_t i = blockIdx.x * FIX_WAVES + wv; i < n_fix; i += gridDim.x * FIX_WAVES) {
        const uint32_t e = fix_list[i];
        const uint32_t pair = e >> 1;
        const int o = (int)(e & 1u);
        PairDesc d = desc[pair];
        DevGenome gl = g;  // the record of the read: the launch's genome, or its slice of the arena (batch calls)
        int64_t coord_off = 0;
        if (A.items) {
            const BatchItem it = A.items[batch_item_of(A, A.pair_base + pair)];
            gl = batch_genome(g, it);
            coord_off = it.off;
        }
        const Addr a = make_addr(A.seed, A.first_ordinal + pair, d.meta >> 16);
        const MateGeom geo = mate_geom(o, d, RL, gl.L, coord_off);
        uint8_t *out_base = A.out[2 * o] + (size_t)pair * M.row;
        const uint8_t *out_qual = A.out[2 * o + 1] + (size_t)pair * M.row;
        // ---- phase 0: the error-test digits, one Philox block per LANE (a K_QM block holds those of 8 positions)
        for (int b = lane; b * 8 < RL; b += 64) {
            const u32x4 w = draw_block(a, K_QM, (uint32_t)b, 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) dqm[b * 8 + c] = (uint16_t)((word_of(w, (c >> 2) * 2 + o) >> (8 * (c & 3))) & 0xffu);
        }
        // ---- phase 1: event masks (every lane runs the read's event process -- wave-uniform -- lane 0 writes), template,
        //      phred row
        for (int n = lane; n < rlp; n += 64) ev[n] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        indel_events(ev_S + (size_t)o * M.ev_ns, ev_E + (size_t)o * M.ev_ns, M.ev_T + (size_t)o * M.ev_ns, M.del_thr + (size_t)o * RL * 4,
                     M.ev_ns, a, o, [&](int n, uint32_t mask) { if (lane == 0) ev[n] |= (uint8_t)mask; });
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            const int n = c * 64 + lane;
            const uint32_t m8 = n < RL - 1 ? (uint32_t)ev[n] : 0u;
            if (n < RL) { map[n] = FIX_NONE; qual[n] = out_qual[xp(n)]; }
            { const uint64_t am = __ballot(m8 != 0); if (lane == 0) act[c] = am; }
        }
        if (!gl.has_exceptions && geo.t_len == RL && (o == 0 ? geo.lo + n_pre <= gl.L : geo.hi - n_pre >= 0 && geo.rs == geo.lo)) {
            // the whole staged stretch lies inside a record of plain A/C/G/T: 16 letters per lane from one packed word
            // (not for a reverse mate whose slice Python wrapped around -- custom fragment lengths, negative bounds: its
            //  tokens behind the template follow the UN-normalised start, 'A' below zero: geom_base)
            const int64_t g0 = o == 0 ? geo.lo : geo.hi - n_pre;  // lowest genome position of the stretch
            for (int wi = lane; wi * 16 < n_pre + 16; wi += 64) {
                const int64_t gp = ((g0 >> 4) + wi) << 4;  // genome position of the word's first base
                const uint32_t w = gl.packed[(g0 >> 4) + wi];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int64_t pos = gp + c;
                    const int64_t k = o == 0 ? pos - geo.lo : geo.hi - 1 - pos;
                    const uint32_t code = (w >> (2 * c)) & 3u;
                    if (k >= 0 && k < n_pre) tmpl[k] = code_to_ascii(o ? code ^ 1u : code);
                }
            }
        } else {
            for (int k = lane; k < n_pre; k += 64) tmpl[k] = (uint8_t)geom_base(gl, o, geo, k);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: every lane runs the same walk (wave-uniform values); lane 0 does the LDS writes
        MutRecord row;  // --store_mutations row being built (wave-uniform)
        row.pair = (int32_t)(A.pair_base + pair); row.mate = (int8_t)o; row.quality = -1;
        int sp = 0, k = 0, n_rec = 1, last = -1;  // `last`: last step whose map entry / record is settled
        if (lane == 0) { rec_n0[0] = 0; rec_k0[0] = 0; }
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            uint64_t m = act[c];
            while (m) {
                const int b = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const int n = c * 64 + b;
                k += n - (last + 1);  // the settled record covers steps last+1 .. n-1 from the source
                const uint32_t m8 = ev[n];
                int tok = sp > 0 ? -(int)stk[--sp] : k++;
                if (tok < geo.t_len) {  // tok >= len(template): n >= len(seq), IndexError swallowed (:223): emitted unvisited
                    const int ch = tok < 0 ? -tok : (int)tmpl[tok];
                    const int bi = base_index(ch);
                    if (bi >= 0) {  // else ambiguous: skipped (:190-192)
                        for (int x = 0; x < 4; ++x)
                            if ((m8 >> x) & 1u) {
                                if (sp == rlp) {  // only the top RL entries can ever surface: drop the bottom
                                    if (lane == 0) for (int z = 1; z < sp; ++z) stk[z - 1] = stk[z];
                                    --sp;
                                }
                                const int letter = M.ins_letter[((size_t)o * RL + n) * 4 + x];
                                if (lane == 0) stk[sp] = (uint8_t)letter;
                                ++sp;
                                if (A.mut) {  // ref = mutable_seq[position], alt = ref + letter (__init__.py:197-207)
                                    row.type = (int8_t)(1 | (x << 2) | 32); row.position = (int16_t)n;
                                    row.ref = (uint8_t)ch; row.alt = (uint8_t)letter;
                                    mut_emit(A, mchunk, lane == 0, row);
                                }
                            }
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        if ((m8 >> (4 + bi)) & 1u) {  // deleted: next token slides in
                            const bool exists = sp > 0 || k < geo.t_len;  // else mutable_seq[position] raises IndexError: no row
                            tok = sp > 0 ? -(int)stk[--sp] : k++;
                            if (A.mut && exists) {  // ref = mutable_seq[position] after the pop (__init__.py:211-221)
                                row.type = (int8_t)(2 | (4 << 2) | 32); row.position = (int16_t)n;
                                row.ref = (uint8_t)(tok < 0 ? -tok : (tok < n_pre ? (int)tmpl[tok] : geom_base(gl, o, geo, tok)));
                                row.alt = '.';
                                mut_emit(A, mchunk, lane == 0, row);
                            }
                        }
                    }
                }
                if (lane == 0) map[n] = (int16_t)tok;
                last = n;
                // steps after n drain the insertion stack until it is empty or the next active step
                while (sp > 0 && last + 1 < RL && ev[last + 1] == 0) {
                    ++last;
                    --sp;
                    if (lane == 0) map[last] = (int16_t)(-(int)stk[sp]);
                }
                if (lane == 0) { rec_n0[n_rec] = (int16_t)(last + 1); rec_k0[n_rec] = (int16_t)k; }
                ++n_rec;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 3
        for (int j = lane; j < RL; j += 64) {
            int tok = map[j];
            if (tok == FIX_NONE) {  // governed by the last record with n0 <= j
                int r = n_rec - 1;
                while (rec_n0[r] > j) --r;
                tok = rec_k0[r] + (j - rec_n0[r]);
            }
            int base = tok < 0 ? -tok : (tok < n_pre ? (int)tmpl[tok] : geom_base(gl, o, geo, tok));
            const uint32_t e8 = dqm[j];
            const int q = qual[j];
            const uint32_t t8 = mut8[q];
            bool err = e8 > t8;
            const int before = base;
            if (e8 >= t8) {
                const u32x4 sb = draw_block(a, K_SUB, (uint32_t)j, (uint32_t)o);
                if (e8 == t8) err = error_test_draw(e8, sb) > M.mut_thr[q];
                if (err) base = substitute(M, sb, o, j, base);
            }
            out_base[xp(j)] = (uint8_t)base;
            if (A.mut) {  // only if the new letter differs from the ORIGINAL read at this index (__init__.py:98)
                MutRecord sub;
                sub.pair = (int32_t)(A.pair_base + pair); sub.mate = (int8_t)o; sub.type = (int8_t)32; sub.position = (int16_t)j;
                sub.ref = (uint8_t)before; sub.alt = (uint8_t)base; sub.quality = (int16_t)q;
                // (a read position past a template the genome end cut short has no "original" letter: the reference raises
                //  IndexError there; the row is kept)
                mut_emit(A, mchunk, err && base_index(before) >= 0 && (j >= geo.t_len || base != (int)tmpl[j]), sub);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace iss
