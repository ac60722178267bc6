/*
 * iss_oracle.c -- CPU restatement of the InSilicoSeq read-generation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under insilicoseq_amd/ (the product) may
 * import, link or execute this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker / the
 * reported CPU baseline.  The product path is the HIP library in
 * insilicoseq_amd/csrc and fails loudly when that library is missing.
 *
 * What it restates (reference = HadrienG/InSilicoSeq v2.0.1, file:line):
 *   simulate_read            iss/generator.py:98-192
 *   reads_generator/gc_bias  iss/generator.py:69-95
 *   introduce_indels         iss/error_models/__init__.py:158-228
 *   adjust_seq_length        iss/error_models/__init__.py:114-156
 *   introduce_error_scores   iss/error_models/__init__.py:52-67
 *   gen_phred_scores (kde)   iss/error_models/kde.py:52-86
 *   gen_phred_scores (basic) iss/error_models/basic.py:40-54
 *   random_insert_size       iss/error_models/kde.py:88-98
 *   mut_sequence             iss/error_models/__init__.py:69-112
 *   phred_to_prob, rev_comp  iss/util.py:16-29, 48-92
 * Third-party arithmetic the reference calls (not under /root/reference),
 * restated from their published algorithms:
 *   CPython 3.10 `random`  (MT19937, init_by_array seeding, genrand_res53,
 *                           _randbelow_with_getrandbits)
 *   numpy 1.26 legacy RandomState (MT19937 init_genrand seeding, random_sample,
 *                           choice(p=) == cumsum/normalise/searchsorted-right,
 *                           legacy polar Box-Muller gauss with cached value)
 *
 * Parity pin: tests/test_oracle_golden.py checks this file, in MT mode, bit for
 * bit (outputs AND stream consumption) against golden vectors captured by
 * importing the reference in the build container
 * (tests/golden/tooling/make_golden.py), and against the reference's own unit
 * goldens (iss/test/test_error_model.py, iss/test/test_generator.py).
 *
 * Two uniform-stream providers:
 *   ISS_RNG_MT      two sequential MT19937 streams (CPython `random` + numpy),
 *                   consumed in the reference's exact order -> equals the
 *                   reference for a given seed;
 *   ISS_RNG_PHILOX  Philox4x32 (7 rounds for the hot digit blocks, 10 otherwise: philox_at), every draw addressed by
 *                   (pair ordinal, attempt, kind, index, sub, word) -> equals the
 *                   HIP kernels (which use the same address map).
 * They feed the same semantic function draw for draw EXCEPT in two places, where the
 * Philox provider samples the same distribution from fewer uniforms (branches the MT
 * goldens cannot reach, pinned separately):
 *   - introduce_indels: MT mode draws `random() < p` per test (:193-196, :209);
 *     Philox mode reads the tests' outcomes from the indel event process
 *     (ev_step / sample_indel_events: one uniform per FIRING test).  Pin:
 *     tests/test_oracle_golden.py::test_event_process_exact derives, for every
 *     state and slot, the exact set of uniforms for which ev_step fires there and
 *     compares its measure with the reference's p_t * prod(1 - p_s) in exact
 *     rational arithmetic (bound: 2^-52 absolute + 2^-47 relative), plus the
 *     nesting and measure of the per-base deletion events;
 *   - BasicErrorModel phreds: MT mode draws the legacy-gauss vector (basic.py:52-53),
 *     Philox mode inverts the score's CDF (insilicoseq_amd/model.py
 *     basic_phred_cdf; pinned at every step boundary of the reference expression).
 * All comparisons here are IEEE f64 on the raw model tables (the device uses an
 * integer-threshold formulation instead -- an independent computation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ISS_RNG_MT 0
#define ISS_RNG_PHILOX 1

#define ISS_OK 0
#define ISS_SKIP_RECORD 1  /* AssertionError: read_length >= len(record)      */
#define ISS_ERR_KEY 2      /* KeyError in the reference (non-IUPAC letter)    */
#define ISS_ERR_INDEX 3    /* uncaught IndexError in the reference            */
#define ISS_ERR_UNSUPPORTED 4

/* ------------------------------------------------------------------ MT19937 */
typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937;

static void mt_init_genrand(mt19937 *m, uint32_t s) {
    m->mt[0] = s;
    for (int i = 1; i < 624; i++)
        m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}

static void mt_init_by_array(mt19937 *m, const uint32_t *key, int len) {
    mt_init_genrand(m, 19650218u);
    int i = 1, j = 0;
    int k = (624 > len) ? 624 : len;
    for (; k; k--) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = 623; k; k--) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
    }
    m->mt[0] = 0x80000000u;
    m->idx = 624;
}

static uint32_t mt_next(mt19937 *m) {
    if (m->idx >= 624) {
        uint32_t *mt = m->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double res53(uint32_t w0, uint32_t w1) {
    uint32_t a = w0 >> 5, b = w1 >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------ Philox4x32-R
 * Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11).  R = 10 is the generator's default;
 * R = 7 is the fewest rounds the authors found to pass all of BigCrush ("Crush-resistant") and what the Philox address map
 * uses for the HOT digit blocks (K_QM) since round 4 -- see philox_at and iss_kernels.hip.h (draw_block). */
#define PHILOX_HOT_ROUNDS 7
static void philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void iss_oracle_philox4x32(const uint32_t *ctr, const uint32_t *key, int rounds, uint32_t *out) {
    philox4x32(ctr, key, rounds, out);
}

/* Draw kinds of the Philox address map (DESIGN.md "RNG address map").
 * A "digit draw" builds its 53-bit numerator as m = (digit << (53 - w)) | trailing bits: the w-bit leading
 * digit comes from a PRIMARY block shared by many draws (the device compares digits and needs three Philox
 * calls per 16 bases), the trailing bits from a SECONDARY block that the device only evaluates when the
 * digit ties with a threshold's.  w = 16 for the quality draw, 8 for the substitution test (kde.py:84,
 * __init__.py:94).  u = m / 2^53 exactly.  The insertion / deletion tests (__init__.py:194, :209) are not drawn one
 * by one in this mode: see ev_tab below. */
enum {
    K_PAIR = 0,   /* index 0; full draws mk53(sub0.word[s], sub1.word[s]), s = isize, bin_fwd, bin_rev, gc */
    K_FS = 1,     /* forward-start randbelow words: word t -> index t/4, word t%4        */
    K_RS = 2,     /* reverse-end fallback randbelow words, same addressing               */
    K_QM = 3,     /* hot digits of the 8 positions of superitem s = p>>3 (both mates): index = s, sub = 0, 1, 2.
                   * c = p&7, half = c>>2, cc = c&3.  Quality digit h16 (16 bits): block sub = 2*half, word
                   * mate*2 + (cc>>1), 16-bit half cc&1.  Error-test digit e8 (8 bits): block sub = 1, word
                   * half*2 + mate, byte cc.                                                        */
    K_SUB = 4,    /* index = p, sub = mate: (w0,w1) substitution choice (full draw); the 45 trailing bits of the
                   * error-test draw = (w2 & 0x1fff) << 32 | w3                                      */
    /* 5, 6, 8, 9: retired (per-test indel draws, replaced by K_EV) */
    K_QM_LO = 7,  /* trailing 37 bits of the quality draw; index = p; sub = mate; (w0,w1)  */
    K_FRAG = 10,  /* custom fragment length: polar candidate t -> index t; x1 from mk53(w0,w1), x2 from mk53(w2,w3) */
    K_EV = 11     /* indel events of a read: draw j of the event process (see ev_tab) -> index j, sub = mate;
                   * (w0,w1): where the next test fires; (w2,w3): which bases a deletion test fires for       */
};

/* ------------------------------------------------------------- RNG provider */
typedef struct {
    int mode;
    mt19937 py;  /* CPython `random` module stream */
    mt19937 np_; /* numpy legacy global RandomState stream */
    int has_gauss;
    double gauss;
    uint32_t key[2];
    uint64_t ordinal; /* address of the current pair (Philox mode) */
    uint32_t attempt;
    uint64_t n_py_words, n_np_words; /* consumption counters (MT mode) */
} iss_rng;

iss_rng *iss_oracle_rng_new(void) { return (iss_rng *)calloc(1, sizeof(iss_rng)); }
void iss_oracle_rng_free(iss_rng *r) { free(r); }

/* random.seed(int) + np.random.seed(int) -- generator.py:234-236, 397-400 */
void iss_oracle_rng_seed_mt(iss_rng *r, uint64_t seed) {
    r->mode = ISS_RNG_MT;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    mt_init_by_array(&r->py, key, key[1] ? 2 : 1);
    mt_init_genrand(&r->np_, (uint32_t)seed); /* numpy requires 0 <= seed < 2**32 */
    r->has_gauss = 0;
    r->gauss = 0.0;
    r->n_py_words = r->n_np_words = 0;
}
void iss_oracle_rng_seed_py(iss_rng *r, uint64_t seed) {
    r->mode = ISS_RNG_MT;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    mt_init_by_array(&r->py, key, key[1] ? 2 : 1);
}
void iss_oracle_rng_seed_np(iss_rng *r, uint32_t seed) {
    r->mode = ISS_RNG_MT;
    mt_init_genrand(&r->np_, seed);
    r->has_gauss = 0;
    r->gauss = 0.0;
}
void iss_oracle_rng_seed_philox(iss_rng *r, uint64_t seed) {
    r->mode = ISS_RNG_PHILOX;
    r->key[0] = (uint32_t)seed;
    r->key[1] = (uint32_t)(seed >> 32);
    r->ordinal = 0;
    r->attempt = 0;
}
void iss_oracle_rng_set_address(iss_rng *r, uint64_t ordinal, uint32_t attempt) {
    r->ordinal = ordinal;
    r->attempt = attempt;
}
double iss_oracle_py_random(iss_rng *r) { uint32_t a = mt_next(&r->py), b = mt_next(&r->py); r->n_py_words += 2; return res53(a, b); }
double iss_oracle_np_random(iss_rng *r) { uint32_t a = mt_next(&r->np_), b = mt_next(&r->np_); r->n_np_words += 2; return res53(a, b); }
uint32_t iss_oracle_py_word(iss_rng *r) { r->n_py_words++; return mt_next(&r->py); }
uint32_t iss_oracle_np_word(iss_rng *r) { r->n_np_words++; return mt_next(&r->np_); }
uint64_t iss_oracle_py_words_used(iss_rng *r) { return r->n_py_words; }
uint64_t iss_oracle_np_words_used(iss_rng *r) { return r->n_np_words; }

static void philox_at(const iss_rng *r, int kind, uint32_t index, uint32_t sub, uint32_t out[4]) {
    uint32_t ctr[4];
    ctr[0] = (uint32_t)r->ordinal;
    ctr[1] = (uint32_t)((r->ordinal >> 32) & 0xffffu) | (r->attempt << 16);
    ctr[2] = ((uint32_t)kind << 24) | (index & 0xffffffu);
    ctr[3] = sub;
    philox4x32(ctr, r->key, kind == K_QM ? PHILOX_HOT_ROUNDS : 10, out);
}

#define STREAM_PY 0
#define STREAM_NP 1

/* Full-width draw: MT mode = next double of `stream`; Philox mode = mk53 of two words.
 * two_block != 0: words `slot` of blocks (kind,index,sub0) and (kind,index,sub0+1);
 * two_block == 0: words 2*slot, 2*slot+1 of block (kind,index,sub0). */
static double draw_double(iss_rng *r, int stream, int kind, uint32_t index, uint32_t sub0, int slot, int two_block) {
    if (r->mode == ISS_RNG_MT)
        return stream == STREAM_PY ? iss_oracle_py_random(r) : iss_oracle_np_random(r);
    uint32_t w[4], w2[4];
    philox_at(r, kind, index, sub0, w);
    if (two_block) {
        philox_at(r, kind, index, sub0 + 1, w2);
        return res53(w[slot], w2[slot]);
    }
    return res53(w[2 * slot], w[2 * slot + 1]);
}

/* The two hot draws of read position p of mate o (see K_QM / K_SUB / K_QM_LO in the enum). */
static double draw_quality(iss_rng *r, int p, int o) { /* kde.py:84 */
    if (r->mode == ISS_RNG_MT) return iss_oracle_np_random(r);
    const int c = p & 7, half = c >> 2, cc = c & 3;
    uint32_t wp[4], wl[4];
    philox_at(r, K_QM, (uint32_t)p >> 3, (uint32_t)(2 * half), wp);
    philox_at(r, K_QM_LO, (uint32_t)p, (uint32_t)o, wl);
    uint64_t h16 = (wp[o * 2 + (cc >> 1)] >> (16 * (cc & 1))) & 0xffffu;
    uint64_t l37 = ((uint64_t)wl[0] << 5) | (wl[1] >> 27);
    return (double)((h16 << 37) | l37) * (1.0 / 9007199254740992.0);
}
static double draw_error_test(iss_rng *r, int p, int o) { /* __init__.py:94 */
    if (r->mode == ISS_RNG_MT) return iss_oracle_py_random(r);
    const int c = p & 7, half = c >> 2, cc = c & 3;
    uint32_t wp[4], wl[4];
    philox_at(r, K_QM, (uint32_t)p >> 3, 1u, wp);
    philox_at(r, K_SUB, (uint32_t)p, (uint32_t)o, wl);
    uint64_t e8 = (wp[half * 2 + o] >> (8 * cc)) & 0xffu;
    uint64_t l45 = ((uint64_t)(wl[2] & 0x1fffu) << 32) | wl[3];
    return (double)((e8 << 45) | l45) * (1.0 / 9007199254740992.0);
}

static int bit_length64(uint64_t n) { int k = 0; while (n) { k++; n >>= 1; } return k; }

/* CPython Random._randbelow_with_getrandbits(n), n > 0 (random.py) with
 * getrandbits(k) as in _randommodule.c (k<=32: one word >> (32-k); else
 * little-endian 32-bit words, the top one shifted). */
static uint64_t py_randbelow(iss_rng *r, uint64_t n, int kind) {
    int k = bit_length64(n);
    uint32_t t = 0; /* running word index for the Philox address */
    for (;;) {
        uint64_t v = 0;
        int rem = k, shift = 0;
        while (rem > 0) {
            uint32_t w;
            if (r->mode == ISS_RNG_MT) {
                w = iss_oracle_py_word(r);
            } else {
                uint32_t blk[4];
                philox_at(r, kind, t >> 2, 0, blk);
                w = blk[t & 3u];
                t++;
            }
            if (rem < 32) w >>= (32 - rem);
            v |= (uint64_t)w << shift;
            shift += 32;
            rem -= 32;
        }
        if (v < n) return v;
    }
}

/* numpy legacy_gauss (polar Box-Muller with cached second value); MT mode only */
static double np_legacy_gauss(iss_rng *r) {
    if (r->has_gauss) {
        double t = r->gauss;
        r->has_gauss = 0;
        r->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * iss_oracle_np_random(r) - 1.0;
        x2 = 2.0 * iss_oracle_np_random(r) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    r->gauss = f * x1;
    r->has_gauss = 1;
    return f * x2;
}
double iss_oracle_np_normal(iss_rng *r, double loc, double scale) { return loc + scale * np_legacy_gauss(r); }

/* ---------------------------------------------------------------- the model */
typedef struct {
    int32_t read_length;
    int32_t n_isize;
    int32_t n_q;            /* entries per per-position quality CDF (41)            */
    int32_t quality_mode;   /* 0 = kde tables, 1 = basic (normal around Q30)        */
    const double *isize_cdf;    /* [n_isize]              kde.py:31                 */
    const double *bin_cdf;      /* [2][4]  cumsum(mean/sum(mean)) / last, kde.py:72-74 */
    const double *qcdf;         /* [2][4][RL][n_q]        kde.py:80-85              */
    const double *subst_cdf;    /* [2][RL][4][3] cumsum(p)/last  __init__.py:95-97  */
    const uint8_t *subst_alt;   /* [2][RL][4][3] ASCII                              */
    const double *ins;          /* [2][RL][4]  in dict iteration order __init__.py:193 */
    const uint8_t *ins_letter;  /* [2][RL][4]  ASCII                                */
    const double *del;          /* [2][RL][4]  indexed A,T,C,G   __init__.py:209    */
    const double *phred_thr;    /* [n_q+1]  1 - 10**(-q/10)      util.py:28-29      */
    int32_t basic_insert_size;  /* basic.py:21 (200)                                */
    int32_t basic_mean_quality; /* basic.py:24 (30)                                 */
} iss_model;

static int base_index(int c) { /* A,T,C,G -> 0..3 (upper-cased input), else -1 */
    switch (c) { case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3; default: return -1; }
}
static int upper_c(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
static int is_ambiguous_upper(int cu) { /* nucl.upper() in "RYWSMKHBVDN" */
    return cu && strchr("RYWSMKHBVDN", cu) != NULL;
}
/* util.rev_comp's dict (util.py:57-88); returns 0 for a KeyError */
static int complement_char(int c) {
    switch (c) {
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        case 'y': return 'r'; case 'r': return 'y'; case 'w': return 'w'; case 's': return 's';
        case 'k': return 'm'; case 'm': return 'k'; case 'n': return 'n'; case 'b': return 'v';
        case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'Y': return 'R'; case 'R': return 'Y'; case 'W': return 'W'; case 'S': return 'S';
        case 'K': return 'M'; case 'M': return 'K'; case 'N': return 'N'; case 'B': return 'V';
        case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        default: return 0;
    }
}

static int64_t searchsorted_left(const double *a, int64_t n, double v) {
    int64_t lo = 0, hi = n; /* first i with a[i] >= v */
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
static int64_t searchsorted_right(const double *a, int64_t n, double v) {
    int64_t lo = 0, hi = n; /* first i with a[i] > v */
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* VCF-row capture (store_mutations), __init__.py:98-108, 197-221 */
typedef struct {
    int32_t pair;      /* pair index i within the call */
    int8_t mate;       /* 0 fwd, 1 rev */
    int8_t type;       /* 0 sub, 1 ins, 2 del */
    int16_t position;  /* 0-based */
    uint8_t ref;       /* ASCII */
    uint8_t alt;       /* ASCII (sub: new base; ins: inserted letter; del: '.') */
    int16_t quality;   /* sub: phred; else -1 ('.') */
} iss_mutation;

typedef struct {
    iss_mutation *buf;
    int64_t cap, n;
    int32_t pair;
} mut_sink;

static void mut_push(mut_sink *s, int mate, int type, int pos, int ref, int alt, int qual) {
    if (!s || !s->buf) return;
    if (s->n < s->cap) {
        iss_mutation *m = &s->buf[s->n];
        m->pair = s->pair; m->mate = (int8_t)mate; m->type = (int8_t)type; m->position = (int16_t)pos;
        m->ref = (uint8_t)ref; m->alt = (uint8_t)alt; m->quality = (int16_t)qual;
    }
    s->n++;
}

/* ------------------------------------------------------------ indel events (position-addressable mode)
 * introduce_indels runs, per loop step n <= RL-2, four insertion tests `random() < p_ins[n][x]` and one deletion test
 * `random() < p_del[n][base]` (__init__.py:193-196, :209): 5 (RL - 1) independent Bernoulli tests per read, all but a
 * handful failing.  Drawing them one by one is what the reference (and the MT mode here) does; in the
 * position-addressable mode the SAME joint distribution is sampled by skipping from one firing test to the next:
 *   slot t = 5 n + k, k = 0..3 insertion of letter slot k, k = 4 the deletion; its probability is T[t] / 2^53 with
 *   T = ceil(p * 2^53) (`u < p`  <=>  `m < T` for the 53-bit numerator m of u), the deletion slot carrying the
 *   largest of its four bases' thresholds;
 *   S[t] = prod_{t' <= t, same segment} (1 - T[t'] / 2^53) in 0.64 fixed point (floor after every factor) -- the
 *   probability that no test of the segment up to t fires; a new segment starts after a slot that leaves S below
 *   2^-16 (precision) -- in particular after a test that always fires;
 *   one uniform r in (0, 1] per draw: the next firing test after slot `cur` is the first t of cur's segment with
 *   S[t] <= r * S[cur] (P(none up to t | none up to cur) = S[t] / S[cur]); none: the draw is spent, go on with the
 *   next segment.  A firing deletion slot fires for base b iff v * T_max < T_b * 2^53 with a second uniform numerator
 *   v of the same block (the four bases share the reference's ONE uniform: nested events).
 * All of it is integer arithmetic (64 x 64 -> 128 bit products): the device repeats it bit for bit.  The result is the
 * event mask of every step -- bits 0-3: insertion of letter slot x fires, bits 4-7: the deletion fires if the token is
 * base b -- whether or not the loop visits the step. */
typedef struct {
    int ns;          /* 5 * (RL - 1) slots */
    uint64_t *S;     /* [ns]   survival inside the segment, 0.64 fixed point */
    uint16_t *E;     /* [ns]   last slot of the slot's segment               */
    uint64_t *T;     /* [ns]   threshold of the slot (deletion: the largest) */
    uint64_t *Tdel;  /* [RL-1][4] deletion thresholds per base                */
    int any;         /* some T != 0 */
} ev_tab;
#define EV_ONE 0xffffffffffffffffull
#define EV_RESTART (1ull << 48)

static uint64_t thr_of(double p) { /* ceil(p * 2^53): u < p  <=>  m < thr */
    double x = ceil(p * 9007199254740992.0);
    if (!(x > 0.0)) return 0;
    if (x >= 9007199254740992.0) return 1ull << 53;
    return (uint64_t)x;
}
static void ev_tab_build(const iss_model *m, int o, ev_tab *t) {
    const int RL = m->read_length, ns = 5 * (RL - 1);
    t->ns = ns;
    t->S = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(ns > 0 ? ns : 1));
    t->T = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(ns > 0 ? ns : 1));
    t->E = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(ns > 0 ? ns : 1));
    t->Tdel = (uint64_t *)malloc(sizeof(uint64_t) * 4 * (size_t)(RL > 1 ? RL - 1 : 1));
    t->any = 0;
    uint64_t prev = EV_ONE;
    int seg_start = 0;
    for (int s = 0; s < ns; s++) {
        const int n = s / 5, k = s % 5;
        uint64_t T;
        if (k < 4) {
            T = thr_of(m->ins[((size_t)o * RL + n) * 4 + k]);
        } else {
            T = 0;
            for (int b = 0; b < 4; b++) {
                uint64_t tb = thr_of(m->del[((size_t)o * RL + n) * 4 + b]);
                t->Tdel[(size_t)n * 4 + b] = tb;
                if (tb > T) T = tb;
            }
        }
        t->T[s] = T;
        if (T) t->any = 1;
        const uint64_t cur = (uint64_t)(((unsigned __int128)prev * ((1ull << 53) - T)) >> 53);
        t->S[s] = cur;
        if (cur < EV_RESTART || s == ns - 1) { /* the segment ends here */
            for (int q = seg_start; q <= s; q++) t->E[q] = (uint16_t)s;
            seg_start = s + 1;
            prev = EV_ONE;
        } else {
            prev = cur;
        }
    }
}
static void ev_tab_free(ev_tab *t) { free(t->S); free(t->T); free(t->E); free(t->Tdel); }

/* ONE draw of the event process.  State `cur`: the last slot decided (-1: none yet).  m53: the numerator of the draw's
 * uniform (words 0-1 of the K_EV block), v53: that of the deletion sub-draw (words 2-3).  Returns the new state; *slot =
 * the slot that fired (then *mask = its event mask) or -1 (nothing fires in the rest of cur's segment; the state is the
 * segment's last slot and the next draw starts the next segment).  sample_indel_events() below is a loop over this
 * function, and iss_oracle_ev_step exports it: tests/test_oracle_golden.py::test_event_process_exact walks its
 * interval boundaries in m53 with exact integer arithmetic against the reference's per-test probabilities. */
static int ev_step(const ev_tab *t, int cur, uint64_t m53, uint64_t v53, int *slot, uint32_t *mask) {
    const int seg_last = t->E[cur + 1];
    const uint64_t base = (cur >= 0 && t->E[cur] == seg_last) ? t->S[cur] : EV_ONE;
    const uint64_t rr = (((1ull << 53) - m53) << 11) - 1; /* (1 - u) in 0.64 fixed point, (0, 1] */
    const uint64_t target = (uint64_t)(((unsigned __int128)rr * base) >> 64);
    *slot = -1;
    *mask = 0;
    if (t->S[seg_last] > target) return seg_last; /* nothing fires in the rest of the segment */
    int lo = cur + 1, hi = seg_last; /* first slot with S <= target */
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (t->S[mid] <= target) hi = mid; else lo = mid + 1;
    }
    const int n = lo / 5, k = lo % 5;
    if (k < 4) {
        *mask = 1u << k;
    } else {
        const uint64_t scaled = (uint64_t)(((unsigned __int128)v53 * t->T[lo]) >> 53); /* floor(v * T_max / 2^53) */
        for (int b = 0; b < 4; b++)
            if (scaled < t->Tdel[(size_t)n * 4 + b]) *mask |= 16u << b;
    }
    *slot = lo;
    return lo;
}

/* the event masks of one read: mask[0 .. RL-1) */
static void sample_indel_events(const ev_tab *t, const iss_rng *r, int o, uint8_t *mask, int RL) {
    memset(mask, 0, (size_t)RL);
    if (!t->any) return;
    int cur = -1;
    uint32_t j = 0;
    while (cur < t->ns - 1) {
        uint32_t w[4], m8;
        int slot;
        philox_at(r, K_EV, j++, (uint32_t)o, w);
        cur = ev_step(t, cur, ((uint64_t)(w[0] >> 5) << 26) | (w[1] >> 6), ((uint64_t)(w[2] >> 5) << 26) | (w[3] >> 6), &slot, &m8);
        if (slot >= 0) mask[slot / 5] |= (uint8_t)m8;
    }
}
/* the tables of the call in progress (built by the exported entry points; one call per thread at a time) */
static __thread const ev_tab *g_ev = NULL;

/* ------------------------------------------------------------ introduce_indels
 * __init__.py:158-228 + adjust_seq_length :114-156.
 * seq/len: the perfect read (<= RL chars); out: exactly RL chars.
 * genome/L: full_sequence; (start,end): bounds.  Returns ISS_OK / ISS_ERR_*.   */
static int introduce_indels(const iss_model *m, iss_rng *r, int o, const uint8_t *seq, int len,
                            const uint8_t *genome, int64_t L, int64_t start, int64_t end, uint8_t *out,
                            mut_sink *sink) {
    const int RL = m->read_length;
    /* a list that can grow by at most 4 per visited position */
    int cap = len + 4 * RL + 8;
    uint8_t *s = (uint8_t *)malloc((size_t)cap);
    memcpy(s, seq, (size_t)len);
    int n_s = len;
    int position = 0;
    const int philox = r->mode != ISS_RNG_MT;
    uint8_t *evm = NULL; /* position-addressable mode: which tests fire at every step (see ev_tab) */
    if (philox) {
        evm = (uint8_t *)malloc((size_t)RL);
        sample_indel_events(&g_ev[o], r, o, evm, RL);
    }
    for (int nucl = 0; nucl < RL - 1; nucl++) {
        if (nucl >= n_s) continue; /* IndexError swallowed, :223-224 (position not advanced) */
        int cu = upper_c(s[nucl]);
        if (is_ambiguous_upper(cu)) { position++; continue; } /* :190-192 */
        const double *insp = m->ins + ((size_t)o * RL + position) * 4;
        const uint8_t *insl = m->ins_letter + ((size_t)o * RL + position) * 4;
        for (int x = 0; x < 4; x++) { /* :193-196, dict order */
            const int fires = philox ? (evm[position] >> x) & 1 : iss_oracle_py_random(r) < insp[x];
            if (fires) {
                memmove(s + position + 2, s + position + 1, (size_t)(n_s - position - 1));
                s[position + 1] = insl[x];
                n_s++;
                /* ref = mutable_seq[position], alt = ref + letter */
                mut_push(sink, o, 1, position, s[position], insl[x], -1);
            }
        }
        int bi = base_index(cu);
        if (bi < 0) { free(s); free(evm); return ISS_ERR_KEY; } /* deletions[position][X] KeyError :209 */
        const int del_fires = philox ? (evm[position] >> (4 + bi)) & 1
                                     : iss_oracle_py_random(r) < m->del[((size_t)o * RL + position) * 4 + bi];
        if (del_fires) {
            memmove(s + position, s + position + 1, (size_t)(n_s - position - 1));
            n_s--;
            if (sink && sink->buf) {
                /* ref = mutable_seq[position] AFTER the pop (:216); IndexError there is swallowed */
                if (position < n_s) mut_push(sink, o, 2, position, s[position], '.', -1);
                else continue; /* IndexError -> except: continue (position not advanced) */
            }
        }
        position++;
    }
    /* adjust_seq_length */
    if (n_s >= RL) {
        memcpy(out, s, (size_t)RL);
    } else {
        memcpy(out, s, (size_t)n_s);
        int to_add = RL - n_s;
        for (int i = 0; i < to_add; i++) {
            int c;
            if (o == 0) {
                c = (end + i >= L) ? 'A' : genome[end + i];
            } else {
                int64_t idx = start - 1 - i;
                if (idx < 0) c = 'A';
                else {
                    if (idx >= L) { free(s); free(evm); return ISS_ERR_INDEX; }
                    c = complement_char(genome[idx]);
                    if (!c) { free(s); free(evm); return ISS_ERR_KEY; }
                }
            }
            out[n_s + i] = (uint8_t)c;
        }
    }
    free(s);
    free(evm);
    return ISS_OK;
}

/* -------------------------------------------------------------- phred scores */
static double py_round_half_even(double x) { return nearbyint(x); } /* default FE_TONEAREST == round() */

static void gen_phred_scores(const iss_model *m, iss_rng *r, int o, uint8_t *qual) {
    const int RL = m->read_length;
    if (m->quality_mode == 1 && r->mode == ISS_RNG_MT) {
        /* basic.py:52-53: np.random.normal(phred_to_prob(30), 0.01, RL) then prob_to_phred.  (The position-addressable
         * mode below inverts the distribution of that score instead -- the model's quality rows hold it, the same
         * at every position: insilicoseq_amd/model.py basic_phred_cdf -- one uniform per position like a KDE row.) */
        double mean = 1.0 - pow(10.0, -(double)m->basic_mean_quality / 10.0);
        for (int p = 0; p < RL; p++) {
            double q = iss_oracle_np_normal(r, mean, 0.01);
            if (q > 0.9999) q = 0.9999; /* min(q, 0.9999) */
            qual[p] = (uint8_t)(int)py_round_half_even(-10.0 * log10(1.0 - q)); /* util.py:44 */
        }
        return;
    }
    /* kde.py:72-78: np.random.choice(range(4), p=norm_mean) == searchsorted(cdf, u, 'right') */
    double ub = draw_double(r, STREAM_NP, K_PAIR, 0, 0, 1 + o, 1);
    int bin = (int)searchsorted_right(m->bin_cdf + 4 * o, 4, ub);
    if (bin >= 4) bin = 3; /* unreachable (cdf[-1] == 1.0 > u), mirrors kde.py:77-78 */
    const double *rows = m->qcdf + (((size_t)o * 4 + bin) * RL) * m->n_q;
    for (int p = 0; p < RL; p++) { /* kde.py:83-85 */
        double u = draw_quality(r, p, o);
        qual[p] = (uint8_t)searchsorted_left(rows + (size_t)p * m->n_q, m->n_q, u);
    }
}

/* ---------------------------------------------------------------- mut_sequence
 * __init__.py:69-112 */
static int mut_sequence(const iss_model *m, iss_rng *r, int o, uint8_t *seq, const uint8_t *qual,
                        const uint8_t *original, mut_sink *sink) {
    const int RL = m->read_length;
    for (int p = 0; p < RL; p++) {
        double u = draw_error_test(r, p, o); /* always drawn, :94 */
        int cu = upper_c(seq[p]);
        if (u > m->phred_thr[qual[p]] && !is_ambiguous_upper(cu)) {
            int bi = base_index(cu);
            if (bi < 0) return ISS_ERR_KEY;
            const size_t row = (((size_t)o * RL + p) * 4 + bi) * 3;
            double us = draw_double(r, STREAM_NP, K_SUB, (uint32_t)p, (uint32_t)o, 0, 0);
            int k = (int)searchsorted_right(m->subst_cdf + row, 3, us);
            if (k > 2) k = 2; /* unreachable: cdf[-1] == 1.0 */
            uint8_t alt = m->subst_alt[row + k];
            if (sink && sink->buf && alt != original[p]) mut_push(sink, o, 0, p, seq[p], alt, qual[p]);
            seq[p] = alt;
        }
    }
    return ISS_OK;
}

/* Python slice normalisation for seq[a:b] on a sequence of length L (step 1). */
static void py_slice(int64_t a, int64_t b, int64_t L, int64_t *lo, int64_t *hi) {
    if (a < 0) { a += L; if (a < 0) a = 0; } else if (a > L) a = L;
    if (b < 0) { b += L; if (b < 0) b = 0; } else if (b > L) b = L;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

typedef struct {
    int32_t sequence_type; /* 0 metagenomics, 1 amplicon */
    int32_t has_fragment;  /* error_model.fragment_length / fragment_sd both not None */
    double fragment_length;
    double fragment_sd;
    int32_t gc_bias;
} iss_run_params;

/* ---------------------------------------------------------------- simulate_read
 * generator.py:98-192.  Outputs RL bytes per array.                            */
static int simulate_read(const iss_model *m, iss_rng *r, const iss_run_params *rp, const uint8_t *g, int64_t L,
                         uint8_t *f_base, uint8_t *f_qual, uint8_t *r_base, uint8_t *r_qual, int64_t *coords,
                         mut_sink *sink) {
    const int RL = m->read_length;
    int64_t insert_size, fragment_length;
    if (rp->has_fragment) { /* :121-123 */
        double x;
        if (r->mode == ISS_RNG_MT) {
            x = iss_oracle_np_normal(r, rp->fragment_length, rp->fragment_sd);
        } else {
            /* Philox mode: every pair runs its own polar Box-Muller loop on the K_FRAG candidates and takes the
             * value numpy returns for a fresh draw (f * x2); nothing is cached across pairs (they are parallel). */
            double x1, x2, r2;
            uint32_t t = 0;
            do {
                uint32_t w[4];
                philox_at(r, K_FRAG, t++, 0, w);
                x1 = 2.0 * res53(w[0], w[1]) - 1.0;
                x2 = 2.0 * res53(w[2], w[3]) - 1.0;
                r2 = x1 * x1 + x2 * x2;
            } while (r2 >= 1.0 || r2 == 0.0);
            double f = sqrt(-2.0 * log(r2) / r2);
            x = rp->fragment_length + rp->fragment_sd * (f * x2);
        }
        fragment_length = (int64_t)x; /* int(): truncation toward zero */
        insert_size = fragment_length - 2 * (int64_t)RL;
    } else if (m->quality_mode == 1) { /* basic.py:56-63 */
        insert_size = m->basic_insert_size;
        fragment_length = insert_size + 2 * (int64_t)RL;
    } else { /* kde.py:97 */
        double u = draw_double(r, STREAM_NP, K_PAIR, 0, 0, 0, 1);
        insert_size = searchsorted_left(m->isize_cdf, m->n_isize, u);
        fragment_length = insert_size + 2 * (int64_t)RL;
    }
    if (!(RL < L)) return ISS_SKIP_RECORD; /* assert, :130 (after the insert-size draw) */
    int64_t fs;
    if (rp->sequence_type == 0) { /* :134-135, 142-144 */
        int64_t width = L - fragment_length;
        if (width > 0) fs = (int64_t)py_randbelow(r, (uint64_t)width, K_FS);
        else { fs = (int64_t)py_randbelow(r, (uint64_t)(L - RL), K_FS); if (fs < 0) fs = 0; }
    } else {
        fs = 0;
    }
    int64_t fe = fs + RL;
    uint8_t *tmp = (uint8_t *)malloc((size_t)RL + 8);
    uint8_t *orig = (uint8_t *)malloc((size_t)RL + 8);
    int rc;
    /* forward */
    int64_t lo, hi;
    py_slice(fs, fe, L, &lo, &hi);
    int flen = (int)(hi - lo);
    memcpy(tmp, g + lo, (size_t)flen);
    memset(orig, 0, (size_t)RL);
    memcpy(orig, tmp, (size_t)flen);
    rc = introduce_indels(m, r, 0, tmp, flen, g, L, fs, fe, f_base, sink);
    if (rc) goto done;
    gen_phred_scores(m, r, 0, f_qual);
    rc = mut_sequence(m, r, 0, f_base, f_qual, orig, sink);
    if (rc) goto done;
    /* reverse, :164-177 */
    int64_t rs, re;
    if (rp->sequence_type == 0) { rs = fe + insert_size; re = rs + RL; }
    else { rs = L - RL; re = rs + RL; }
    if (re > L) {
        re = RL + (int64_t)py_randbelow(r, (uint64_t)(L - RL), K_RS); /* randrange(RL, L) */
        rs = re - RL;
    }
    py_slice(rs, re, L, &lo, &hi);
    int rlen = (int)(hi - lo);
    for (int i = 0; i < rlen; i++) {
        int c = complement_char(g[hi - 1 - i]);
        if (!c) { rc = ISS_ERR_KEY; goto done; }
        tmp[i] = (uint8_t)c;
    }
    memset(orig, 0, (size_t)RL);
    memcpy(orig, tmp, (size_t)rlen);
    rc = introduce_indels(m, r, 1, tmp, rlen, g, L, rs, re, r_base, sink);
    if (rc) goto done;
    gen_phred_scores(m, r, 1, r_qual);
    rc = mut_sequence(m, r, 1, r_base, r_qual, orig, sink);
    if (coords) { coords[0] = fs; coords[1] = rs; coords[2] = re; coords[3] = insert_size; }
done:
    free(tmp);
    free(orig);
    return rc;
}

/* -------------------------------------------------------------- reads_generator
 * generator.py:69-95: n_pairs pairs from one record; gc_bias = one extra numpy
 * double per candidate pair and a 10 % rejection (the 40<gc<60 window is dead
 * with Biopython >= 1.80 gc_fraction in [0,1]).
 * Arrays are [n_pairs][pitch]; returns status; *n_done = pairs emitted.
 * first_ordinal: Philox address of pair 0 (ignored in MT mode).               */
int iss_oracle_simulate(const iss_model *m, iss_rng *r, const iss_run_params *rp, const uint8_t *genome,
                        int64_t L, int64_t n_pairs, uint64_t first_ordinal, int64_t pitch, uint8_t *r1_base,
                        uint8_t *r1_qual, uint8_t *r2_base, uint8_t *r2_qual, int64_t *coords /* [n][4] or NULL */,
                        iss_mutation *mut_buf, int64_t mut_cap, int64_t *n_mut, int64_t *n_done) {
    mut_sink sink = {mut_buf, mut_cap, 0, 0};
    int64_t i = 0;
    int rc = ISS_OK;
    uint32_t attempt = 0;
    ev_tab ev[2];
    if (r->mode != ISS_RNG_MT) { ev_tab_build(m, 0, &ev[0]); ev_tab_build(m, 1, &ev[1]); g_ev = ev; }
    while (i < n_pairs) {
        r->ordinal = first_ordinal + (uint64_t)i;
        r->attempt = attempt;
        sink.pair = (int32_t)i;
        int64_t mut_mark = sink.n;
        rc = simulate_read(m, r, rp, genome, L, r1_base + i * pitch, r1_qual + i * pitch, r2_base + i * pitch,
                           r2_qual + i * pitch, coords ? coords + 4 * i : NULL, mut_buf ? &sink : NULL);
        if (rc) break; /* ISS_SKIP_RECORD: the record is abandoned (generator.py:77-80) */
        if (rp->gc_bias) {
            double u = draw_double(r, STREAM_NP, K_PAIR, 0, 0, 3, 1);
            if (u < 0.90) { i++; attempt = 0; }
            else { sink.n = mut_mark; attempt++; } /* `continue`: same i again */
        } else {
            i++;
        }
    }
    if (r->mode != ISS_RNG_MT) { g_ev = NULL; ev_tab_free(&ev[0]); ev_tab_free(&ev[1]); }
    if (n_done) *n_done = i;
    if (n_mut) *n_mut = sink.n;
    return rc;
}

/* ---- function-level entry points (pin the reference's unit goldens) -------- */
int iss_oracle_introduce_indels(const iss_model *m, iss_rng *r, int orientation, const uint8_t *seq, int len,
                                const uint8_t *genome, int64_t L, int64_t start, int64_t end, uint8_t *out) {
    ev_tab ev[2];
    if (r->mode != ISS_RNG_MT) { ev_tab_build(m, 0, &ev[0]); ev_tab_build(m, 1, &ev[1]); g_ev = ev; }
    const int rc = introduce_indels(m, r, orientation, seq, len, genome, L, start, end, out, NULL);
    if (r->mode != ISS_RNG_MT) { g_ev = NULL; ev_tab_free(&ev[0]); ev_tab_free(&ev[1]); }
    return rc;
}
/* the event masks of the read at the rng's current address (position-addressable mode): mask[0 .. RL) */
/* The event process, one draw at a time, for n independent (state, uniform) inputs: next[i] = the new state, slot[i] the slot that
 * fired or -1, mask[i] its event mask (see ev_step).  cur[i] in [-1, ns - 2]. */
int iss_oracle_ev_step(const iss_model *m, int orientation, int64_t n, const int32_t *cur, const uint64_t *m53, const uint64_t *v53,
                       int32_t *next, int32_t *slot, uint8_t *mask) {
    ev_tab ev;
    ev_tab_build(m, orientation, &ev);
    int rc = 0;
    for (int64_t i = 0; i < n; i++) {
        if (cur[i] < -1 || cur[i] > ev.ns - 2 || m53[i] >> 53 || v53[i] >> 53) { rc = 1; break; }
        int sl;
        uint32_t m8;
        next[i] = ev_step(&ev, cur[i], m53[i], v53[i], &sl, &m8);
        slot[i] = sl;
        mask[i] = (uint8_t)m8;
    }
    ev_tab_free(&ev);
    return rc;
}
/* the segment ends of the sampler's tables (E[s] = last slot of slot s's segment), ns = 5 (RL - 1) entries */
int iss_oracle_ev_segments(const iss_model *m, int orientation, int32_t *seg_last) {
    ev_tab ev;
    ev_tab_build(m, orientation, &ev);
    for (int s = 0; s < ev.ns; s++) seg_last[s] = ev.E[s];
    const int ns = ev.ns;
    ev_tab_free(&ev);
    return ns;
}

int iss_oracle_indel_event_masks(const iss_model *m, iss_rng *r, int orientation, uint8_t *mask) {
    if (r->mode == ISS_RNG_MT) return ISS_ERR_KEY;
    ev_tab ev;
    ev_tab_build(m, orientation, &ev);
    sample_indel_events(&ev, r, orientation, mask, m->read_length);
    ev_tab_free(&ev);
    return ISS_OK;
}
void iss_oracle_gen_phred_scores(const iss_model *m, iss_rng *r, int orientation, uint8_t *qual) {
    gen_phred_scores(m, r, orientation, qual);
}
int iss_oracle_mut_sequence(const iss_model *m, iss_rng *r, int orientation, uint8_t *seq, const uint8_t *qual) {
    uint8_t *orig = (uint8_t *)malloc((size_t)m->read_length);
    memcpy(orig, seq, (size_t)m->read_length);
    int rc = mut_sequence(m, r, orientation, seq, qual, orig, NULL);
    free(orig);
    return rc;
}
int64_t iss_oracle_random_insert_size(const iss_model *m, iss_rng *r) {
    double u = draw_double(r, STREAM_NP, K_PAIR, 0, 0, 0, 1);
    return searchsorted_left(m->isize_cdf, m->n_isize, u);
}
uint64_t iss_oracle_py_randbelow(iss_rng *r, uint64_t n) { return py_randbelow(r, n, K_FS); }
int iss_oracle_rev_comp(const uint8_t *in, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; i++) {
        int c = complement_char(in[n - 1 - i]);
        if (!c) return ISS_ERR_KEY;
        out[i] = (uint8_t)c;
    }
    return ISS_OK;
}
