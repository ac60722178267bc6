/*
 * iss_mi355x.h -- C ABI of the MI355X (gfx950) read-generation engine.
 *
 * Drop-in boundary for ONE hot path of InSilicoSeq v2.0.1: the per-read loop of
 * `iss generate`.  Every entry point names the reference
 * interface it replaces (file:line into the reference tree).  Plain pointers and
 * sizes only; no torch / numpy types.  All functions return 0 on success and a
 * negative ISS_E_* code on failure (never exit()); iss_last_error() returns the
 * message of the last failure on that context (or the global one when ctx == NULL).
 *
 * Threading: one context per GPU; calls on one context are serialised by the
 * caller; contexts are independent across threads / processes (the reference's
 * workers share nothing either, iss/app.py:99-106).
 *
 * Arithmetic contract: every f64 comparison of the reference is carried out as an
 * exact integer comparison on 53-bit uniforms m = (w0>>5)*2^26 + (w1>>6) against
 * thresholds prepared on the host (floor/ceil of table*2^53, exact), see DESIGN.md.
 * Uniform words come from Philox4x32 (Salmon et al., SC'11; ten rounds, seven -- the fewest that pass
 * BigCrush -- for the hot digit blocks since ABI 6) addressed by
 * (seed, pair ordinal, attempt, kind, index, sub) -- the address map is part of the
 * contract and is documented in DESIGN.md ("RNG address map").
 */
#ifndef ISS_MI355X_H
#define ISS_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 8: iss_main_kernel (which instantiation of the hot kernel the last Philox-mode call launched: k_main or k_main_g -- the
 *    rows of a group of passes wait in registers for their byte patches).
 * 7: W workers of the reference-identical mode side by side in one context (iss_mt_workers_seed, iss_generate_mt_workers,
 *    iss_mt_workers_peek); 36-bit coordinates in MT mode and the batch arena.
 * 6: indels inside k_main (edit scripts): iss_stats_read reports the scripted reads too; iss_build_id.
 * 5: the ErrorModel methods as batched entries (iss_gen_phred_scores, ...), iss_ev_step.
 * 4: device rows interleaved per pair (whole 128-byte lines per store, see iss_output_reserve), iss_output_row;
 *    Philox address map of the hot draws: three blocks per 16 bases.
 * 3: iss_fastq_compress / iss_deflate_code_build (gzip members built on the device), iss_generate_batch,
 *    iss_fastq_emit_batch (a whole work list per call).  2: iss_fastq_emit / iss_fastq_flush, MT-mode path counters. */
#define ISS_ABI_VERSION 8

#define ISS_E_INVALID (-1)     /* bad argument / model / genome content            */
#define ISS_E_HIP (-2)         /* HIP runtime failure (message has the hip error)  */
#define ISS_E_NOMEM (-3)
#define ISS_E_SHORT_RECORD (-4) /* read_length >= len(record): the reference's AssertionError,
                                   iss/generator.py:130 -> record skipped at :77-80 */
#define ISS_E_IO (-5)

#define ISS_SEQ_METAGENOMICS 0 /* iss/generator.py:134-135, 164-166 */
#define ISS_SEQ_AMPLICON 1     /* iss/generator.py:136-137, 167-169 */

typedef struct iss_ctx iss_ctx;

int iss_abi_version(void);
/* A hash of the kernel sources this library was compiled from (hex string; "unknown" for a build that did not pass
 * -DISS_BUILD_ID).  bench.py ties committed PMC traffic figures (profiles/ *_traffic.json) to the binary that runs. */
const char *iss_build_id(void);

/* One context = one GPU = one reference worker (`cpu_number`), iss/generator.py:223. */
int iss_ctx_create(int device_ordinal, iss_ctx **out);
void iss_ctx_destroy(iss_ctx *ctx);
const char *iss_last_error(const iss_ctx *ctx);

/* Optional: launch on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the context's own stream.  NULL restores the context stream. */
int iss_ctx_set_stream(iss_ctx *ctx, void *hip_stream);

/*
 * Model tables, replaces KDErrorModel.__init__ / load_npz (iss/error_models/kde.py:24-50,
 * iss/error_models/__init__.py:27-50): the 13-key .npz flattened on the host
 * (insilicoseq_amd/model.py) and uploaded once to HBM.  All thresholds are integers in
 * [0, 2^53] (see header comment); orientation 0 = forward, 1 = reverse; base order A,T,C,G.
 */
typedef struct {
    int32_t read_length;
    int32_t n_isize;
    int32_t n_q;                 /* entries per per-position quality CDF (41)              */
    const uint64_t *isize_thr;   /* [n_isize]        floor(cdf*2^53): insert = #(thr <  m)  kde.py:97   */
    const uint64_t *bin_thr;     /* [2][4]           ceil (cdf*2^53): bin    = #(thr <= m)  kde.py:74   */
    const uint8_t *bin_nonempty; /* [2][4]                                                  kde.py:80   */
    const uint64_t *q_thr;       /* [2][4][RL][n_q]  floor: phred = #(thr < m)              kde.py:84   */
    const uint64_t *subst_thr;   /* [2][RL][4][3]    ceil : alt index = #(thr <= m)  __init__.py:95-97 */
    const uint8_t *subst_alt;    /* [2][RL][4][3]    ASCII alternatives                                */
    const uint64_t *ins_thr;     /* [2][RL][4]       ceil : insert iff m < thr       __init__.py:194   */
    const uint8_t *ins_letter;   /* [2][RL][4]       ASCII, in the dict's iteration order              */
    const uint64_t *del_thr;     /* [2][RL][4]       ceil : delete iff m < thr       __init__.py:209   */
    const uint64_t *mut_thr;     /* [n_q+1]          floor(phred_to_prob(q)*2^53): error iff m > thr   */
    /* BasicErrorModel (iss/error_models/basic.py:18-63), quality_mode 1: constant insert size (no draw), phred
     * scores int(round(-10 * log10(1 - min(np.random.normal(basic_mean, basic_sd), basic_cap)))) per position
     * (n_q + 1 must cover the phreds: 41 -> 0..41).  The reference-compatible mode (iss_generate_mt) draws the
     * normal deviates in the reference's order; iss_generate / iss_generate_batch invert q_thr like a KDE row, so
     * the caller puts the distribution of that score there (the same row at every position, in every bin;
     * insilicoseq_amd/model.py basic_phred_cdf).  The insert-size and bin tables are unused. */
    int32_t quality_mode;        /* 0: KDE tables (kde.py), 1: basic                                     */
    int32_t basic_insert_size;   /* basic.py:21 (200)                                                    */
    double basic_mean;           /* util.phred_to_prob(30), basic.py:24, :52                             */
    double basic_sd;             /* 0.01, basic.py:52                                                     */
    double basic_cap;            /* 0.9999, basic.py:52                                                   */
} iss_model_tables;

int iss_model_upload(iss_ctx *ctx, const iss_model_tables *tables);

/*
 * Genome upload, replaces handing `record.seq` to the worker (iss/generator.py:116,
 * pickled through iss/app.py:99-106).  ASCII in, validated against the reference's
 * rev_comp alphabet (iss/util.py:57-88; any other letter is the reference's KeyError and is
 * rejected here with ISS_E_INVALID); stored in HBM as 2-bit codes (A,T,C,G = 0..3) + a
 * 1-bit "exception" mask (IUPAC / lower case) + the ASCII copy the exceptions are read from.
 * 1 <= length <= 2^34 - 4096 (ABI 6: the reference takes records of 2^31 - 1 bases and more through its memmap
 * spill, iss/generator.py:313-331; ABI 7: iss_generate_mt takes them too -- 36-bit coordinates, two stream words per
 * `randrange` candidate once the bound passes 2^32, as CPython's getrandbits).
 */
int iss_genome_upload(iss_ctx *ctx, const uint8_t *ascii, int64_t length, int32_t *genome_id);
/* The same for a record of plain A/C/G/T handed over as 2-bit codes (16 bases per little-endian 32-bit word, base i in
 * bits 2*(i%16).., A,T,C,G = 0..3), from host memory or -- codes_on_device != 0 -- from device memory of this GPU: what a
 * rank of a multi-GPU run receives in the one RCCL broadcast of the packed genomes (no ASCII, no host bounce). */
int iss_genome_upload_packed(iss_ctx *ctx, const uint32_t *codes, int64_t length, int32_t codes_on_device, int32_t *genome_id);
int iss_genome_clear(iss_ctx *ctx);

/* Device output rows (R1 bases, R1 phred, R2 bases, R2 phred).  Reserve before generating.
 * On the HOST (iss_output_download, iss_fastq_write) they are four arrays [pairs][pitch],
 * pitch = 8*ceil(read_length/8) (ask iss_output_pitch).  On the DEVICE a pair owns one row of
 * row = 128*ceil(pitch/32) bytes; the 32 positions 32l..32l+31 are line l (128 bytes) of the row: bytes 0-63
 * mate 1, 64-127 mate 2, each four 16-byte pieces [8 bases][8 phreds], i.e. position p of array k
 * (0 R1 bases, 1 R1 phred, 2 R2 bases, 3 R2 phred) of pair i is byte
 * i*row + 128*(p/32) + 64*(k/2) + 16*((p/8)%4) + 8*(k%2) + p%8 (the kernel writes whole lines this way). */
int iss_output_reserve(iss_ctx *ctx, int64_t capacity_pairs);
int iss_output_pitch(const iss_ctx *ctx);
/* raw device pointers (for a caller that consumes the reads on the GPU / benchmarks): byte 0 of
 * each array's part of row 0 (layout above); iss_output_row = row */
int iss_output_device_ptrs(const iss_ctx *ctx, void **r1_base, void **r1_qual, void **r2_base, void **r2_qual);
int iss_output_row(const iss_ctx *ctx);

/*
 * The hot path: n_pairs read pairs from one record, replaces
 *   reads_generator + simulate_read            iss/generator.py:69-95, 98-192
 *   ErrorModel.introduce_indels / adjust_seq_length / introduce_error_scores / mut_sequence
 *                                              iss/error_models/__init__.py:158-228, 114-156, 52-67, 69-112
 *   KDErrorModel.gen_phred_scores / random_insert_size   iss/error_models/kde.py:52-98
 * Asynchronous on the context stream.  Pair k of the call is written to row
 * (out_first_pair + k) and draws its uniforms at Philox address (seed, first_ordinal + k).
 * `seed` is the worker seed (reference: seed + cpu_number, iss/generator.py:234-236).
 * Returns ISS_E_SHORT_RECORD (nothing launched) when read_length >= genome length.
 */
int iss_generate(iss_ctx *ctx, int32_t genome_id, int64_t n_pairs, uint64_t first_ordinal, uint64_t seed,
                 int32_t sequence_type, int32_t gc_bias, int64_t out_first_pair);

/*
 * The same for a whole work list in ONE set of launches (reads_generator over several work items,
 * iss/generator.py:69-95 + the loop of worker_iterator, :245-249): item k = (genome_ids[k], n_pairs[k]); its pairs take
 * the ordinals and output rows that n_items consecutive iss_generate calls would give them (first_ordinal /
 * out_first_pair onwards), so the rows are identical to those calls' -- without a kernel launch sequence per record.
 * The records are laid side by side in one device arena (kept until a call names another list of records) and pair
 * descriptors carry arena coordinates (36-bit like a single record's: the records of one call may hold 2^34 - 4096 bases
 * together, ISS_E_INVALID beyond -- ABI 7; 2^31 until then); iss_output_download_coords returns record coordinates as before.  Custom
 * fragment lengths (iss_set_fragment) apply as in iss_generate.  A record not longer than the read
 * length: ISS_E_SHORT_RECORD, nothing generated (leave such records out, as reads_generator skips them).
 */
int iss_generate_batch(iss_ctx *ctx, int32_t n_items, const int32_t *genome_ids, const int64_t *n_pairs,
                       uint64_t first_ordinal, uint64_t seed, int32_t sequence_type, int32_t gc_bias,
                       int64_t out_first_pair);

/* Custom fragment length on the Philox path (--fragment-length / --fragment-length-sd, iss/generator.py:121-123):
 * fragment = int(mu + sd * gaussian), each pair running its own polar Box-Muller loop on its K_FRAG uniforms
 * (nothing is cached across pairs).  Negative inserts, templates cut by the genome end and Python's slice rules
 * are honoured.  Values within 1e-6 of an integer are re-evaluated by the host with libm. */
int iss_set_fragment(iss_ctx *ctx, int32_t enabled, double fragment_length, double fragment_sd);

int iss_synchronize(iss_ctx *ctx);

/*
 * The inner plugin surface: the per-read methods of the reference's ErrorModel duck type, batched (n reads per call, read i
 * drawing at Philox ordinal first_ordinal + i of the worker stream `seed`, attempt 0; same address map as iss_generate, so a
 * read generated by iss_generate at that ordinal saw exactly these draws).  Host arrays in and out, rows of read_length bytes.
 *   iss_gen_phred_scores   KDErrorModel.gen_phred_scores(cdfs, orientation)      iss/error_models/kde.py:52-86
 *                          (through ErrorModel.introduce_error_scores, iss/error_models/__init__.py:52-67)
 *   iss_mut_sequence       ErrorModel.mut_sequence(record, orientation)           iss/error_models/__init__.py:69-112
 *                          seq in place (ASCII), quality = its phred scores; status[i] = 2: a letter the substitution table
 *                          does not hold (the reference's KeyError)
 *   iss_random_insert_size KDErrorModel.random_insert_size()                      iss/error_models/kde.py:88-98
 *   iss_introduce_indels   ErrorModel.introduce_indels(record, orientation, full_seq, bounds) incl. adjust_seq_length
 *                          iss/error_models/__init__.py:158-228, 114-156.  seq: n rows of read_length bytes holding seq_len[i]
 *                          letters in read direction; bounds: n x (read_start, read_end) in full_seq; status[i]: 0, 2 KeyError,
 *                          3 IndexError
 * orientation: 0 "forward", 1 "reverse".  Kept out of the hot path: one lane per read, exact 53-bit comparisons.
 */
int iss_gen_phred_scores(iss_ctx *ctx, int32_t orientation, int64_t n, uint64_t first_ordinal, uint64_t seed, uint8_t *quality);
int iss_mut_sequence(iss_ctx *ctx, int32_t orientation, int64_t n, uint64_t first_ordinal, uint64_t seed, uint8_t *seq,
                     const uint8_t *quality, int32_t *status);
int iss_random_insert_size(iss_ctx *ctx, int64_t n, uint64_t first_ordinal, uint64_t seed, int64_t *insert_size);
int iss_introduce_indels(iss_ctx *ctx, int32_t orientation, int64_t n, uint64_t first_ordinal, uint64_t seed, const uint8_t *seq,
                         const int32_t *seq_len, const uint8_t *full_seq, int64_t full_len, const int64_t *bounds, uint8_t *out,
                         int32_t *status);
/* One draw of the Philox path's indel event process, the sampler behind the tests `random() < p` of introduce_indels
 * (iss/error_models/__init__.py:193-196, :209), for n independent inputs: state cur[i] (the last test
 * slot decided, -1 .. 5 (read_length - 1) - 2), numerator m53[i] of the draw's uniform and v53[i] of the deletion sub-draw
 * (< 2^53) -> next[i] (new state), slot[i] (the test that fires, or -1: none in the rest of the state's segment), mask[i]
 * (bits 0-3: insertion of letter slot x, bits 4-7: the deletion fires if the base is A, T, C, G).  A test hook: the kernels
 * loop over the same device function; the CPU tests pin its CPU twin to the reference's probabilities. */
int iss_ev_step(iss_ctx *ctx, int32_t orientation, int64_t n, const int32_t *cur, const uint64_t *m53, const uint64_t *v53,
                int32_t *next, int32_t *slot, uint8_t *mask);

/* --store_mutations: one VCF row of the reference (iss/error_models/__init__.py:98-108, 197-221, written by
 * write_mutations, iss/generator.py:598-620). */
typedef struct {
    int32_t pair;     /* pair index i within the call (read id {record.id}_{i}_{cpu}/{mate+1}) */
    int8_t mate;      /* 0 forward, 1 reverse */
    int8_t type;      /* 0 substitution, 1 insertion (VCF alt = ref + alt), 2 deletion (alt '.') */
    int16_t position; /* 0-based (VCF POS = position + 1) */
    uint8_t ref, alt; /* ASCII */
    int16_t quality;  /* phred for substitutions, -1 ('.') otherwise */
} iss_mutation;
/* Philox path: after iss_mutations_reserve(capacity > 0) iss_generate records the rows of each call (capacity
 * counts 256-row reservation chunks per wavefront, so reserve generously: ~2 rows per expected mutation + 64 k);
 * iss_mutations_download waits for the call, drops the rows of reads the indel fix-up rebuilt, sorts into the
 * reference's order (pair, mate, indel rows in loop order, substitution rows by position) and returns them.
 * ISS_E_NOMEM when the buffer was too small for the call; *n_rows is then the number of row slots the call asked for. */
int iss_mutations_reserve(iss_ctx *ctx, int64_t capacity);
int iss_mutations_download(iss_ctx *ctx, iss_mutation *out, int64_t capacity, int64_t *n_rows);

/* Copy rows [first_pair, first_pair + n_pairs) to host arrays of pitch iss_output_pitch(). */
int iss_output_download(iss_ctx *ctx, int64_t first_pair, int64_t n_pairs, uint8_t *r1_base, uint8_t *r1_qual,
                        uint8_t *r2_base, uint8_t *r2_qual);

/* Per-pair coordinates (forward_start, reverse_start, reverse_end, insert_size) of rows
 * [first_pair, first_pair+n_pairs) as int64[n][4] -- iss/generator.py:135, 165-176. */
int iss_output_download_coords(iss_ctx *ctx, int64_t first_pair, int64_t n_pairs, int64_t *coords);

/* HIP-event timing of the kernels launched by iss_generate (on the launch stream).
 * enable: 0 off, 1 every kernel, 2 k_main only (an event is a bubble in the stream: the five-kernel timing costs
 * about 6 % of a step).  iss_timing_read synchronises, returns accumulated milliseconds per kernel
 * (setup, main, indel-scan, indel-fixup) and the number of iss_generate launches, then
 * resets the accumulators. */
int iss_timing_enable(iss_ctx *ctx, int enable);
int iss_timing_read(iss_ctx *ctx, double ms[4], int64_t *n_launches);

/* Counters since the last read: reads rebuilt by the indel fix-up kernel (one wavefront per read), reads k_main built from an
 * edit script (models whose reads often have indels).  Either pointer may be NULL. */
int iss_stats_read(iss_ctx *ctx, int64_t *n_fixup_reads, int64_t *n_scripted_reads);

/* Name of the kernel the last iss_generate / iss_generate_batch call launched for the hot path (simulate_read's per-base work,
 * iss/generator.py:146-180 + iss/error_models/kde.py:52-86): "k_main<mutations, plain, indel>" or "k_main_g<NI, NP>" -- what a
 * profile of the call lists, so that a measurement can name what it measured.  Returns the name's length (it is cut to
 * capacity - 1 characters and always closed by a NUL), 0 before the first call. */
int iss_main_kernel(iss_ctx *ctx, char *name, int capacity);

/*
 * Reference-compatible RNG mode (sequential; for bit-identity with the reference, not throughput).
 * iss_mt_seed == `random.seed(seed); np.random.seed(seed)` of worker_iterator (iss/generator.py:234-236,
 * seed = args.seed + cpu_number, must be < 2^32 like numpy's legacy seeding).  iss_generate_mt consumes the
 * two MT19937 streams exactly as reads_generator/simulate_read do (iss/generator.py:69-192), so rows
 * [out_first_pair, +*n_done) equal the reference's reads for that worker byte for byte; the stream state
 * carries over to the next call (the next work item of the worker).  On ISS_E_SHORT_RECORD the one numpy
 * double the reference draws before its assertion is consumed too.  iss_mt_peek copies the next n <= 624 words of both streams (not consumed).
 */
int iss_mt_seed(iss_ctx *ctx, uint64_t seed);
int iss_generate_mt(iss_ctx *ctx, int32_t genome_id, int64_t n_pairs, int32_t sequence_type, int32_t gc_bias,
                    int64_t out_first_pair, int64_t *n_done);
int iss_mt_peek(iss_ctx *ctx, uint32_t *py_words, uint32_t *np_words, int32_t n);
/* Pairs produced so far by each device path of iss_generate_mt: the offset resolver + parallel emitter
 * (plain pairs) and the sequential walker (pairs with an indel candidate, letters outside ACGT or a template cut by
 * a genome end; indel-heavy models, the BasicErrorModel; everything when ISS_MT_PATH=walk is set in the environment). */
int iss_mt_path_counts(iss_ctx *ctx, int64_t *n_resolved, int64_t *n_walked);
/*
 * W reference workers side by side in ONE context (ABI 7).  The reference's own parallelism is N workers, each a sequential
 * chain over its two MT19937 streams seeded `seed + cpu_number` (pool.starmap over worker_iterator: iss/app.py:99-106,
 * iss/generator.py:234-236).  iss_mt_workers_seed(ctx, W, seeds) == W times iss_mt_seed(seeds[w]);
 * iss_generate_mt_workers == for every worker w: iss_generate_mt(genome_ids[w], n_pairs[w], ..., out_first_pair[w]) on ITS streams
 * -- same rows, same stream positions afterwards -- but every kernel of the path is launched once for all workers, one workgroup
 * per worker (n_pairs[w] == 0: the worker sits this call out).  status[w] (may be NULL): 0, or ISS_E_SHORT_RECORD for a worker whose
 * record is not longer than a read (its draw is consumed like iss_generate_mt does); any other failure fails the call.  The
 * workers' row ranges must not overlap.  Custom fragment lengths and the BasicErrorModel run the workers one after the other
 * through iss_generate_mt (draws the host's libm settles); --store_mutations rows are per context (ISS_E_INVALID here).
 * iss_mt_workers_peek: iss_mt_peek for worker w.  iss_mt_path_counts counts the set's pairs too.
 */
int iss_mt_workers_seed(iss_ctx *ctx, int32_t n_workers, const uint64_t *seeds);
int iss_generate_mt_workers(iss_ctx *ctx, int32_t n_workers, const int32_t *genome_ids, const int64_t *n_pairs,
                            const int64_t *out_first_pair, int32_t sequence_type, int32_t gc_bias, int64_t *n_done, int32_t *status);
int iss_mt_workers_peek(iss_ctx *ctx, int32_t worker, uint32_t *py_words, uint32_t *np_words, int32_t n);
/* Custom fragment length in MT mode (--fragment-length / --fragment-length-sd, iss/generator.py:121-123):
 * fragment = int(np.random.normal(mu, sd)) with numpy's legacy polar Box-Muller incl. its cached second value
 * (reset by iss_mt_seed like np.random.seed does).  The device evaluates it; draws that land within 1e-6 of an
 * integer are re-evaluated on the host with libm so the truncation provably equals numpy's. */
int iss_mt_set_fragment(iss_ctx *ctx, int32_t enabled, double fragment_length, double fragment_sd);

/* --store_mutations in MT mode: after iss_mt_mutations_reserve(capacity > 0) every iss_generate_mt call records
 * its rows (iss_mutation, in the reference's order) from index 0; iss_mt_mutations_download copies
 * min(rows, capacity) of them and reports the total row count. */
int iss_mt_mutations_reserve(iss_ctx *ctx, int64_t capacity);
int iss_mt_mutations_download(iss_ctx *ctx, iss_mutation *out, int64_t capacity, int64_t *n_total);

/*
 * FASTQ text built ON THE DEVICE (the host formatter below tops out near 3 GB/s of text, four hundred times
 * under the kernel's rate): rows [first_pair, +n_pairs) of the output buffers become the records of
 * simulate_reads' SeqIO.write calls (iss/generator.py:64-65, ids per :150, 181, i from first_i) -- one wavefront
 * per record, byte offsets in closed form -- and are appended to fd_r1 / fd_r2 at their current positions.
 * Asynchronous: the format kernel runs on the context's stream behind the generation it reads, the copy to
 * pinned host memory on a copy stream, the file writes (pwrite, n_threads pieces per file) on a writer thread,
 * so the next batch can be generated meanwhile (two text slots).  iss_fastq_flush waits until every queued
 * byte is in the files and leaves the descriptors positioned at the end; call it before using the files.
 */
int iss_fastq_emit(iss_ctx *ctx, int fd_r1, int fd_r2, const char *record_id, int64_t first_i, int32_t cpu_number,
                   int64_t first_pair, int64_t n_pairs, int32_t n_threads);
/* The same for the rows of several work items in ONE text job (one format launch, one pair of copies, one write per
 * file): item k = rows [first_pair[k], +n_pairs[k]) under record_ids[k] with pair ids from first_i[k].  The bytes
 * appended are those of n_items iss_fastq_emit calls in item order (compressed mode: one gzip member for the call). */
int iss_fastq_emit_batch(iss_ctx *ctx, int fd_r1, int fd_r2, int32_t n_items, const char *const *record_ids,
                         const int64_t *first_i, const int64_t *first_pair, const int64_t *n_pairs, int32_t cpu_number);
/* The same for items that belong to DIFFERENT workers and go to DIFFERENT places of the two files (text mode only): item k is
 * worker cpu_numbers[k]'s ("{id}_{i}_{cpu_numbers[k]}/1") and its text is written at byte file_off[k] of fd_r1 and of fd_r2 -- the
 * two files of a pair hold records of equal lengths.  The reference's parent concatenates its workers' temp files in worker order
 * (iss/app.py:123-127, iss/util.py:213-234); a worker's text size is arithmetic (constant read length, decimal pair numbers), so
 * the W workers of a set (iss_generate_mt_workers) write straight into the final files and the concatenation -- a second copy of
 * every byte under the destination's inode lock -- disappears.  The files' running offsets (iss_fastq_emit) are not moved; the
 * caller sizes the files (ftruncate) and owns the layout. */
int iss_fastq_emit_scatter(iss_ctx *ctx, int fd_r1, int fd_r2, int32_t n_items, const char *const *record_ids, const int64_t *first_i,
                           const int64_t *first_pair, const int64_t *n_pairs, const int32_t *cpu_numbers, const int64_t *file_off,
                           int32_t n_threads);
int iss_fastq_flush(iss_ctx *ctx);

/*
 * `--compress` (iss/app.py:134-143 -> util.compress, iss/util.py:255-268: gzip of the finished FASTQ files) moved in
 * front of the files: with mode 1 every iss_fastq_emit appends ONE GZIP MEMBER per file (RFC 1952) holding the
 * batch's text instead of the text -- DEFLATE blocks with a dynamic Huffman code, run and previous-record matches, built on the device
 * from the batch's own token histogram, so only the compressed bytes (about 1/3.5) cross PCIe and reach the file
 * system.  Concatenated members are one valid .gz file whose content is the text mode 0 writes.  mode 0 (default):
 * plain text.  Changing the mode flushes.
 */
int iss_fastq_compress(iss_ctx *ctx, int32_t mode);

/*
 * The code builder of the compressed mode as a host function (tests, tools): hist[273] token counts (literals
 * 0..255, [256] the number of blocks, [257..272] the length codes of matches of 3..32 bytes) and the batch's record
 * length (the distance of the "previous record" matches; 0: only runs, distance 1) -> entry[s] = bit-reversed code |
 * length << 16 for the 273 symbols, the dynamic-block header (BFINAL = 0 ... both code-length tables) as hdr_bits
 * bits, least significant first, in hdr_words[64], and dist_code[3] = {distance symbol, extra bits, their value} of
 * the record length: a run's length code (+ its extra bits) is followed by the bit 0, a previous-record match's by the
 * bit 1 and the distance's extra bits.  No GPU needed.
 */
int iss_deflate_code_build(const uint32_t *hist, uint32_t record_distance, uint32_t *entry, uint32_t *hdr_bits,
                           uint32_t *hdr_words, uint32_t *dist_code);

/*
 * FASTQ emission, replaces SeqIO.write(record, handle, "fastq-sanger") in
 * simulate_reads (iss/generator.py:64-65): records "@{record_id}_{i}_{cpu_number}/{1|2}\n
 * SEQ\n+\nQUAL\n" with QUAL = chr(33+q), ids per iss/generator.py:150, 181; i runs from
 * first_i.  Host-side formatter (multi-threaded) appending to two file descriptors.
 */
int iss_fastq_write(int fd_r1, int fd_r2, const char *record_id, int64_t first_i, int32_t cpu_number,
                    int64_t n_pairs, int32_t read_length, int32_t pitch, const uint8_t *r1_base,
                    const uint8_t *r1_qual, const uint8_t *r2_base, const uint8_t *r2_qual, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* ISS_MI355X_H */
