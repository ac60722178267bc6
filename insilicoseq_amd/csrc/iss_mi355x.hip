// iss_mi355x.hip -- C-ABI shared library of the MI355X read-generation engine (see include/iss_mi355x.h).
// Host side: context, HBM uploads (model tables, genomes; small records from an arena), launch sequencing on one HIP
// stream for one record (iss_generate) or a whole work list (iss_generate_batch: records side by side in one arena),
// HIP-event timing, downloads, the FASTQ pipeline (text or gzip members built on the device, copy stream, writer thread).
// Device side: iss_kernels.hip.h (the Philox path), iss_mt_compat.hip.h (the reference's Mersenne-Twister streams),
// iss_fastq.hip.h, iss_deflate.hip.h.
#include "iss_mi355x.h"

#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "iss_kernels.hip.h"
#include "iss_fastq.hip.h"
#include "iss_deflate.hip.h"
#include "iss_mt_compat.hip.h"
#include "iss_units.hip.h"

namespace {

thread_local std::string g_last_error;

struct Genome {
    uint32_t *packed_alloc = nullptr, *mask_alloc = nullptr;  // allocations (one leading pad word)
    uint32_t *packed = nullptr;
    uint32_t *mask = nullptr;
    uint8_t *ascii = nullptr;
    int64_t L = 0;
    bool has_exceptions = false;
    bool in_arena = false;  // small record: its three buffers are slices of a GenomeArena slab
};

// Records of a long work list (draft genomes: thousands of contigs) are small: their buffers are cut from slabs
// instead of three hipMallocs each, and their letters are checked on the host instead of waiting for the pack kernel.
constexpr size_t PK_PAD = 12;  // padding words of a packed genome: one in front (windows start a word early), the rest behind (the
                               // 16-byte window loads of k_indel_script may reach a few words past a read's window)
constexpr int64_t SMALL_RECORD = 1 << 20;
constexpr size_t ARENA_SLAB = 64u << 20;
struct GenomeArena {
    std::vector<uint8_t *> slabs;
    size_t used = ARENA_SLAB;  // of the last slab
    uint8_t *take(size_t bytes, hipError_t *err) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (used + bytes > ARENA_SLAB) {
            void *p = nullptr;
            *err = hipMalloc(&p, ARENA_SLAB);
            if (*err != hipSuccess) return nullptr;
            slabs.push_back(static_cast<uint8_t *>(p));
            used = 0;
        }
        uint8_t *r = slabs.back() + used;
        used += bytes;
        return r;
    }
    void clear() {
        for (auto *p : slabs) (void)hipFree(p);
        slabs.clear();
        used = ARENA_SLAB;
    }
};

// 0: outside util.rev_comp's alphabet (iss/util.py:57-88), 1: plain A/C/G/T, 2: IUPAC or lower case (an "exception")
inline int letter_class(uint8_t c) {
    if (c == 'A' || c == 'T' || c == 'C' || c == 'G') return 1;
    const bool letter = (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
    const uint8_t u = c & ~0x20u;
    const bool ok = letter && (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'Y' || u == 'R' || u == 'W' || u == 'S' ||
                               u == 'K' || u == 'M' || u == 'N' || u == 'B' || u == 'V' || u == 'D' || u == 'H');
    return ok ? 2 : 0;
}

constexpr int FIX_SLOTS_C = 16;  // (= FIX_SLOTS below)
struct TimedLaunch {
    // ev0 setup [ev3 scan + script ev4] ev1 main ev2 ev5 fixup ev6;  ev7: the end of the setup-stream kernels (k_setup, and for
    // models with frequent indels k_indel_scan + k_indel_script) when they run beside the previous call's kernels
    hipEvent_t ev[8];
    bool has_scan;
    bool scan_first;  // the scan stands between k_setup and k_main on one stream (ev0 setup ev3 scan ev4 = ev1 main ev2)
};

constexpr int FIX_SLOTS = FIX_SLOTS_C;  // ring of fix-list / read-list counters (one per chunk in flight)

// Device-formatted FASTQ on its way to the files: two slots of (device text, pinned host text) per mate; the
// format kernel runs on the context's stream, the copy back on a copy stream, the file writes on a writer thread.
struct FastqJob {
    int slot;
    size_t bytes;     // text bytes per file
    int fd[2];
    int64_t off[2];   // plain text: final offsets of this job's bytes (compressed: the writer keeps the running offsets)
    int threads;
    bool gzip;
    uint32_t n_blocks;
    std::vector<uint64_t> item_off;  // text offsets of the job's work items (the writer checks the record structure there)
    std::vector<int64_t> item_file_off;  // iss_fastq_emit_scatter: where each item's text goes in BOTH files (empty: the job is one piece at `off`)
};
struct FastqPipe {
    bool ready = false;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_fmt[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
    uint8_t *d_text[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [slot][mate]
    uint8_t *h_text[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // per slot: the item table and the record ids of the emit call (pinned host copy + device copy)
    iss::FastqItem *h_items[2] = {nullptr, nullptr}, *d_items[2] = {nullptr, nullptr};
    char *h_ids[2] = {nullptr, nullptr}, *d_ids[2] = {nullptr, nullptr};
    size_t items_cap[2] = {0, 0}, ids_cap[2] = {0, 0};
    size_t cap = 0;
    int next = 0;
    int fd[2] = {-1, -1};
    // Offsets of the next byte of each file.  ONLY touched with `mu` held once the writer thread runs: in text mode the
    // caller advances them when it queues a job, in compressed mode the writer does when it knows a member's size
    // (round 2 advanced them outside the lock in text mode while the writer added its -- zero -- byte count under it:
    // a lost update there made the next job overwrite the previous one's bytes).
    int64_t off[2] = {0, 0};
    int64_t attached_off[2] = {0, 0};  // offsets when the files were attached ...
    int64_t accounted[2] = {0, 0};     // ... and the bytes queued (text) / written (gzip) since: off == attached_off + accounted
    std::thread writer;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<FastqJob> jobs;
    bool busy[2] = {false, false};
    bool stop = false;
    std::string error;
    // compressed mode (iss_fastq_compress): per slot and mate the device-side state of iss_deflate.hip.h, the
    // compressed bytes land in h_text; the writer thread fetches exactly the bytes a member has
    int gzip = 0;
    hipStream_t data_stream = nullptr;
    size_t comp_cap = 0;                 // bytes of d_comp / h_text per (slot, mate)
    uint32_t blocks_cap = 0;
    uint8_t *d_comp[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint32_t *d_hist[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    iss::DeflateCode *d_code[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint32_t *d_bbytes[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *d_bcrc[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint64_t *d_boff[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint32_t *h_bcrc[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // pinned
    uint64_t *h_total[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // pinned, one value
    uint32_t op_block[32];               // CRC operator "append DEFLATE_BLOCK zero bytes"
};
constexpr size_t FASTQ_ID_MAX = 4096;

}  // namespace

struct iss_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // setup + main kernels
    hipStream_t indel_stream = nullptr;  // the second k_indel_script launch of a heavy model's step, beside the first
    // MT mode: the stream words are produced here, one turn ahead of their consumption.  LOWEST priority: its hardware queue then
    // comes from another pool than the main stream's (as the setup stream's does, at the highest).  Streams of one priority share
    // four hardware queues, handed out as the streams are first used: with another engine and torch's streams alive in the process
    // (bench.py) the fill stream and the main stream of an MT-mode engine sat on ONE queue, fill and resolver ran one after the
    // other and a worker made 2.3e5 pairs/s instead of 3.8e5 (round 4's "2.2e5 in the bench line, 3.7e5 by itself";
    // tools/mt_context_probe2.py: 2.31e5 -> 3.84e5 with GPU_MAX_HW_QUEUES=8, and with this priority without the variable).
    hipStream_t fill_stream = nullptr;
    // k_setup of a call runs on its own stream, beside the kernels of the call (or chunk) before: it reads nothing they
    // write, and what it writes -- descriptors, flags, the fix-up list -- is double-buffered by call parity (`desc`, `flags`,
    // `fix_list` below point at the current call's set).  ISS_SETUP_AHEAD=0: everything in order on one stream.
    hipStream_t setup_stream = nullptr;
    // The worker set's emitter (k_mt_emit_w: 0.5 TB/s of reads over the whole chip, beside the NEXT turn's resolver): a stream of
    // its own at the lowest priority.  On the setup stream (highest priority) it took the resolvers' issue slots -- the chain the
    // turn waits for: 1.79 -> 1.90e7 pairs/s at W = 64, 4.70 -> 5.04e7 at W = 256.  A stream bound to a subset of the CUs
    // (hipExtStreamCreateWithCUMask, 32 .. 128 CUs) was worse than either: the emitter needs the chip (1.2 -> 2.1e7 at W = 64).
    hipStream_t emit_stream = nullptr;
    bool setup_ahead = true;
    iss::PairDesc *desc_buf[2] = {nullptr, nullptr};
    uint32_t *flags_buf[2] = {nullptr, nullptr}, *fixl_buf[2] = {nullptr, nullptr};
    hipEvent_t ev_call_done[2] = {nullptr, nullptr};  // the last kernel of the last call that used the set
    bool ev_call_valid[2] = {false, false};
    hipEvent_t ev_setup_done[FIX_SLOTS_C] = {};         // k_setup of a chunk -> its k_main (ring, like the counters)
    hipEvent_t ev_fork[FIX_SLOTS_C] = {}, ev_join[FIX_SLOTS_C] = {};  // the two k_indel_script launches of a chunk side by side
    hipEvent_t ev_slot_done[FIX_SLOTS_C] = {};          // the last kernel of the chunk that used a counter slot: the setup stream waits
    bool ev_slot_valid[FIX_SLOTS_C] = {};               //   for it before the slot's next user clears the counters
    hipEvent_t ev_inputs = nullptr;                     // tables / arena copies queued on the main stream for this call's k_setup
    uint64_t call_seq = 0;
    bool inputs_pending = false;  // copies for this call's k_setup were queued on the main stream (ev_inputs)
    bool timing_all = false;      // HIP events around every kernel: one stream
    GenomeArena arena;
    // iss_generate_batch: the records of the last batch copied side by side into one arena (ids + items cached)
    std::vector<int32_t> comm_ids;
    std::vector<iss::BatchItem> comm_items;
    uint32_t *comm_packed = nullptr, *comm_mask = nullptr;
    uint8_t *comm_ascii = nullptr;
    iss::BatchItem *d_items[2] = {nullptr, nullptr}, *h_items[2] = {nullptr, nullptr};  // device / pinned host, two sets:
    int64_t *d_item_first[2] = {nullptr, nullptr}, *h_item_first[2] = {nullptr, nullptr};  // a call's launches may still read
    hipEvent_t ev_items[2] = {nullptr, nullptr};                                          // its set while the next is filled
    size_t d_items_cap = 0;
    uint64_t batch_seq = 0;
    bool comm_exceptions = false;
    int64_t comm_cap = 0;  // bases the arena buffers hold
    uint64_t chunk_seq = 0;
    std::string last_error;
    // model
    bool have_model = false;
    iss::DevModel M{};
    std::vector<void *> model_allocs;
    // genomes
    std::vector<Genome> genomes;
    // outputs
    int64_t capacity = 0;
    uint8_t *out[4] = {nullptr, nullptr, nullptr, nullptr};  // ONE allocation of interleaved rows (iss::xp): out[k] = out[0] + iss::row_array_off(k)
    uint8_t *d_stage = nullptr;  // iss_output_download: the four plain arrays of the rows being copied
    size_t stage_cap = 0;
    iss::PairDesc *desc = nullptr;
    uint32_t *flags = nullptr;
    uint32_t *fix_list = nullptr;
    uint32_t *fix_count = nullptr;  // one counter per launch chunk is reset in-stream
    // indel events (k_indel_scan -> k_indel_script -> k_main), per row; two sets for the models whose scan runs
    // on the setup stream, beside the kernels of the call before (otherwise [1] aliases [0])
    uint32_t *ev_count[2] = {nullptr, nullptr}, *ev_list[2] = {nullptr, nullptr};
    uint4 *read_list[2] = {nullptr, nullptr};
    uint2 *read_list1[2] = {nullptr, nullptr};  // (the reads with one event step: RunArgs::read_list1)
    uint32_t *read_count = nullptr;  // FIX_SLOTS x 2 x SCAN_MAX_WGS segment lengths of the two read lists, like fix_count
    // models with frequent indels: the edit scripts of the reads with an event (k_indel_script -> k_main), DevModel::sc_stride
    // bytes per read, two sets like the event lists
    uint8_t *script[2] = {nullptr, nullptr};
    double light_below = 2e-3;  // ISS_LIGHT_INDELS (read once, at iss_ctx_create): models whose reads have an event less often are "light"
    int env_tiles = 0, env_guide_bits = 0;  // ISS_TILES / ISS_GUIDE_BITS: tuning aids of the tile sweeps (0: the cost model decides)
    bool debug_model = false;               // ISS_DEBUG_MODEL
    int64_t env_chunk_pairs = 0;            // ISS_CHUNK_PAIRS: pairs per launch chunk at most (tests: a call of many chunks)
    int env_main_wgs = 0;                   // ISS_MAIN_WGS: workgroups of k_main / k_main_g at most (tests: many passes per workgroup from few pairs)
    int env_group = -1, env_group_min = 0;  // ISS_MAIN_GROUP: passes per group of k_main_g (0: k_main; unset: chosen per model); ISS_MAIN_GROUP_MIN: min_round
    double mt_guard = 1e-6;                 // ISS_MT_GUARD: how close to a rounding boundary the device still decides (tests widen it)
    bool light = false;  // reads with an indel are rare (< ISS_LIGHT_INDELS of the reads, default 2e-3): all of them take k_indel_fixup
    double mt_bounce_rate = 0;  // MT mode: expected indel candidates per pair (decides resolver vs. sequential walker)
    // custom fragment length on the Philox path
    bool has_frag = false;
    double frag_mu = 0, frag_sd = 0;
    iss::FragAmb *d_amb = nullptr;
    uint32_t *d_amb_count = nullptr;
    uint32_t *d_ov_pairs = nullptr;
    int64_t *d_ov_frags = nullptr;
    int64_t amb_cap = 0;
    // --store_mutations on the Philox path
    iss::MutRecord *d_pmut = nullptr;
    uint32_t *d_pmut_count = nullptr;
    int64_t pmut_cap = 0;
    int64_t last_row0 = 0, last_n = 0;  // rows of the last iss_generate call (their flags tell which rows are stale)
    std::vector<int64_t> last_first;     // the last call was a batch: its item_first (rows last_row0 + ...), else empty
    std::vector<int64_t> last_off;       // ... and the arena offsets its descriptors carry
    unsigned max_main_grid = 0;
    std::string main_kernel;             // the hot kernel of the last Philox-mode call (iss_main_kernel)
    uint64_t *stats = nullptr;
    // reference-compatible MT19937 mode (iss_mt_compat.hip.h)
    struct {
        bool seeded = false;
        iss::MtState *d_state = nullptr;      // [2]: CPython random, numpy
        uint32_t *buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // stream x ping-pong
        int cur[2] = {0, 0};
        size_t cap[2] = {0, 0}, fill[2] = {0, 0}, used[2] = {0, 0};
        iss::MtWalkResult *d_res = nullptr;
        iss::MtGauss *d_gauss = nullptr;
        bool has_frag = false;
        double frag_mu = 0, frag_sd = 0;
        iss::MutRecord *d_mut = nullptr;  // --store_mutations rows of the last iss_generate_mt call
        int64_t mut_cap = 0, mut_n = 0;
        hipEvent_t ev_main = nullptr, ev_fill = nullptr;  // ordering between ctx->stream and the fill stream
        iss::MtPhredAmb *d_amb = nullptr;  // BasicErrorModel: [0, CAP) phreds for the host, [CAP, 2 CAP) its answers
        iss::MtPairRec *d_rec = nullptr;  // k_mt_resolve -> k_mt_emit: stream offsets of one launch's pairs
        int32_t *d_mut_cnt = nullptr;     // k_mt_emit, --store_mutations: rows per (pair, mate), then their offsets
        int64_t *d_mut_off = nullptr;
        int64_t n_resolved = 0, n_walked = 0;  // pairs by path (statistics, iss_mt_path_counts)
        int64_t pool_ch = 0;  // != 0: the chain (streams, buffers, records) is a worker's of the set below, lent for one call:
                              // iss_generate_mt takes turns of this many pairs and leaves the buffers as they are
    } mt;
    // MT mode, W workers per launch (iss_mt_workers_seed / iss_generate_mt_workers): the reference's N workers (seed + cpu_number,
    // iss/generator.py:234-236) as N chains side by side -- one workgroup per worker and kernel, job tables in HBM
    struct MtSet {
        int W = 0;
        bool started = false, poisoned = false;  // a call that fails after it began leaves streams and rows undefined: re-seed (iss_generate_mt_workers)
        int64_t ch = 0;                      // pairs per worker and turn
        size_t cap[2] = {0, 0};              // words per (worker, stream, ping-pong buffer)
        int buf_turns = 0;                   // ... = this many turns' words (worst case)
        iss::MtState *d_state = nullptr;     // [W][2]: CPython random, numpy
        // [stream][buffer]: W x cap[stream] words, MT_SET_BUFS buffers in rotation.  A stream's words are appended to its current
        // buffer turn after turn; at the buffer's end the stream moves to the next one of the rotation (mt_set_reserve).  Two: the
        // words of turn t + 1 then go into the buffer the emitter of turn t - 1 may still read, so that fill starts behind it.  (Three -- the
        // fill never waits for an emitter -- were built and measured in round 5: 3.1e7 against 4.2e7 pairs/s at W = 256: fill,
        // emitter and resolver then all start together and the resolver, the chain everything waits for, is the one that loses.)
        uint32_t *buf[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // (the first MT_SET_BUFS of each are allocated)
        std::vector<int64_t> last_read;      // [W * 2][3]: the turn whose emitter reads that buffer (-1: none in flight)
        iss::MtWalkResult *d_res = nullptr;  // [W]
        iss::MtGauss *d_gauss = nullptr;     // [W]
        iss::MtPairRec *d_rec = nullptr;     // [2][W][ch]: the resolver of turn t + 1 runs beside the emitter of turn t
        hipEvent_t ev_emit[2] = {nullptr, nullptr};  // the emitter of the last turn of either parity
        hipEvent_t ev_side = nullptr, ev_turn = nullptr;  // side stream (the walker beside the resolver) <-> main stream
        std::vector<int> cur;                // [W * 2]
        std::vector<size_t> fill, used;      // [W * 2]
        // job tables: pinned host staging + device copies, two sets (turn parity) of
        // [fill: ensure 2W | fill: ahead 2W | move: ensure 2W | move: commit 2W] and [resolve W | walk W | emit W]
        uint8_t *h_jobs = nullptr, *d_jobs = nullptr;
        size_t jobs_bytes = 0;               // of ONE set
        iss::MtWalkResult *h_res = nullptr;  // pinned [W]
        int64_t turns = 0;
        int64_t n_resolved = 0, n_walked = 0;
    } mts;
    FastqPipe fq;
    // timing
    bool timing = false, timing_main_only = false;
    std::vector<TimedLaunch> timed;
    double ms_acc[4] = {0, 0, 0, 0};
    int64_t n_launches = 0;
};

namespace {

int fail(iss_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->last_error = msg;
    g_last_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(ctx, ISS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

template <typename T>
int upload(iss_ctx *ctx, const T *host, size_t n, T **dev, std::vector<void *> *track) {
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    if (track) track->push_back(p);
    if (n) HIP_TRY(ctx, hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T *>(p);
    return 0;
}

// The switches of the library -- each selects a code path the tests force: which indel path a model
// takes, the tile / guide-bit sweeps, the rounding guard of MT mode.  Read at iss_ctx_create, at every model upload and once
// per generate call (never per launch).
void read_switches(iss_ctx *ctx) {
    const char *e;
    ctx->light_below = (e = getenv("ISS_LIGHT_INDELS")) ? atof(e) : 2e-3;
    ctx->env_tiles = (e = getenv("ISS_TILES")) ? atoi(e) : 0;
    ctx->env_guide_bits = (e = getenv("ISS_GUIDE_BITS")) ? std::min(8, std::max(6, atoi(e))) : 0;
    ctx->mt_guard = (e = getenv("ISS_MT_GUARD")) ? atof(e) : 1e-6;
    ctx->debug_model = getenv("ISS_DEBUG_MODEL") != nullptr;
    ctx->env_chunk_pairs = (e = getenv("ISS_CHUNK_PAIRS")) ? std::max<int64_t>(1, atoll(e)) : 0;
    ctx->env_main_wgs = (e = getenv("ISS_MAIN_WGS")) ? std::max(1, atoi(e)) : 0;
    ctx->env_group = (e = getenv("ISS_MAIN_GROUP")) ? atoi(e) : -1;
    ctx->env_group_min = (e = getenv("ISS_MAIN_GROUP_MIN")) ? atoi(e) : 0;
}

void free_model(iss_ctx *ctx) {
    for (void *p : ctx->model_allocs) (void)hipFree(p);
    ctx->model_allocs.clear();
    ctx->have_model = false;
}

void free_outputs(iss_ctx *ctx) {
    if (ctx->out[0]) (void)hipFree(ctx->out[0]);
    for (auto &p : ctx->out) p = nullptr;
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    ctx->d_stage = nullptr; ctx->stage_cap = 0;
    if (ctx->desc) (void)hipFree(ctx->desc);
    for (int k = 0; k < 2; ++k) {
        if (ctx->desc_buf[k]) (void)hipFree(ctx->desc_buf[k]);
        if (ctx->flags_buf[k]) (void)hipFree(ctx->flags_buf[k]);
        if (ctx->fixl_buf[k]) (void)hipFree(ctx->fixl_buf[k]);
        ctx->desc_buf[k] = nullptr; ctx->flags_buf[k] = nullptr; ctx->fixl_buf[k] = nullptr;
        ctx->ev_call_valid[k] = false;
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->ev_count[k] && (k == 0 || ctx->ev_count[k] != ctx->ev_count[0])) (void)hipFree(ctx->ev_count[k]);
        if (ctx->ev_list[k] && (k == 0 || ctx->ev_list[k] != ctx->ev_list[0])) (void)hipFree(ctx->ev_list[k]);
        if (ctx->read_list[k] && (k == 0 || ctx->read_list[k] != ctx->read_list[0])) (void)hipFree(ctx->read_list[k]);
        if (ctx->read_list1[k] && (k == 0 || ctx->read_list1[k] != ctx->read_list1[0])) (void)hipFree(ctx->read_list1[k]);
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->script[k]) (void)hipFree(ctx->script[k]);
        ctx->script[k] = nullptr;
    }
    for (auto &v : ctx->ev_slot_valid) v = false;  // (free_outputs follows a sync_all: nothing of the old buffers is in flight)
    for (int k = 0; k < 2; ++k) { ctx->ev_count[k] = ctx->ev_list[k] = nullptr; ctx->read_list[k] = nullptr; ctx->read_list1[k] = nullptr; }
    ctx->desc = nullptr; ctx->flags = nullptr; ctx->fix_list = nullptr;
    ctx->capacity = 0;
}

void free_mt_set(iss_ctx *ctx) {
    auto &t = ctx->mts;
    if (t.d_state) (void)hipFree(t.d_state);
    if (t.d_res) (void)hipFree(t.d_res);
    if (t.d_gauss) (void)hipFree(t.d_gauss);
    if (t.d_rec) (void)hipFree(t.d_rec);
    for (auto &st : t.buf) for (auto &b : st) { if (b) (void)hipFree(b); b = nullptr; }
    if (t.h_jobs) (void)hipHostFree(t.h_jobs);
    if (t.d_jobs) (void)hipFree(t.d_jobs);
    if (t.h_res) (void)hipHostFree(t.h_res);
    for (auto &e : t.ev_emit) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (t.ev_side) (void)hipEventDestroy(t.ev_side);
    if (t.ev_turn) (void)hipEventDestroy(t.ev_turn);
    t.ev_side = t.ev_turn = nullptr;
    t.d_state = nullptr; t.d_res = nullptr; t.d_gauss = nullptr; t.d_rec = nullptr; t.h_jobs = nullptr; t.d_jobs = nullptr; t.h_res = nullptr;
    t.W = 0; t.ch = 0; t.buf_turns = 0; t.cap[0] = t.cap[1] = 0; t.jobs_bytes = 0;
    t.cur.clear(); t.fill.clear(); t.used.clear(); t.last_read.clear();
}

void free_mt(iss_ctx *ctx) {
    free_mt_set(ctx);
    if (ctx->mt.d_state) (void)hipFree(ctx->mt.d_state);
    if (ctx->mt.d_res) (void)hipFree(ctx->mt.d_res);
    if (ctx->mt.d_mut) (void)hipFree(ctx->mt.d_mut);
    if (ctx->mt.d_gauss) (void)hipFree(ctx->mt.d_gauss);
    if (ctx->mt.d_rec) (void)hipFree(ctx->mt.d_rec);
    if (ctx->mt.d_mut_cnt) (void)hipFree(ctx->mt.d_mut_cnt);
    if (ctx->mt.d_mut_off) (void)hipFree(ctx->mt.d_mut_off);
    ctx->mt.d_rec = nullptr; ctx->mt.d_mut_cnt = nullptr; ctx->mt.d_mut_off = nullptr;
    if (ctx->mt.d_amb) (void)hipFree(ctx->mt.d_amb);
    ctx->mt.d_amb = nullptr;
    if (ctx->mt.ev_main) (void)hipEventDestroy(ctx->mt.ev_main);
    if (ctx->mt.ev_fill) (void)hipEventDestroy(ctx->mt.ev_fill);
    ctx->mt.ev_main = ctx->mt.ev_fill = nullptr;
    ctx->mt.d_gauss = nullptr;
    ctx->mt.d_mut = nullptr; ctx->mt.mut_cap = 0;
    for (auto &st : ctx->mt.buf) for (auto &b : st) { if (b) (void)hipFree(b); b = nullptr; }
    ctx->mt.d_state = nullptr; ctx->mt.d_res = nullptr; ctx->mt.seeded = false;
    ctx->mt.cap[0] = ctx->mt.cap[1] = 0;
}

// MT19937 seeding, as CPython's random.seed(int) (init_by_array over the 32-bit digits of |seed|) and
// numpy's legacy RandomState.seed(int) (init_genrand) do it -- iss/generator.py:234-236.
void mt_init_genrand(uint32_t *mt, uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}
void mt_init_by_array(uint32_t *mt, const uint32_t *key, int len) {
    mt_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    for (int k = std::max(624, len); k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        if (++i >= 624) { mt[0] = mt[623]; i = 1; }
        if (++j >= len) j = 0;
    }
    for (int k = 623; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

// int(loc + scale * gaussian) exactly as numpy's legacy_gauss / normal evaluate it (libm, no contraction):
// f = sqrt(-2*log(r2)/r2); fresh value f*x2, cached value f*x1.
int64_t host_int_normal(double x1v, double x2v, bool cached, double loc, double scale) {
    volatile double x1 = x1v, x2 = x2v;
    volatile double r2 = x1 * x1;
    volatile double t2 = x2 * x2;
    r2 = r2 + t2;
    volatile double f = -2.0 * log(r2);
    f = f / r2;
    f = sqrt(f);
    volatile double gval = cached ? f * x1 : f * x2;
    volatile double sc = scale * gval;
    const double x = loc + sc;
    return (int64_t)x;
}

// MT19937 blocks are generated on the auxiliary stream (ctx->fill_stream) so that the NEXT chunk's words can be
// produced while the current chunk is consumed on ctx->stream.  The fill first waits for everything queued on
// ctx->stream so far (an earlier k_mt_emit may still read the target buffer); ctx->stream waits for ev_fill
// before it touches the new words (mt_fill_join).
int mt_fill_async(iss_ctx *ctx, uint32_t *const dst[2], const uint32_t blocks[2]) {
    auto &m = ctx->mt;
    if (!blocks[0] && !blocks[1]) return 0;
    if (!m.ev_main) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&m.ev_main, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&m.ev_fill, hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipEventRecord(m.ev_main, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->fill_stream, m.ev_main, 0));
    hipLaunchKernelGGL(iss::k_mt_fill, dim3(2), dim3(iss::FILL_THREADS), 0, ctx->fill_stream, m.d_state, dst[0], dst[1], blocks[0],
                       blocks[1]);
    HIP_TRY(ctx, hipEventRecord(m.ev_fill, ctx->fill_stream));
    return 0;
}
int mt_fill_join(iss_ctx *ctx) {
    if (ctx->mt.ev_fill) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->mt.ev_fill, 0));
    return 0;
}

// BasicErrorModel phred of one gaussian exactly as the reference computes it (libm, no contraction):
// legacy_gauss value f*x2 (fresh) / f*x1 (cached), loc + scale*g, min(q, cap), int(round(-10*log10(1 - p))).
int host_basic_phred(double x1v, double x2v, bool cached, double loc, double scale, double cap) {
    volatile double x1 = x1v, x2 = x2v;
    volatile double r2 = x1 * x1;
    volatile double t2 = x2 * x2;
    r2 = r2 + t2;
    volatile double f = -2.0 * log(r2);
    f = f / r2;
    f = sqrt(f);
    volatile double gval = cached ? f * x1 : f * x2;
    volatile double sc = scale * gval;
    volatile double p = loc + sc;
    if (p > cap) p = cap;
    volatile double y = 1.0 - p;
    volatile double x = -10.0 * log10(y);
    return (int)nearbyint(x);  // round-half-even, like Python's round() on a float
}

// make at least `want[s]` unconsumed words available in stream s (capacity permitting)
int mt_ensure(iss_ctx *ctx, const size_t want[2]) {
    uint32_t blocks[2] = {0, 0};
    uint32_t *dst[2] = {nullptr, nullptr};
    { int rc_ = mt_fill_join(ctx); if (rc_) return rc_; }
    for (int s = 0; s < 2; ++s) {
        auto &m = ctx->mt;
        const size_t left = m.fill[s] - m.used[s];
        if (left >= want[s]) continue;
        const int nxt = m.cur[s] ^ 1;
        if (left)
            HIP_TRY(ctx, hipMemcpyAsync(m.buf[s][nxt], m.buf[s][m.cur[s]] + m.used[s], left * sizeof(uint32_t),
                                        hipMemcpyDeviceToDevice, ctx->stream));
        const size_t room = (m.cap[s] - left) / 624;
        blocks[s] = (uint32_t)std::min(room, (want[s] - left + 623) / 624);
        dst[s] = m.buf[s][nxt] + left;
        m.cur[s] = nxt;
        m.used[s] = 0;
        m.fill[s] = left + (size_t)blocks[s] * 624;
    }
    { int rc_ = mt_fill_async(ctx, dst, blocks); if (rc_) return rc_; }
    return mt_fill_join(ctx);
}

// Prefetch for the chunk AFTER the one about to be launched: stream s gets `want_next[s]` fresh words in its
// other buffer, placed behind room for everything that is unconsumed now (the running chunk will consume some
// of it).  mt_prefetch_commit, called once the running chunk has finished, moves the actual leftover in front
// of the prefetched words and switches buffers.
struct MtPrefetch {
    bool on[2] = {false, false};
    size_t at[2] = {0, 0};
    uint32_t blocks[2] = {0, 0};
};
int mt_prefetch_begin(iss_ctx *ctx, const size_t want_cur[2], const size_t want_next[2], MtPrefetch *pf) {
    auto &m = ctx->mt;
    uint32_t *dst[2] = {nullptr, nullptr};
    for (int s = 0; s < 2; ++s) {
        const size_t avail = m.fill[s] - m.used[s];
        if (avail >= want_cur[s] + want_next[s]) continue;  // enough for both chunks already
        const size_t blocks = (want_next[s] + 623) / 624;
        if (avail + blocks * 624 > m.cap[s]) continue;       // no room: the next mt_ensure fills synchronously
        pf->on[s] = true;
        pf->at[s] = avail;
        pf->blocks[s] = (uint32_t)blocks;
        dst[s] = m.buf[s][m.cur[s] ^ 1] + avail;
    }
    return mt_fill_async(ctx, dst, pf->blocks);
}
int mt_prefetch_commit(iss_ctx *ctx, const MtPrefetch &pf) {
    auto &m = ctx->mt;
    for (int s = 0; s < 2; ++s) {
        if (!pf.on[s]) continue;
        const size_t left = m.fill[s] - m.used[s];  // <= pf.at[s]
        const int nxt = m.cur[s] ^ 1;
        if (left)
            HIP_TRY(ctx, hipMemcpyAsync(m.buf[s][nxt] + (pf.at[s] - left), m.buf[s][m.cur[s]] + m.used[s],
                                        left * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        m.cur[s] = nxt;
        m.used[s] = pf.at[s] - left;
        m.fill[s] = pf.at[s] + (size_t)pf.blocks[s] * 624;
    }
    return 0;
}

// dynamic LDS of k_main: quality rows + deferred-work queues
// k_main_g: the instantiations (iterations per pass NI, passes per group NP) the library holds -- a group is at most five
// iterations (8 registers of rows each) -- and the choice of NP for a model.  X(NI, NP) with a trailing separator per entry.
#define ISS_MAIN_G_LIST(X) X(5, 1) X(4, 1) X(3, 1) X(2, 2) X(2, 1) X(1, 2)
#define ISS_MAIN_G_PTR(NI_, NP_) reinterpret_cast<const void *>(iss::k_main_g<true, NI_, NP_>),
constexpr uint32_t MAIN_GROUP_MIN_ROUND = 1;
constexpr int64_t MAIN_CHUNK_PAIRS = 12582912;  // pairs per launch of a call at most (generate_core; ISS_CHUNK_PAIRS overrides)
// Passes per group (0: k_main).  `want` (ISS_MAIN_GROUP) if the library holds it.  Else by the lane-items a wavefront defers per
// iteration, E = 64 (1 - (1 - p_defer)^16): a group should end with about one round's worth of entries (64 / E iterations), and a
// model that defers little gains less from patches in time than a closing round per group costs.  Measured, interleaved on one
// box (profiles/r06_ab_runs.txt; k_main ms per 5 M pairs, k_main -> k_main_g): HiSeq (E 18) 1.235 -> 1.12 with groups of 2 x 2
// iterations, 1.17 with 1 x 2; MiSeq (E 29) 4.03 -> 3.44 with 1 x 2, 3.64 with 2 x 2; NextSeq (E 23, four iterations per
// pass) 2.85 -> 2.60; NovaSeq (E 9, five iterations per pass) 1.155 -> 1.20: k_main stays.
static int main_group_passes(const iss::DevModel &M, int ni, int want) {
    const double e = 64.0 * (1.0 - std::pow(1.0 - std::min(std::max((double)M.p_defer, 0.0), 1.0), 16.0));
    if (want < 0 && e < 15.0) return 0;
    const int target = want > 0 ? want : std::max(1, (int)std::lround(64.0 / std::max(e, 1.0) / (double)ni));
    int best = 0;
#define ISS_MAIN_G_PICK(NI_, NP_) if (ni == NI_ && (want > 0 ? NP_ == want : (NP_ <= target && NP_ > best))) best = NP_;
    ISS_MAIN_G_LIST(ISS_MAIN_G_PICK)
#undef ISS_MAIN_G_PICK
    if (!best && want <= 0) {  // (no instantiation that small: the smallest one for ni)
#define ISS_MAIN_G_PICK(NI_, NP_) if (ni == NI_ && (!best || NP_ < best)) best = NP_;
        ISS_MAIN_G_LIST(ISS_MAIN_G_PICK)
#undef ISS_MAIN_G_PICK
    }
    return best;
}

size_t main_lds_bytes(const iss::DevModel &M) {
    return ((size_t)iss::MAIN_LUT_WORDS + M.tile_words + iss::MAIN_MUT_WORDS + (size_t)2 * M.TP * 4 + (size_t)(iss::MAIN_THREADS / 64) * iss::SLOW_RING * 3) * 4;
}

int settle_timing(iss_ctx *ctx) {
    static const int first[4] = {0, 1, 3, 5};
    for (auto &t : ctx->timed) {
        HIP_TRY(ctx, hipEventSynchronize(t.ev[2]));
        if (t.has_scan && t.ev[6]) HIP_TRY(ctx, hipEventSynchronize(t.ev[6]));
        for (int k = 0; k < 4; ++k) {
            if (k >= 2 && !t.has_scan) continue;
            hipEvent_t e_end = t.ev[first[k] + 1];
            if (k == 0 && t.scan_first && t.ev[3]) e_end = t.ev[3];
            if (k == 0 && t.ev[7]) e_end = t.ev[7];
            if (!t.ev[first[k]] || !e_end) continue;  // k_main-only timing
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, t.ev[first[k]], e_end));
            ctx->ms_acc[k] += ms;
        }
        for (auto &e : t.ev) if (e) (void)hipEventDestroy(e);
    }
    ctx->timed.clear();
    return 0;
}

// everything queued on both streams has finished
int sync_all(iss_ctx *ctx) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->setup_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->indel_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->fill_stream));
    if (ctx->emit_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->emit_stream));
    return 0;
}

int pwrite_all(int fd, const uint8_t *p, size_t n, int64_t off) {
    while (n) {
        const ssize_t k = pwrite(fd, p, n, (off_t)off);
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        p += k; n -= (size_t)k; off += k;
    }
    return 0;
}

void fastq_writer_loop(iss_ctx *ctx) {
    FastqPipe &q = ctx->fq;
    (void)hipSetDevice(ctx->device);
    for (;;) {
        FastqJob job;
        {
            std::unique_lock<std::mutex> lk(q.mu);
            q.cv.wait(lk, [&] { return q.stop || !q.jobs.empty(); });
            if (q.jobs.empty()) return;
            job = q.jobs.front();
        }
        std::string err;
        int64_t gz_wrote[2] = {0, 0};  // compressed mode: only this thread moves the file offsets (under the mutex)
        const bool dbg = getenv("ISS_FASTQ_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        if (hipEventSynchronize(q.ev_copy[job.slot]) != hipSuccess) err = "device copy of the FASTQ text failed";
        const auto t1 = std::chrono::steady_clock::now();
        if (err.empty() && job.gzip) {
            // one gzip member per file: header, the DEFLATE blocks (fetched now that their size is known), an empty
            // final block, CRC-32 and ISIZE of the text (RFC 1952)
            for (int mate = 0; mate < 2 && err.empty(); ++mate) {
                const uint64_t total = *q.h_total[job.slot][mate];
                if (total > q.comp_cap) { err = "compressed FASTQ larger than its buffer"; break; }
                if (hipMemcpyAsync(q.h_text[job.slot][mate], q.d_comp[job.slot][mate], total, hipMemcpyDeviceToHost,
                                   q.data_stream) != hipSuccess) err = "device copy of the compressed FASTQ failed";
            }
            if (err.empty() && hipStreamSynchronize(q.data_stream) != hipSuccess) err = "device copy of the compressed FASTQ failed";
            if (err.empty()) {
                std::thread th[2];
                int rc[2] = {0, 0};
                uint64_t wrote[2] = {0, 0};
                int64_t gz_at[2];
                {
                    std::lock_guard<std::mutex> lk(q.mu);
                    gz_at[0] = q.off[0];
                    gz_at[1] = q.off[1];
                }
                for (int mate = 0; mate < 2; ++mate) {
                    th[mate] = std::thread([&, mate] {
                        const uint64_t total = *q.h_total[job.slot][mate];
                        // raw CRC of the text from the per-block raw CRCs, then the initial / final conditioning
                        uint32_t raw = 0;
                        const uint32_t *bc = q.h_bcrc[job.slot][mate];
                        const uint64_t last_len = job.bytes - (uint64_t)(job.n_blocks - 1) * iss::DEFLATE_BLOCK;
                        uint32_t op_last[32], op_all[32];
                        iss::crc_shift_operator(last_len, op_last);
                        iss::crc_shift_operator(job.bytes, op_all);
                        for (uint32_t b = 0; b < job.n_blocks; ++b)
                            raw = iss::gf2_times(b + 1 == job.n_blocks ? op_last : q.op_block, raw) ^ bc[b];
                        const uint32_t crc = raw ^ iss::gf2_times(op_all, 0xffffffffu) ^ 0xffffffffu;
                        const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
                        uint8_t tail[10] = {0x03, 0x00};
                        const uint32_t isize = (uint32_t)job.bytes;
                        memcpy(tail + 2, &crc, 4);
                        memcpy(tail + 6, &isize, 4);
                        const int64_t at = gz_at[mate];
                        if (pwrite_all(job.fd[mate], head, 10, at) || pwrite_all(job.fd[mate], q.h_text[job.slot][mate], total, at + 10) ||
                            pwrite_all(job.fd[mate], tail, 10, at + 10 + (int64_t)total))
                            rc[mate] = errno;
                        wrote[mate] = 20 + total;
                    });
                }
                for (auto &t : th) t.join();
                for (int mate = 0; mate < 2; ++mate) {
                    if (rc[mate]) err = std::string("write failed: ") + strerror(rc[mate]);
                    gz_wrote[mate] = (int64_t)wrote[mate];
                }
            }
        } else if (err.empty()) {
            // invariant: every work item's text starts with '@' right behind a line feed and the job ends with one (the
            // closed-form sizes the host computed are the layout the device wrote)
            for (int mate = 0; mate < 2 && err.empty(); ++mate) {
                const uint8_t *t = q.h_text[job.slot][mate];
                bool ok = job.bytes > 0 && t[job.bytes - 1] == '\n';
                for (uint64_t at : job.item_off) ok = ok && at < job.bytes && t[at] == '@' && (at == 0 || t[at - 1] == '\n');
                if (!ok) err = "FASTQ text does not have the record layout its size was computed from";
            }
            // scattered items (the workers of a set, every one at its own place of the final files): the items dealt to a few
            // threads per file; the text of item k is [item_off[k], item_off[k + 1])
            if (err.empty() && !job.item_file_off.empty()) {
                const size_t n_it = job.item_off.size();
                const int per_file = std::max(1, std::min<int>(job.threads, 8));
                std::vector<std::thread> th;
                std::vector<int> rc((size_t)2 * per_file, 0);
                for (int mate = 0; mate < 2; ++mate)
                    for (int t = 0; t < per_file; ++t) {
                        int *r = &rc[(size_t)mate * per_file + t];
                        th.emplace_back([&, mate, t, r] {
                            for (size_t k = (size_t)t; k < n_it && !*r; k += (size_t)per_file) {
                                const uint64_t a = job.item_off[k], b = k + 1 < n_it ? job.item_off[k + 1] : job.bytes;
                                if (pwrite_all(job.fd[mate], q.h_text[job.slot][mate] + a, b - a, job.item_file_off[k])) *r = errno ? errno : EIO;
                            }
                        });
                    }
                for (auto &t : th) t.join();
                for (int r : rc) if (r) err = std::string("write failed: ") + strerror(r);
            }
            // both files in parallel, each cut into pieces written with pwrite at their final offsets (a small job --
            // one record of a long work list -- is written by this thread: spawning threads would cost more)
            const bool small_job = job.bytes <= (1u << 20) || !job.item_file_off.empty();
            if (!job.item_file_off.empty()) job.bytes = 0;  // (written above)
            for (int mate = 0; small_job && mate < 2 && err.empty(); ++mate)
                if (err.empty() && pwrite_all(job.fd[mate], q.h_text[job.slot][mate], job.bytes, job.off[mate]))
                    err = std::string("write failed: ") + strerror(errno);
            const size_t piece = std::max<size_t>((job.bytes + (size_t)job.threads - 1) / (size_t)job.threads, 1 << 20);
            std::vector<std::thread> th;
            std::vector<int> rc;
            for (int mate = 0; mate < 2 && !small_job && err.empty(); ++mate)
                for (size_t at = 0; at < job.bytes; at += piece) rc.push_back(0);
            size_t k = 0;
            for (int mate = 0; mate < 2 && !small_job && !rc.empty(); ++mate)
                for (size_t at = 0; at < job.bytes; at += piece, ++k) {
                    const size_t n = std::min(piece, job.bytes - at);
                    const uint8_t *src = q.h_text[job.slot][mate] + at;
                    int *r = &rc[k];
                    const int fd = job.fd[mate];
                    const int64_t off = job.off[mate] + (int64_t)at;
                    th.emplace_back([=] { *r = pwrite_all(fd, src, n, off) ? errno : 0; });
                }
            for (auto &t : th) t.join();
            for (int r : rc) if (r) err = std::string("write failed: ") + strerror(r);
        }
        if (dbg) {
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[fastq] slot %d: %.1f MB per file, waited %.1f ms for the copy, wrote in %.1f ms (%d pieces per file)\n",
                    job.slot, job.bytes / 1e6, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count(), job.threads);
        }
        {
            std::lock_guard<std::mutex> lk(q.mu);
            q.jobs.pop_front();
            q.busy[job.slot] = false;
            if (job.gzip)  // (text jobs were accounted for when they were queued)
                for (int mate = 0; mate < 2; ++mate) { q.off[mate] += gz_wrote[mate]; q.accounted[mate] += gz_wrote[mate]; }
            if (!err.empty() && q.error.empty()) q.error = err;
        }
        q.cv.notify_all();
    }
}

// all queued text is in the files; the descriptors stand at the end of what was written
int fastq_flush(iss_ctx *ctx, bool keep_files = false) {
    FastqPipe &q = ctx->fq;
    if (!q.ready) return 0;
    std::string err;
    {
        std::unique_lock<std::mutex> lk(q.mu);
        q.cv.wait(lk, [&] { return q.jobs.empty(); });
        err = q.error;
        q.error.clear();
    }
    for (int m = 0; m < 2; ++m) {
        if (q.fd[m] < 0) continue;
        // invariants: the offset is the attach offset plus every job's bytes, and the file holds at least that much
        struct stat st;
        if (err.empty() && q.off[m] != q.attached_off[m] + q.accounted[m]) err = "FASTQ pipeline: file offset and queued bytes disagree";
        if (err.empty() && fstat(q.fd[m], &st) == 0 && S_ISREG(st.st_mode) && (int64_t)st.st_size < q.off[m])
            err = "FASTQ pipeline: file shorter than the bytes written to it";
        (void)lseek(q.fd[m], (off_t)q.off[m], SEEK_SET);
    }
    if (!keep_files) q.fd[0] = q.fd[1] = -1;
    if (!err.empty()) return fail(ctx, ISS_E_IO, err);
    return 0;
}

// the same, but the files stay attached (buffers are about to be reallocated in the middle of a run)
int fastq_flush_keep(iss_ctx *ctx) { return fastq_flush(ctx, true); }

void fastq_free_buffers(iss_ctx *ctx) {
    FastqPipe &q = ctx->fq;
    for (auto &sl : q.d_text) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.h_text) for (auto &p : sl) { if (p) (void)hipHostFree(p); p = nullptr; }
    for (auto &sl : q.d_comp) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.d_bbytes) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.d_bcrc) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.d_boff) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.h_bcrc) for (auto &p : sl) { if (p) (void)hipHostFree(p); p = nullptr; }
    q.cap = 0;
    q.comp_cap = 0;
    q.blocks_cap = 0;
}

void fastq_shutdown(iss_ctx *ctx) {
    FastqPipe &q = ctx->fq;
    if (!q.ready) return;
    (void)fastq_flush(ctx);
    {
        std::lock_guard<std::mutex> lk(q.mu);
        q.stop = true;
    }
    q.cv.notify_all();
    if (q.writer.joinable()) q.writer.join();
    fastq_free_buffers(ctx);
    for (auto &sl : q.d_hist) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.d_code) for (auto &p : sl) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto &sl : q.h_total) for (auto &p : sl) { if (p) (void)hipHostFree(p); p = nullptr; }
    if (q.data_stream) (void)hipStreamDestroy(q.data_stream);
    for (int sl = 0; sl < 2; ++sl) {
        if (q.h_items[sl]) (void)hipHostFree(q.h_items[sl]);
        if (q.d_items[sl]) (void)hipFree(q.d_items[sl]);
        if (q.h_ids[sl]) (void)hipHostFree(q.h_ids[sl]);
        if (q.d_ids[sl]) (void)hipFree(q.d_ids[sl]);
        q.h_items[sl] = q.d_items[sl] = nullptr;
        q.h_ids[sl] = q.d_ids[sl] = nullptr;
        q.items_cap[sl] = q.ids_cap[sl] = 0;
    }
    for (auto &e : q.ev_fmt) if (e) (void)hipEventDestroy(e);
    for (auto &e : q.ev_copy) if (e) (void)hipEventDestroy(e);
    if (q.copy_stream) (void)hipStreamDestroy(q.copy_stream);
    q.ready = false;
}

}  // namespace

extern "C" {

int iss_abi_version(void) { return ISS_ABI_VERSION; }

#ifndef ISS_BUILD_ID
#define ISS_BUILD_ID "unknown"
#endif
const char *iss_build_id(void) { return ISS_BUILD_ID; }

const char *iss_last_error(const iss_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

int iss_ctx_create(int device_ordinal, iss_ctx **out) {
    if (!out) return fail(nullptr, ISS_E_INVALID, "iss_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, ISS_E_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, ISS_E_INVALID, "device ordinal out of range");
    iss_ctx *ctx = new iss_ctx();
    ctx->device = device_ordinal;
    HIP_TRY(ctx, hipSetDevice(device_ordinal));
    {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, device_ordinal));
        ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const void *mains[8] = {reinterpret_cast<const void *>(iss::k_main<false, false, false>), reinterpret_cast<const void *>(iss::k_main<false, true, false>),
                                reinterpret_cast<const void *>(iss::k_main<true, false, false>), reinterpret_cast<const void *>(iss::k_main<true, true, false>),
                                reinterpret_cast<const void *>(iss::k_main<false, false, true>), reinterpret_cast<const void *>(iss::k_main<false, true, true>),
                                reinterpret_cast<const void *>(iss::k_main<true, false, true>), reinterpret_cast<const void *>(iss::k_main<true, true, true>)};
        for (const void *f : mains) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        const void *grouped[] = {ISS_MAIN_G_LIST(ISS_MAIN_G_PTR)};
        for (const void *f : grouped) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(iss::k_mt_walk),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(iss::k_indel_fixup),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(iss::k_setup),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(iss::k_indel_scan),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        const void *scripts[] = {reinterpret_cast<const void *>(iss::k_indel_script<false, 12, false>), reinterpret_cast<const void *>(iss::k_indel_script<true, 12, false>),
                                 reinterpret_cast<const void *>(iss::k_indel_script<false, 26, false>), reinterpret_cast<const void *>(iss::k_indel_script<true, 26, false>),
                                 reinterpret_cast<const void *>(iss::k_indel_script<false, 12, true>), reinterpret_cast<const void *>(iss::k_indel_script<true, 12, true>),
                                 reinterpret_cast<const void *>(iss::k_indel_script<false, 26, true>), reinterpret_cast<const void *>(iss::k_indel_script<true, 26, true>)};
        for (const void *f : scripts) HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    }
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->indel_stream, hipStreamNonBlocking));
    {   // The setup stream gets the highest priority: its hardware queue then comes from another pool than the main stream's
        // (streams of one priority share a few queues), and its small kernels are dispatched as soon as a CU has room.  Measured
        // with engines created one after the other in one process (tools/placement_probe.py, default bench's step): 1.25-1.26 ms
        // per step for every engine, against 1.24-1.29 (one box) and 1.28 / 1.41 alternating (another) at the default priority.
        int prio_least = 0, prio_greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->setup_stream, hipStreamNonBlocking, prio_greatest));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->fill_stream, hipStreamNonBlocking, prio_least));  // (MT mode: see iss_ctx::fill_stream)
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->emit_stream, hipStreamNonBlocking, prio_least));  // (MT mode's worker set: see iss_ctx::emit_stream)
    }
    for (auto &e : ctx->ev_call_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ctx->ev_setup_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ctx->ev_slot_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ctx->ev_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ctx->ev_join) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_inputs, hipEventDisableTiming));
    if (const char *e = getenv("ISS_SETUP_AHEAD")) ctx->setup_ahead = atoi(e) != 0;  // 0: everything in order on one stream
    read_switches(ctx);
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, 256));
    ctx->fix_count = static_cast<uint32_t *>(p);  // FIX_SLOTS counters; +128 B stats; +192 B genome-pack status
    ctx->stats = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(p) + 128);
    HIP_TRY(ctx, hipMemset(p, 0, 256));
    HIP_TRY(ctx, hipMalloc(&p, sizeof(uint32_t) * FIX_SLOTS * 2 * iss::SCAN_MAX_WGS));
    ctx->read_count = static_cast<uint32_t *>(p);
    HIP_TRY(ctx, hipMemset(p, 0, sizeof(uint32_t) * FIX_SLOTS * 2 * iss::SCAN_MAX_WGS));
    ctx->max_main_grid = 2u * (unsigned)ctx->n_cu;
    *out = ctx;
    return 0;
}

void iss_ctx_destroy(iss_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    fastq_shutdown(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->indel_stream) (void)hipStreamSynchronize(ctx->indel_stream);
    if (ctx->fill_stream) (void)hipStreamSynchronize(ctx->fill_stream);
    if (ctx->setup_stream) (void)hipStreamSynchronize(ctx->setup_stream);
    if (ctx->emit_stream) (void)hipStreamSynchronize(ctx->emit_stream);
    for (auto &t : ctx->timed) for (auto &e : t.ev) if (e) (void)hipEventDestroy(e);
    free_model(ctx);
    free_outputs(ctx);
    iss_genome_clear(ctx);
    if (ctx->fix_count) (void)hipFree(ctx->fix_count);
    if (ctx->read_count) (void)hipFree(ctx->read_count);
    if (ctx->d_amb) (void)hipFree(ctx->d_amb);
    if (ctx->d_pmut) (void)hipFree(ctx->d_pmut);
    if (ctx->d_ov_pairs) (void)hipFree(ctx->d_ov_pairs);
    if (ctx->d_ov_frags) (void)hipFree(ctx->d_ov_frags);
    free_mt(ctx);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->indel_stream) (void)hipStreamDestroy(ctx->indel_stream);
    if (ctx->fill_stream) (void)hipStreamDestroy(ctx->fill_stream);
    if (ctx->setup_stream) (void)hipStreamDestroy(ctx->setup_stream);
    if (ctx->emit_stream) (void)hipStreamDestroy(ctx->emit_stream);
    for (auto &e : ctx->ev_call_done) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_setup_done) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_slot_done) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_fork) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_join) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_inputs) (void)hipEventDestroy(ctx->ev_inputs);
    delete ctx;
}

int iss_ctx_set_stream(iss_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return 0;
}

int iss_model_upload(iss_ctx *ctx, const iss_model_tables *t) {
    if (!ctx || !t) return fail(ctx, ISS_E_INVALID, "iss_model_upload: NULL argument");
    if (t->read_length < 2 || t->read_length > iss::FIX_MAX_RL)
        return fail(ctx, ISS_E_INVALID, "read_length must be in [2, 1024]");
    if (t->n_isize > 8000) return fail(ctx, ISS_E_INVALID, "insert-size CDF longer than 8000 entries");
    if (t->n_isize < 1 || t->n_q < 1 || t->n_q > 60)
        return fail(ctx, ISS_E_INVALID, "bad table sizes (per-position quality CDFs must have 1..60 entries)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    free_model(ctx);
    read_switches(ctx);
    const int RL = t->read_length, nq = t->n_q;
    const uint64_t two53 = 1ull << 53;
    auto check = [&](const uint64_t *p, size_t n) { for (size_t i = 0; i < n; ++i) if (p[i] > two53) return false; return true; };
    const size_t n_qthr = (size_t)2 * 4 * RL * nq;
    if (!check(t->isize_thr, t->n_isize) || !check(t->bin_thr, 8) || !check(t->q_thr, n_qthr) ||
        !check(t->subst_thr, (size_t)2 * RL * 12) || !check(t->ins_thr, (size_t)2 * RL * 4) ||
        !check(t->del_thr, (size_t)2 * RL * 4) || !check(t->mut_thr, nq + 1))
        return fail(ctx, ISS_E_INVALID, "threshold above 2^53");
    for (int i = 1; i < t->n_isize; ++i)
        if (t->isize_thr[i] < t->isize_thr[i - 1]) return fail(ctx, ISS_E_INVALID, "insert-size thresholds not monotone");
    for (int o = 0; o < 2; ++o)
        for (int b = 0; b < 4; ++b) {
            if (!t->bin_nonempty[o * 4 + b]) {
                const uint64_t prev = b ? t->bin_thr[o * 4 + b - 1] : 0;
                if (t->bin_thr[o * 4 + b] != prev)
                    return fail(ctx, ISS_E_INVALID, "a mean-quality bin with non-zero probability has no histograms");
                continue;
            }
            for (int p = 0; p < RL; ++p) {
                const uint64_t *row = t->q_thr + ((size_t)(o * 4 + b) * RL + p) * nq;
                for (int k = 1; k < nq; ++k)
                    if (row[k] < row[k - 1]) return fail(ctx, ISS_E_INVALID, "quality thresholds not monotone");
            }
        }
    iss::DevModel &M = ctx->M;
    M = iss::DevModel{};
    M.RL = RL; M.n_isize = t->n_isize; M.n_q = nq;
    if (t->quality_mode != 0 && t->quality_mode != 1) return fail(ctx, ISS_E_INVALID, "quality_mode must be 0 (kde) or 1 (basic)");
    if (t->quality_mode == 1 && (nq < 41 || !(t->basic_sd >= 0.0) || !(t->basic_cap < 1.0) || t->basic_insert_size < 0))
        return fail(ctx, ISS_E_INVALID, "basic model: needs phred thresholds 0..41, sd >= 0, cap < 1, insert size >= 0");
    M.quality_mode = t->quality_mode;
    M.basic_insert_size = t->basic_insert_size;
    M.basic_mean = t->basic_mean; M.basic_sd = t->basic_sd; M.basic_cap = t->basic_cap;
    M.S = (RL + 7) / 8; M.pitch = M.S * 8; M.G = M.S * 2;
    M.row = 128 * ((M.S + 3) / 4);
    // ---- compressed quality rows for k_main: per (orientation, bin slot, position) the distinct
    // 16-bit leading digits of the thresholds, packed t16 << 16 | phred << 8 | te8 (te8 = leading 8 bits of the
    // phred's substitution-test threshold), + a guide (first entry for each value of the top GB bits) + sentinels.
    int n_slots[2] = {0, 0};
    for (int o = 0; o < 2; ++o)
        for (int b = 0; b < 4; ++b) {
            M.bin_slot[o * 4 + b] = -1;
            M.slot_bin[o * 4 + b] = 0;
        }
    for (int o = 0; o < 2; ++o)
        for (int b = 0; b < 4; ++b)
            if (t->bin_nonempty[o * 4 + b]) {
                M.bin_slot[o * 4 + b] = (int8_t)n_slots[o];
                M.slot_bin[o * 4 + n_slots[o]] = (int8_t)b;
                ++n_slots[o];
            }
    if (!n_slots[0] || !n_slots[1]) return fail(ctx, ISS_E_INVALID, "model has no quality histograms");
    M.NB = std::max(n_slots[0], n_slots[1]);
    for (int o = 0; o < 2; ++o)
        for (int sl = n_slots[o]; sl < M.NB; ++sl) M.slot_bin[o * 4 + sl] = M.slot_bin[o * 4];
    // (a threshold of 2^53 -- never an error -- clamps to 255: the digit 255 then ties and is resolved exactly)
    auto te8 = [&](int q) { return (uint32_t)std::min<uint64_t>(t->mut_thr[q] >> 45, 255u); };
    auto build_row = [&](int o, int bin, int p, std::vector<uint32_t> &entries) {
        const uint64_t *row = t->q_thr + ((size_t)(o * 4 + bin) * RL + p) * nq;
        entries.clear();
        for (int i = 0; i < nq; ++i) {
            const uint32_t v = (uint32_t)std::min<uint64_t>(row[i] >> 37, 0xffffu);  // 2^53 (cdf == 1.0) clamps: a tie
            if (entries.empty() || (entries.back() >> 16) != v) entries.push_back((v << 16) | ((uint32_t)i << 8) | te8(i));
        }
        // two closing sentinels (the hot loop reads entries j and j + 1 unconditionally); digit 0xffff
        // "ties" with them and is resolved exactly
        if ((entries.back() >> 16) != 0xffffu) entries.push_back((0xffffu << 16) | ((uint32_t)nq << 8) | te8(nq));
        entries.push_back(entries.back());
        entries.push_back(entries.back());
    };
    std::vector<uint32_t> entries;
    size_t s_max = 0;
    for (int o = 0; o < 2; ++o)
        for (int sl = 0; sl < M.NB; ++sl)
            for (int p = 0; p < RL; ++p) {
                build_row(o, M.slot_bin[o * 4 + sl], p, entries);
                s_max = std::max(s_max, entries.size());
            }
    // Guide resolution: the hot loop resolves a draw with two probes unless > 2 thresholds of its guide
    // bucket lie below the digit ("more", sent to the exact path).  Pick the smallest number of guide bits
    // (6..8) that keeps the expected "more" rate under 0.4 % per draw.
    auto more_rate = [&](int gb) {
        double acc = 0;
        size_t rows = 0;
        for (int o = 0; o < 2; ++o)
            for (int sl = 0; sl < n_slots[o]; ++sl)
                for (int p = 0; p < RL; p += 3) {
                    build_row(o, M.slot_bin[o * 4 + sl], p, entries);
                    const uint32_t width = 1u << (16 - gb);
                    size_t j = 0;
                    for (uint32_t b = 0; b < (1u << gb); ++b) {
                        const uint32_t lo = b * width, hi = lo + width;
                        while (j < entries.size() && (entries[j] >> 16) < lo) ++j;
                        size_t k = j;
                        int inside = 0;
                        uint32_t second = 0;
                        while (k < entries.size() && (entries[k] >> 16) < hi) { if (++inside == 2) second = entries[k] >> 16; ++k; }
                        if (inside >= 2 && hi - 1 > second) acc += (double)(hi - 1 - second);
                    }
                    ++rows;
                }
        return acc / 65536.0 / (double)std::max<size_t>(rows, 1);
    };
    // Guide bits and position tiles, chosen together by a small cost model fitted to measurements (DESIGN.md section 7:
    // NovaSeq / HiSeq / NextSeq / MiSeq sweeps): a workgroup keeps ONE tile of tables in LDS (<= 158 KB: one workgroup per
    // CU is as fast as two, bigger tiles are what pays), the work of a pass has a fixed part next to its ceil(TS / 4)
    // iterations, and every base the two-probe lookup cannot decide costs about twelve hot bases.
    auto tiles_needed = [&](int gb, int *ts_out) {  // fewest tiles whose tables fit one workgroup per CU
        const size_t gs = 4 * ((size_t)(1 << gb) / 4 + s_max) + 1;
        for (int nt = 1; nt <= M.S; ++nt) {
            const int ts = nt > 1 ? ((M.S + nt - 1) / nt + 3) / 4 * 4 : M.S;
            const size_t tg = 2 * (size_t)ts;
            const size_t words = (size_t)iss::MAIN_LUT_WORDS + (2 * (size_t)M.NB * tg * gs + 3) / 4 * 4 + iss::MAIN_MUT_WORDS + 2 * tg * 4 * 4 +
                                 (size_t)(iss::MAIN_THREADS / 64) * iss::SLOW_RING * 3;
            if (words * 4 <= 158 * 1024) { *ts_out = ts; return (M.S + ts - 1) / ts; }
        }
        *ts_out = 0;
        return 0;
    };
    M.GB = 6;
    if (ctx->env_guide_bits) M.GB = ctx->env_guide_bits;
    else {
        double best = 1e30;
        for (int gb = 6; gb <= 8; ++gb) {
            int ts = 0;
            if (!tiles_needed(gb, &ts)) continue;
            // (round 4 refit -- tools/guide_bits_sweep.sh, guide bits 6 / 7 / 8 for four model families: a base the two-probe lookup
            //  cannot decide costs about TWELVE hot bases since its late phred patch is a read-modify-write in HBM (round 2's
            //  fit said eight): HiSeq now takes 8 guide bits and two tiles, 1.37 -> 1.24 ms per 5 M pairs)
            const double cost = (1.0 + 12.0 * more_rate(gb)) * (1.0 + 0.3 / (double)((ts + 3) / 4));
            if (cost < best - 1e-9) { best = cost; M.GB = gb; }
        }
    }
    {   // expected share of bases that leave the hot loop for the exact path (k_main_g's grouping, below, is chosen by it): more
        // than two thresholds of the guide bucket below the digit, or the 8-bit error digit reaching the phred's threshold digit
        double flag = 0;
        size_t rows = 0;
        for (int o = 0; o < 2; ++o)
            for (int sl = 0; sl < n_slots[o]; ++sl)
                for (int p = 0; p < RL; p += 3, ++rows) {
                    const uint64_t *row = t->q_thr + ((size_t)(o * 4 + M.slot_bin[o * 4 + sl]) * RL + p) * nq;
                    double prev = 0;
                    for (int q = 0; q <= nq; ++q) {  // P(phred == q) = cdf[q] - cdf[q-1]; phred nq has the rest
                        const double c = q < nq ? (double)row[q] / 9007199254740992.0 : 1.0;
                        flag += (c - prev) * (double)(256u - te8(q)) / 256.0;
                        prev = c;
                    }
                }
        M.p_defer = (float)(more_rate(M.GB) + flag / (double)std::max<size_t>(rows, 1));
    }
    if (ctx->debug_model) {  // expected share of bases that leave the hot loop
        double err = 0;
        size_t rows = 0;
        for (int o = 0; o < 2; ++o)
            for (int sl = 0; sl < n_slots[o]; ++sl)
                for (int p = 0; p < RL; p += 3, ++rows) {
                    const uint64_t *row = t->q_thr + ((size_t)(o * 4 + M.slot_bin[o * 4 + sl]) * RL + p) * nq;
                    double prev = 0;
                    for (int q = 0; q <= nq; ++q) {  // P(phred == q) = cdf[q] - cdf[q-1]; phred nq has the rest
                        const double c = q < nq ? (double)row[q] / 9007199254740992.0 : 1.0;
                        err += (c - prev) * (1.0 - (double)t->mut_thr[q] / 9007199254740992.0);
                        prev = c;
                    }
                }
        fprintf(stderr, "[model] per base: P(> 2 thresholds below in the guide bucket) %.5f (GB 6: %.5f, 7: %.5f, 8: %.5f), "
                        "P(substitution test fires) %.5f, P(a base leaves the hot loop) %.5f, s_max %zu\n", more_rate(M.GB), more_rate(6), more_rate(7), more_rate(8),
                err / (double)std::max<size_t>(rows, 1), (double)M.p_defer, s_max);
    }
    const int gwords = (1 << M.GB) / 4;
    M.stride_w = (int32_t)(gwords + s_max);
    M.GS = 4 * M.stride_w + 1;
    // Position tiling: the fewest tiles one workgroup per CU can hold (two workgroups share a CU when the tile is small
    // enough anyway).
    auto fits = [&](int n_tiles, size_t budget) {
        M.TS = (M.S + n_tiles - 1) / n_tiles;
        if (n_tiles > 1) M.TS = (M.TS + 3) / 4 * 4;  // tiles start at whole 128-byte lines of the output rows (4 superitems)
        M.TG = M.TS * 2;
        M.TP = M.TG * 4;
        M.tile_words = (2 * M.NB * M.TG * M.GS + 3) / 4 * 4;
        return main_lds_bytes(M) <= budget;
    };
    const size_t two_per_cu = 79 * 1024, one_per_cu = 158 * 1024;
    const int env_tiles = ctx->env_tiles;  // tuning aid
    M.n_tiles = 0;
    if (env_tiles > 0 && fits(env_tiles, one_per_cu)) M.n_tiles = env_tiles;
    (void)two_per_cu;
    for (int nt = 1; !M.n_tiles && nt <= M.S; ++nt)
        if (fits(nt, one_per_cu)) M.n_tiles = nt;
    if (!M.n_tiles) return fail(ctx, ISS_E_INVALID, "quality tables do not fit the LDS even for one superitem (8 positions)");
    if ((M.S + M.TS - 1) / M.TS > iss::MAX_TILES) return fail(ctx, ISS_E_INVALID, "quality tables need more position tiles than the engine supports");
    (void)fits(M.n_tiles, one_per_cu);
    M.n_tiles = (M.S + M.TS - 1) / M.TS;
    if (ctx->debug_model)
        fprintf(stderr, "[model] RL %d G %d NB %d GB %d stride_w %d GS %d TG %d n_tiles %d tile %.1f KB (k_main LDS %.1f KB)\n",
                M.RL, M.G, M.NB, M.GB, M.stride_w, M.GS, M.TG, M.n_tiles, M.tile_words * 4 / 1024.0,
                main_lds_bytes(M) / 1024.0);
    std::vector<uint32_t> qrows((size_t)M.n_tiles * M.tile_words, 0);
    for (int tl = 0; tl < M.n_tiles; ++tl)
        for (int o = 0; o < 2; ++o)
            for (int sl = 0; sl < M.NB; ++sl)
                for (int pp = 0; pp < M.TP; ++pp) {
                    const int p = std::min(tl * M.TP + pp, RL - 1);
                    build_row(o, M.slot_bin[o * 4 + sl], p, entries);
                    uint32_t *dst = qrows.data() + (size_t)tl * M.tile_words +
                                    ((size_t)(o * M.NB + sl) * M.TG + pp / 4) * M.GS + (size_t)(pp & 3) * M.stride_w;
                    uint8_t *guide = reinterpret_cast<uint8_t *>(dst);
                    size_t j = 0;
                    for (uint32_t b = 0; b < (1u << M.GB); ++b) {
                        while ((entries[j] >> 16) < (b << (16 - M.GB))) ++j;
                        guide[b] = (uint8_t)(4 * j);  // byte offset of the entry (<= 4 * 63)
                    }
                    std::copy(entries.begin(), entries.end(), dst + gwords);
                    for (size_t k = gwords + entries.size(); k < (size_t)M.stride_w; ++k) dst[k] = entries.back();
                }
    // substitution table of k_main's exact path (LDS): leading 13 bits of the two thresholds + the alternatives as indices
    // into the (<= 4) distinct letters the model uses
    std::vector<uint32_t> subst13((size_t)M.n_tiles * 2 * M.TP * 4, 0);
    {
        uint8_t letters[4] = {0, 0, 0, 0};
        int n_letters = 0;
        auto letter_index = [&](uint8_t c) {
            for (int i = 0; i < n_letters; ++i) if (letters[i] == c) return i;
            if (n_letters == 4) return -1;
            letters[n_letters] = c;
            return n_letters++;
        };
        for (int tl = 0; tl < M.n_tiles; ++tl)
            for (int o = 0; o < 2; ++o)
                for (int pp = 0; pp < M.TP; ++pp)
                    for (int bi = 0; bi < 4; ++bi) {
                        const int p = std::min(tl * M.TP + pp, RL - 1);
                        const size_t row = ((size_t)(o * RL + p) * 4 + bi) * 3;
                        auto d13 = [](uint64_t T) { return (uint32_t)std::min<uint64_t>(T >> 40, 0x1fffu); };
                        uint32_t alts = 0;
                        for (int k = 0; k < 3; ++k) {
                            const int li = letter_index(t->subst_alt[row + k]);
                            if (li < 0) return fail(ctx, ISS_E_INVALID, "substitution alternatives use more than four distinct letters");
                            alts |= (uint32_t)li << (2 * k);
                        }
                        subst13[(size_t)tl * 2 * M.TP * 4 + ((size_t)(o * M.TP + pp) * 4 + bi)] =
                            d13(t->subst_thr[row]) | (d13(t->subst_thr[row + 1]) << 13) | (alts << 26);
                    }
        M.alt_letters = (uint32_t)letters[0] | ((uint32_t)letters[1] << 8) | ((uint32_t)letters[2] << 16) | ((uint32_t)letters[3] << 24);
    }
    {   // edit scripts (k_indel_script -> k_main): four 16-byte rows per tile and group of 8 iterations
        M.sc_gpt = ((M.TS + 3) / 4 + 7) / 8;
        M.sc_stride = M.n_tiles * M.sc_gpt * 64;
        M.ins_plain = 1;
        for (size_t i = 0; i < (size_t)2 * RL * 4; ++i) {
            const uint8_t c = t->ins_letter[i];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') M.ins_plain = 0;
        }
    }
    std::vector<uint64_t> del_max((size_t)2 * RL);
    for (int o = 0; o < 2; ++o)
        for (int n = 0; n < RL; ++n) {
            const size_t e = (size_t)o * RL + n;
            uint64_t dm = 0;
            for (int x = 0; x < 4; ++x) dm = std::max(dm, t->del_thr[e * 4 + x]);
            del_max[e] = dm;
        }
    // the indel event process (iss_kernels.hip.h indel_events): per mate the slots
    // 5 n + k of the loop steps n <= RL-2 (__init__.py:187) -- k = 0..3 the insertion tests, k = 4 the deletion test with
    // the largest of its four thresholds -- their survival products in 0.64 fixed point (floor after every factor; a
    // new segment after a slot that leaves less than 2^-16) and the last slot of each slot's segment
    const int ev_ns = 5 * (RL - 1);
    if (ev_ns > 0xffff) return fail(ctx, ISS_E_INVALID, "read_length too large for the indel event tables");
    std::vector<uint64_t> ev_S((size_t)2 * ev_ns), ev_T((size_t)2 * ev_ns);
    std::vector<uint16_t> ev_E((size_t)2 * ev_ns);
    bool any_indel = false;
    for (int o = 0; o < 2; ++o) {
        uint64_t prev = iss::EV_ONE;
        int seg_start = 0;
        for (int sl = 0; sl < ev_ns; ++sl) {
            const int n = sl / 5, k = sl % 5;
            const uint64_t T = k < 4 ? t->ins_thr[((size_t)o * RL + n) * 4 + k] : del_max[(size_t)o * RL + n];
            if (T > ((uint64_t)1 << 53)) return fail(ctx, ISS_E_INVALID, "an indel threshold exceeds 2^53");
            any_indel |= T != 0;
            const uint64_t cur = (uint64_t)(((unsigned __int128)prev * (((uint64_t)1 << 53) - T)) >> 53);
            ev_T[(size_t)o * ev_ns + sl] = T;
            ev_S[(size_t)o * ev_ns + sl] = cur;
            if (cur < ((uint64_t)1 << 48) || sl == ev_ns - 1) {  // the segment ends here
                for (int q = seg_start; q <= sl; ++q) ev_E[(size_t)o * ev_ns + q] = (uint16_t)sl;
                seg_start = sl + 1;
                prev = iss::EV_ONE;
            } else {
                prev = cur;
            }
        }
    }
    M.ev_ns = ev_ns;
    M.n_scan = any_indel ? 1 : 0;
    {   // how often a read has an event at all: models where that is rare (the shipped NovaSeq / HiSeq profiles: a few reads in
        // 10^5) keep k_main's plain variant and hand those reads to the one-wavefront-per-read kernel
        double p_any = 0;
        for (int o = 0; o < 2; ++o) {
            double none = 1.0;
            for (int sl = 0; sl < ev_ns; ++sl) none *= 1.0 - (double)ev_T[(size_t)o * ev_ns + sl] / 9007199254740992.0;
            p_any = std::max(p_any, 1.0 - none);
        }
        M.p_read_event = (float)p_any;
        ctx->light = p_any < ctx->light_below;
    }
    // k_mt_resolve tables: un-merged 16-bit leading digits per (orientation, bin slot, position) -- a row of n_q
    // digits padded to an odd number of words -- and 27-bit leading parts of the indel thresholds
    M.mt_row_w = (nq + 2) / 2;  // >= one 0xffff padding digit after the n_q digits
    if (!(M.mt_row_w & 1)) ++M.mt_row_w;
    std::vector<uint16_t> mt_rows((size_t)2 * M.NB * RL * M.mt_row_w * 2, 0xffffu);
    for (int o = 0; o < 2; ++o)
        for (int sl = 0; sl < M.NB; ++sl)
            for (int p = 0; p < RL; ++p) {
                const uint64_t *row = t->q_thr + ((size_t)(o * 4 + M.slot_bin[o * 4 + sl]) * RL + p) * nq;
                uint16_t *dst = mt_rows.data() + ((size_t)(o * M.NB + sl) * RL + p) * M.mt_row_w * 2;
                for (int i = 0; i < nq; ++i) dst[i] = (uint16_t)std::min<uint64_t>(row[i] >> 37, 0xffffu);
            }
    // mt_lim = ceil(thr / 2^26): the test `m < thr` (m a 53-bit numerator, thr the integer threshold of DESIGN.md section 3) can
    // only fire if the 27 leading bits of m are BELOW it -- 0 for a probability of zero: such a test is never a candidate (round 5:
    // `leading bits <= thr >> 26` made every one of the 1 500 zero-probability tests of a NovaSeq pair a candidate with
    // probability 2^-27 -- 1.1e-5 per pair, most of the pairs the resolver handed to the walker)
    auto lim_of = [](uint64_t thr) { return (uint32_t)((thr + (((uint64_t)1 << 26) - 1)) >> 26); };
    std::vector<uint32_t> mt_lim((size_t)2 * RL * 5);
    for (size_t e = 0; e < (size_t)2 * RL; ++e) {
        for (int x = 0; x < 4; ++x) mt_lim[e * 5 + x] = lim_of(t->ins_thr[e * 4 + x]);
        mt_lim[e * 5 + 4] = lim_of(del_max[e]);
    }
    {   // expected share of pairs the resolver hands to the sequential walker (an indel candidate in either mate)
        double rate = 0;
        for (size_t e = 0; e < (size_t)2 * RL; ++e)
            for (int x = 0; x < 5; ++x) rate += (double)mt_lim[e * 5 + x] / 134217728.0;
        ctx->mt_bounce_rate = rate;
    }
    int rc = 0;
    auto *tr = &ctx->model_allocs;
#define UP(field, src, n, T) if ((rc = upload<T>(ctx, src, n, const_cast<T **>(&M.field), tr))) return rc
    UP(isize_thr, t->isize_thr, (size_t)t->n_isize, uint64_t);
    UP(bin_thr, t->bin_thr, 8, uint64_t);
    UP(q_thr, t->q_thr, n_qthr, uint64_t);
    UP(qrows, qrows.data(), qrows.size(), uint32_t);
    UP(subst13, subst13.data(), subst13.size(), uint32_t);
    UP(subst_thr, t->subst_thr, (size_t)2 * RL * 12, uint64_t);
    UP(subst_alt, t->subst_alt, (size_t)2 * RL * 12, uint8_t);
    UP(ins_thr, t->ins_thr, (size_t)2 * RL * 4, uint64_t);
    UP(ins_letter, t->ins_letter, (size_t)2 * RL * 4, uint8_t);
    UP(del_thr, t->del_thr, (size_t)2 * RL * 4, uint64_t);
    UP(mut_thr, t->mut_thr, (size_t)nq + 1, uint64_t);
    UP(ev_S, ev_S.data(), ev_S.size(), uint64_t);
    UP(ev_E, ev_E.data(), ev_E.size(), uint16_t);
    UP(ev_T, ev_T.data(), ev_T.size(), uint64_t);
    UP(mt_rows, mt_rows.data(), mt_rows.size(), uint16_t);
    UP(mt_lim, mt_lim.data(), mt_lim.size(), uint32_t);
#undef UP
    ctx->have_model = true;
    free_outputs(ctx);  // pitch may have changed
    return 0;
}

int iss_genome_upload(iss_ctx *ctx, const uint8_t *ascii, int64_t length, int32_t *genome_id) {
    if (!ctx || !ascii || !genome_id) return fail(ctx, ISS_E_INVALID, "iss_genome_upload: NULL argument");
    // (records of 2^31 - 1 bases and more: the reference spills them to a memmap, generator.py:313-331; here coordinates are
    //  36-bit and word offsets into the packed genome 32-bit -- iss::MAX_RECORD, on both RNG paths since round 5)
    if (length < 1 || length > iss::MAX_RECORD) return fail(ctx, ISS_E_INVALID, "genome length must be in [1, 2^34 - 4096]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ASCII -> HBM, then packed on the device (k_pack_genome).  One readable padding word in front
    // (k_main's funnel shifts touch positions >= -3) and three behind.
    const size_t n_mk = (size_t)(length + 31) / 32, n_pk = 2 * n_mk;
    Genome G;
    G.L = length;
    unsigned long long *status = reinterpret_cast<unsigned long long *>(ctx->fix_count) + 24;  // 3 words at +192 B
    if (length <= SMALL_RECORD) {
        // small record: letters checked here (no wait for the device), buffers cut from the arena
        bool exceptions = false;
        for (int64_t i = 0; i < length; ++i) {
            const int cls = letter_class(ascii[i]);
            if (cls == 0) {
                int64_t bad = 0;
                for (int64_t j = i; j < length; ++j) bad += letter_class(ascii[j]) == 0;
                char buf[200];
                snprintf(buf, sizeof buf, "genome letter 0x%02x at offset %llu is outside the rev_comp alphabet (%llu such letters; "
                         "the reference raises KeyError, iss/util.py:90)", ascii[i], (unsigned long long)i, (unsigned long long)bad);
                return fail(ctx, ISS_E_INVALID, buf);
            }
            exceptions |= cls == 2;
        }
        hipError_t he = hipSuccess;
        uint8_t *blk = ctx->arena.take((n_pk + PK_PAD) * 4 + 256 + (n_mk + 4) * 4 + 256 + (size_t)length, &he);
        if (!blk) return fail(ctx, ISS_E_HIP, std::string("genome upload: ") + hipGetErrorString(he));
        const size_t pk_bytes = ((n_pk + PK_PAD) * 4 + 255) & ~(size_t)255, mk_bytes = ((n_mk + 4) * 4 + 255) & ~(size_t)255;
        G.packed_alloc = reinterpret_cast<uint32_t *>(blk);
        G.mask_alloc = reinterpret_cast<uint32_t *>(blk + pk_bytes);
        G.ascii = blk + pk_bytes + mk_bytes;
        G.in_arena = true;
        // a synchronous copy: the caller's buffer may go away as soon as this call returns, and nothing waits for the
        // stream here any more (the slice is fresh memory, so no earlier launch can be using it)
        he = hipMemcpy(G.ascii, ascii, (size_t)length, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemsetAsync(blk, 0, pk_bytes + mk_bytes, ctx->stream);
        if (he != hipSuccess) return fail(ctx, ISS_E_HIP, std::string("genome upload: ") + hipGetErrorString(he));
        hipLaunchKernelGGL(iss::k_pack_genome, dim3((unsigned)((n_mk + 255) / 256)), dim3(256), 0, ctx->stream, G.ascii, length,
                           G.packed_alloc + 1, G.mask_alloc + 1, status);
        G.has_exceptions = exceptions;
    } else {
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, (n_pk + PK_PAD) * sizeof(uint32_t)));
        G.packed_alloc = static_cast<uint32_t *>(p);
        HIP_TRY(ctx, hipMalloc(&p, (n_mk + 4) * sizeof(uint32_t)));
        G.mask_alloc = static_cast<uint32_t *>(p);
        HIP_TRY(ctx, hipMalloc(&p, (size_t)length));
        G.ascii = static_cast<uint8_t *>(p);
        auto release = [&]() { (void)hipFree(G.packed_alloc); (void)hipFree(G.mask_alloc); (void)hipFree(G.ascii); };
        const unsigned long long init[3] = {0ull, (unsigned long long)length, 0ull};
        hipError_t he = hipMemcpyAsync(G.ascii, ascii, (size_t)length, hipMemcpyHostToDevice, ctx->stream);
        if (he == hipSuccess) he = hipMemsetAsync(G.packed_alloc, 0, (n_pk + PK_PAD) * sizeof(uint32_t), ctx->stream);
        if (he == hipSuccess) he = hipMemsetAsync(G.mask_alloc, 0, (n_mk + 4) * sizeof(uint32_t), ctx->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(status, init, sizeof init, hipMemcpyHostToDevice, ctx->stream);
        if (he != hipSuccess) { release(); return fail(ctx, ISS_E_HIP, std::string("genome upload: ") + hipGetErrorString(he)); }
        hipLaunchKernelGGL(iss::k_pack_genome, dim3((unsigned)((n_mk + 255) / 256)), dim3(256), 0, ctx->stream, G.ascii, length,
                           G.packed_alloc + 1, G.mask_alloc + 1, status);
        unsigned long long res[3] = {0, 0, 0};
        he = hipMemcpyAsync(res, status, sizeof res, hipMemcpyDeviceToHost, ctx->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) { release(); return fail(ctx, ISS_E_HIP, std::string("genome pack: ") + hipGetErrorString(he)); }
        if (res[0]) {
            release();
            char buf[200];
            snprintf(buf, sizeof buf, "genome letter 0x%02x at offset %llu is outside the rev_comp alphabet (%llu such letters; "
                     "the reference raises KeyError, iss/util.py:90)", ascii[res[1]], res[1], res[0]);
            return fail(ctx, ISS_E_INVALID, buf);
        }
        G.has_exceptions = res[2] != 0;
    }
    G.packed = G.packed_alloc + 1;
    G.mask = G.mask_alloc + 1;
    // (the packing kernel / copies of this record may still run on the main stream: k_setup, on the setup stream, waits for them)
    HIP_TRY(ctx, hipEventRecord(ctx->ev_inputs, ctx->stream));
    ctx->inputs_pending = true;
    ctx->genomes.push_back(G);
    *genome_id = (int32_t)ctx->genomes.size() - 1;
    return 0;
}

int iss_genome_upload_packed(iss_ctx *ctx, const uint32_t *codes, int64_t length, int32_t codes_on_device, int32_t *genome_id) {
    if (!ctx || !codes || !genome_id) return fail(ctx, ISS_E_INVALID, "iss_genome_upload_packed: NULL argument");
    // (records of 2^31 - 1 bases and more: the reference spills them to a memmap, generator.py:313-331; here coordinates are
    //  36-bit and word offsets into the packed genome 32-bit -- iss::MAX_RECORD, on both RNG paths since round 5)
    if (length < 1 || length > iss::MAX_RECORD) return fail(ctx, ISS_E_INVALID, "genome length must be in [1, 2^34 - 4096]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_mk = (size_t)(length + 31) / 32, n_pk = 2 * n_mk, n_in = (size_t)(length + 15) / 16;
    Genome G;
    G.L = length;
    auto release = [&]() {  // (whatever was allocated so far: hipFree(nullptr) is a no-op)
        (void)hipFree(G.packed_alloc); (void)hipFree(G.mask_alloc); (void)hipFree(G.ascii);
        G.packed_alloc = G.mask_alloc = nullptr; G.ascii = nullptr;
    };
    void *p = nullptr;
    hipError_t he = hipMalloc(&p, (n_pk + PK_PAD) * sizeof(uint32_t));
    if (he == hipSuccess) { G.packed_alloc = static_cast<uint32_t *>(p); he = hipMalloc(&p, (n_mk + 4) * sizeof(uint32_t)); }
    if (he == hipSuccess) { G.mask_alloc = static_cast<uint32_t *>(p); he = hipMalloc(&p, (size_t)length); }
    if (he == hipSuccess) G.ascii = static_cast<uint8_t *>(p);
    if (he != hipSuccess) { release(); return fail(ctx, he == hipErrorOutOfMemory ? ISS_E_NOMEM : ISS_E_HIP, std::string("genome upload: ") + hipGetErrorString(he)); }
    he = hipMemsetAsync(G.packed_alloc, 0, (n_pk + PK_PAD) * sizeof(uint32_t), ctx->stream);
    if (he == hipSuccess) he = hipMemsetAsync(G.mask_alloc, 0, (n_mk + 4) * sizeof(uint32_t), ctx->stream);
    if (he == hipSuccess)
        he = hipMemcpyAsync(G.packed_alloc + 1, codes, n_in * sizeof(uint32_t), codes_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                            ctx->stream);
    if (he != hipSuccess) { release(); return fail(ctx, ISS_E_HIP, std::string("genome upload: ") + hipGetErrorString(he)); }
    // the ASCII copy (exact path, FASTA-free consumers) from the codes; codes past the end are cleared
    hipLaunchKernelGGL(iss::k_unpack_genome, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, ctx->stream, G.packed_alloc + 1, length,
                       G.ascii);
    he = hipStreamSynchronize(ctx->stream);  // the caller's buffer may go away once this call returns
    if (he != hipSuccess) { release(); return fail(ctx, ISS_E_HIP, std::string("genome unpack: ") + hipGetErrorString(he)); }
    G.has_exceptions = false;
    G.packed = G.packed_alloc + 1;
    G.mask = G.mask_alloc + 1;
    // (the packing kernel / copies of this record may still run on the main stream: k_setup, on the setup stream, waits for them)
    HIP_TRY(ctx, hipEventRecord(ctx->ev_inputs, ctx->stream));
    ctx->inputs_pending = true;
    ctx->genomes.push_back(G);
    *genome_id = (int32_t)ctx->genomes.size() - 1;
    return 0;
}

static void free_community(iss_ctx *ctx);
static void free_item_tables(iss_ctx *ctx);

int iss_genome_clear(iss_ctx *ctx) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    (void)sync_all(ctx);
    for (auto &G : ctx->genomes)
        if (!G.in_arena) { (void)hipFree(G.packed_alloc); (void)hipFree(G.mask_alloc); (void)hipFree(G.ascii); }
    ctx->genomes.clear();
    ctx->arena.clear();
    free_community(ctx);
    free_item_tables(ctx);
    return 0;
}

int iss_output_reserve(iss_ctx *ctx, int64_t capacity_pairs) {
    if (!ctx || !ctx->have_model) return fail(ctx, ISS_E_INVALID, "iss_output_reserve: upload a model first");
    if (capacity_pairs < 1) return fail(ctx, ISS_E_INVALID, "capacity must be >= 1");
    if (capacity_pairs <= ctx->capacity) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = fastq_flush(ctx); if (rc_) return rc_; }
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    free_outputs(ctx);
    void *q = nullptr;
    // (every buffer of the reservation through one checked allocation: out of memory frees what the call has allocated so far and
    //  is reported as ISS_E_NOMEM with the reservation's footprint -- the edit scripts of a five-tile heavy model are 1.3 KB per
    //  pair, more than its rows)
    const bool heavy_ = ctx->M.n_scan > 0 && !ctx->light;
    const double per_pair = (double)ctx->M.row + 3.0 * sizeof(iss::PairDesc) + 2.0 * 12.0 +
                            (heavy_ ? 2.0 : 1.0) * (8.0 + 8.0 * iss::EV_K + 32.0 + 16.0) + (heavy_ ? 4.0 * ctx->M.sc_stride : 0.0);
#define ISS_RES_ALLOC(bytes)                                                                                                        \
    do {                                                                                                                            \
        const hipError_t e_ = hipMalloc(&q, (bytes));                                                                               \
        if (e_ != hipSuccess) {                                                                                                     \
            (void)hipGetLastError();                                                                                                \
            free_outputs(ctx);                                                                                                      \
            char msg_[256];                                                                                                         \
            snprintf(msg_, sizeof msg_, "iss_output_reserve: %lld pairs need %.1f GB of HBM (%.0f B per pair%s): %s", (long long)capacity_pairs, \
                     per_pair * (double)capacity_pairs / 1e9, per_pair, heavy_ ? ", edit scripts included" : "", hipGetErrorString(e_));  \
            return fail(ctx, e_ == hipErrorOutOfMemory ? ISS_E_NOMEM : ISS_E_HIP, msg_);                                           \
        }                                                                                                                           \
    } while (0)
    // (plain hipMalloc: physically contiguous rows -- hipExtMallocWithFlags(hipDeviceMallocContiguous) -- were measured at 1.82-1.88
    //  instead of 1.25-1.34 ms per step of the default bench, whatever the grid)
    //  instead of 1.25-1.34 ms per step of the default bench, whatever the grid; rows mapped from separately created physical
    //  chunks -- hipMemCreate / hipMemMap, 64 KB to 16 MB, in order or shuffled -- at 1.23-1.9: no layout helped on every box)
    ISS_RES_ALLOC((size_t)ctx->M.row * (size_t)capacity_pairs);
    for (int k = 0; k < 4; ++k) ctx->out[k] = static_cast<uint8_t *>(q) + iss::row_array_off(k);
    for (int k = 0; k < 2; ++k) {  // (two sets: k_setup of a call runs beside the kernels of the call before)
        ISS_RES_ALLOC(sizeof(iss::PairDesc) * (size_t)capacity_pairs);
        ctx->desc_buf[k] = static_cast<iss::PairDesc *>(q);
        ISS_RES_ALLOC(sizeof(uint32_t) * (size_t)capacity_pairs);
        ctx->flags_buf[k] = static_cast<uint32_t *>(q);
        ISS_RES_ALLOC(sizeof(uint32_t) * 2 * (size_t)capacity_pairs);
        ctx->fixl_buf[k] = static_cast<uint32_t *>(q);
    }
    ISS_RES_ALLOC(sizeof(iss::PairDesc) * (size_t)capacity_pairs);
    ctx->desc = static_cast<iss::PairDesc *>(q);  // what the host reads (iss_output_download_coords) and the MT kernels write
    ctx->flags = ctx->flags_buf[0]; ctx->fix_list = ctx->fixl_buf[0];
    const bool heavy = ctx->M.n_scan > 0 && !ctx->light;
    for (int k = 0; k < (heavy ? 2 : 1); ++k) {
        ISS_RES_ALLOC(sizeof(uint32_t) * 2 * (size_t)capacity_pairs);
        ctx->ev_count[k] = static_cast<uint32_t *>(q);
        ISS_RES_ALLOC(sizeof(uint32_t) * 2 * iss::EV_K * (size_t)capacity_pairs);
        ctx->ev_list[k] = static_cast<uint32_t *>(q);
        ISS_RES_ALLOC(sizeof(uint4) * 2 * (size_t)capacity_pairs);
        ctx->read_list[k] = static_cast<uint4 *>(q);
        ISS_RES_ALLOC(sizeof(uint2) * 2 * (size_t)capacity_pairs);
        ctx->read_list1[k] = static_cast<uint2 *>(q);
    }
    if (!heavy) { ctx->ev_count[1] = ctx->ev_count[0]; ctx->ev_list[1] = ctx->ev_list[0]; ctx->read_list[1] = ctx->read_list[0]; ctx->read_list1[1] = ctx->read_list1[0]; }
    if (heavy)  // the edit scripts of the reads with an event (sparse: a read's slot is written only if it has one)
        for (int k = 0; k < 2; ++k) {
            ISS_RES_ALLOC((size_t)ctx->M.sc_stride * 2 * (size_t)capacity_pairs);
            ctx->script[k] = static_cast<uint8_t *>(q);
        }
#undef ISS_RES_ALLOC
    ctx->capacity = capacity_pairs;
    return 0;
}

int iss_output_pitch(const iss_ctx *ctx) { return (ctx && ctx->have_model) ? ctx->M.pitch : ISS_E_INVALID; }
int iss_output_row(const iss_ctx *ctx) { return (ctx && ctx->have_model) ? ctx->M.row : ISS_E_INVALID; }

int iss_output_device_ptrs(const iss_ctx *ctx, void **a, void **b, void **c, void **d) {
    if (!ctx || !ctx->capacity) return ISS_E_INVALID;
    if (a) *a = ctx->out[0];
    if (b) *b = ctx->out[1];
    if (c) *c = ctx->out[2];
    if (d) *d = ctx->out[3];
    return 0;
}

static int generate_core(iss_ctx *ctx, const iss::DevGenome &dg, bool any_exceptions, const iss::BatchItem *items,
                         const int64_t *item_first, int32_t n_items, int64_t n_pairs, uint64_t first_ordinal, uint64_t seed,
                         int32_t sequence_type, int32_t gc_bias, int64_t out_first_pair);

// Do the kernels in front of k_main (k_setup, k_indel_scan, k_indel_script) of a call run on the setup stream, beside the kernels
// of the call before?  Not with custom fragment lengths (the host reads k_setup's results back), not while every kernel is
// timed, and not when k_indel_script appends --store_mutations rows (the call clears the row buffer on the main stream).
static bool setup_runs_ahead(const iss_ctx *ctx) {
    const bool heavy = ctx->M.n_scan > 0 && !ctx->light;
    return ctx->setup_ahead && !ctx->has_frag && !ctx->timing_all && !(heavy && ctx->d_pmut);
}

int iss_generate(iss_ctx *ctx, int32_t genome_id, int64_t n_pairs, uint64_t first_ordinal, uint64_t seed,
                 int32_t sequence_type, int32_t gc_bias, int64_t out_first_pair) {
    if (!ctx || !ctx->have_model) return fail(ctx, ISS_E_INVALID, "iss_generate: upload a model first");
    if (genome_id < 0 || genome_id >= (int32_t)ctx->genomes.size()) return fail(ctx, ISS_E_INVALID, "unknown genome id");
    if (sequence_type != ISS_SEQ_METAGENOMICS && sequence_type != ISS_SEQ_AMPLICON)
        return fail(ctx, ISS_E_INVALID, "sequence type is not supported");  // generator.py:139, 171
    if (n_pairs < 0 || out_first_pair < 0 || out_first_pair + n_pairs > ctx->capacity)
        return fail(ctx, ISS_E_INVALID, "output rows out of the reserved range");
    const Genome &G = ctx->genomes[genome_id];
    const iss::DevModel &M = ctx->M;
    if (!(M.RL < G.L)) return fail(ctx, ISS_E_SHORT_RECORD, "record shorter than read length for this ErrorModel");
    if (n_pairs == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const iss::DevGenome dg{G.packed, G.mask, G.ascii, G.L, G.has_exceptions ? 1 : 0};
    return generate_core(ctx, dg, G.has_exceptions, nullptr, nullptr, 0, n_pairs, first_ordinal, seed, sequence_type, gc_bias,
                         out_first_pair);
}

// The launches of one generate call: `dg` is the record, or (items != NULL) the arena holding the records of a batch.
// Per chunk: k_setup [k_indel_scan, k_indel_script: models whose reads often have indels] -> k_main -> k_indel_fixup.  The
// kernels in front of k_main read nothing the call before writes and write double-buffered sets (descriptors, flags, fix-up
// lists, event lists, scripts): they run on the setup stream, beside the kernels of the call (or chunk) before.
static int generate_core(iss_ctx *ctx, const iss::DevGenome &dg, bool any_exceptions, const iss::BatchItem *items,
                         const int64_t *item_first, int32_t n_items, int64_t n_pairs, uint64_t first_ordinal, uint64_t seed,
                         int32_t sequence_type, int32_t gc_bias, int64_t out_first_pair) {
    const iss::DevModel &M = ctx->M;
    read_switches(ctx);
    const size_t lds_bytes = main_lds_bytes(M);
    const bool heavy = M.n_scan > 0 && !ctx->light;  // reads with an indel event are common: scan + edit scripts + k_main<.., INDEL>
    // k_main's deferred queue: 13 bits for (pass of a workgroup, iteration of the pass); the tile with the fewest
    // workgroups (a short last tile) makes the most passes
    const unsigned it_max = ((unsigned)M.TS + 3u) / 4u - 1u;
    unsigned it_bits = 0;
    while ((1u << it_bits) <= it_max && it_max) ++it_bits;
    const int64_t max_passes = ((int64_t)1 << (13 - it_bits)) - 1;
    const unsigned budget_all = std::min(std::min((unsigned)ctx->n_cu, ctx->max_main_grid), ctx->env_main_wgs ? (unsigned)ctx->env_main_wgs : ~0u);  // ONE 1024-lane workgroup per CU (k_main: 4 wavefronts / SIMD)
    unsigned weight_all = 0;
    for (int t = 0; t < M.n_tiles; ++t) weight_all += 1u + (unsigned)(std::min(M.TS, M.S - t * M.TS) + 3) / 4u;
    const unsigned last_weight = 1u + (unsigned)(M.S - (M.n_tiles - 1) * M.TS + 3) / 4u;
    const unsigned min_tile_wg = std::max(1u, (unsigned)((uint64_t)budget_all * last_weight / weight_all));
    // the pass number of a workgroup (>= 1 workgroup per tile, 256 pairs per pass); 32 bits for the read numbers of k_indel_scan
    // and for k_main's pair numbers (row offsets are 64-bit since round 5: 5 M MiSeq pairs of 1 280-byte rows are one launch)
    const int64_t max_chunk = std::max<int64_t>(1, std::min<int64_t>(((int64_t)1 << 31) / std::max(M.n_scan, 1) - iss::MAIN_PAIRS,
                                                                    max_passes * iss::MAIN_PAIRS * min_tile_wg));
    // Pairs per launch.  The address limits above allow 2^31 reads, but k_main's own time per pair rises with the launch: BASELINE
    // configs[3]'s shape on one GPU (50 M HiSeq pairs per step), interleaved on one box (profiles/r06_ab_runs.txt): ONE launch
    // 12.9-13.2 ms of k_main (3.7-3.8 x 10^9 pairs/s), launches of 12.5 M or 5 M pairs 11.7-12.0 ms (4.1 x 10^9) -- round 5 had
    // dropped the <= 4 GB chunks when the row offsets became 64-bit, and that was the 7 % it lost on this shape; k_main_g:
    // 11.4-11.5 / 10.8 / 10.8 ms.  (k_setup of chunk k + 1 runs beside k_main of chunk k either way.)
    const int64_t chunk_pairs = std::min(max_chunk, ctx->env_chunk_pairs ? ctx->env_chunk_pairs : MAIN_CHUNK_PAIRS);
    if (ctx->d_pmut) {  // rows of THIS call only
        ctx->d_pmut_count = reinterpret_cast<uint32_t *>(ctx->fix_count) + 60;  // +240 B of the scratch block
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_pmut, 0xff, (size_t)ctx->pmut_cap * sizeof(iss::MutRecord), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_pmut_count, 0, sizeof(uint32_t), ctx->stream));
    }
    ctx->last_row0 = out_first_pair;
    ctx->last_n = n_pairs;
    if (!items) { ctx->last_first.clear(); ctx->last_off.clear(); }
    // this call's set of descriptors / flags / lists; the setup-stream kernels start once the call before last -- the last
    // user of the set -- is done (custom fragment lengths: the host reads k_setup's results back: everything on one stream;
    // --store_mutations: the rows are cleared on the main stream above)
    const int par = (int)(ctx->call_seq++ & 1u);
    ctx->flags = ctx->flags_buf[par];
    ctx->fix_list = ctx->fixl_buf[par];
    const bool ahead = setup_runs_ahead(ctx);
    hipStream_t s_setup = ahead ? ctx->setup_stream : ctx->stream;
    hipStream_t s_main = ctx->stream;
    if (ahead) {
        if (ctx->ev_call_valid[par]) HIP_TRY(ctx, hipStreamWaitEvent(s_setup, ctx->ev_call_done[par], 0));
        if (ctx->inputs_pending) HIP_TRY(ctx, hipStreamWaitEvent(s_setup, ctx->ev_inputs, 0));  // (arena / table copies of this call)
    }
    ctx->inputs_pending = false;
    for (int64_t done = 0; done < n_pairs;) {
        const int64_t n = std::min(chunk_pairs, n_pairs - done);
        const int64_t row0 = out_first_pair + done;
        iss::RunArgs A{};
        A.n_pairs = n;
        A.first_ordinal = first_ordinal + (uint64_t)done;
        A.seed = seed;
        A.sequence_type = sequence_type;
        A.gc_bias = gc_bias ? 1 : 0;
        A.gc_thr = 8106479329266893ull;  // ceil(0.90 * 2^53), 0.90 being the f64 nearest to 0.9
        for (int k = 0; k < 4; ++k) A.out[k] = ctx->out[k] + (size_t)row0 * M.row;
        iss::PairDesc *desc = ctx->desc_buf[par] + row0;
        A.desc_out = ctx->desc + row0;
        uint32_t *flags = ctx->flags + row0;
        uint32_t *fix_list = ctx->fix_list + 2 * row0;
        TimedLaunch tl{};
        tl.has_scan = M.n_scan > 0 || ctx->has_frag;
        auto mark = [&](int k, hipStream_t st) -> hipError_t {
            if (!ctx->timing) return hipSuccess;
            if (ctx->timing_main_only && k != 1 && k != 2) return hipSuccess;  // (every event costs a bubble in the stream)
            hipError_t e = hipEventCreate(&tl.ev[k]);
            if (e != hipSuccess) return e;
            return hipEventRecord(tl.ev[k], st);
        };
        // fix-list / read-list counters of this chunk: rings of FIX_SLOTS counters.  The setup stream runs ahead of the main
        // stream: before a slot's counters are cleared for its next user, the chunk that used it last must be done with them
        // (its k_indel_fixup reads the fix-list counter on the main stream).  (The flags are cleared by k_setup itself.)
        const unsigned slot_i = (unsigned)(ctx->chunk_seq++ % FIX_SLOTS);
        uint32_t *counter = ctx->fix_count + slot_i;
        uint32_t *read_counter = ctx->read_count + (size_t)slot_i * 2 * iss::SCAN_MAX_WGS;  // (two per workgroup of k_indel_scan, all of them written by it)
        if (ahead && ctx->ev_slot_valid[slot_i]) HIP_TRY(ctx, hipStreamWaitEvent(s_setup, ctx->ev_slot_done[slot_i], 0));
        HIP_TRY(ctx, hipMemsetAsync(counter, 0, sizeof(uint32_t), s_setup));
        A.mut = ctx->d_pmut;
        A.mut_count = ctx->d_pmut_count;
        A.mut_cap = (uint32_t)ctx->pmut_cap;
        A.pair_base = done;
        A.items = items;
        A.item_first = item_first;
        A.n_items = n_items;
        A.flags = flags;
        A.fix_list = fix_list;
        A.fix_count = counter;
        A.ev_count = M.n_scan > 0 ? ctx->ev_count[par] + 2 * row0 : nullptr;
        A.ev_list = ctx->ev_list[par] + 2 * (size_t)iss::EV_K * row0;
        A.read_list = ctx->read_list[par] + 2 * row0;
        A.read_list1 = ctx->read_list1[par] + 2 * row0;
        A.read_count = read_counter;
        A.scan_wgs = (uint32_t)std::min<uint64_t>(std::min<uint64_t>((uint64_t)ctx->n_cu * 2, iss::SCAN_MAX_WGS), (2 * (uint64_t)n + iss::SCAN_THREADS - 1) / iss::SCAN_THREADS);
        A.light = ctx->light ? (iss::setup_lds_bytes(M.n_isize, M.ev_ns, true) <= (size_t)150 * 1024 ? 1 : 2) : 0;
        A.script = heavy ? ctx->script[par] + (size_t)2 * (size_t)row0 * (size_t)M.sc_stride : nullptr;
        A.has_frag = ctx->has_frag ? 1 : 0;
        A.frag_mu = ctx->frag_mu;
        A.frag_sd = ctx->frag_sd;
        A.frag_guard = ctx->mt_guard;
        if (ctx->has_frag) {
            if (ctx->amb_cap < n) {
                if (ctx->d_amb) (void)hipFree(ctx->d_amb);
                if (ctx->d_ov_pairs) (void)hipFree(ctx->d_ov_pairs);
                if (ctx->d_ov_frags) (void)hipFree(ctx->d_ov_frags);
                void *p = nullptr;
                HIP_TRY(ctx, hipMalloc(&p, (size_t)n * sizeof(iss::FragAmb)));
                ctx->d_amb = static_cast<iss::FragAmb *>(p);
                HIP_TRY(ctx, hipMalloc(&p, (size_t)n * sizeof(uint32_t)));
                ctx->d_ov_pairs = static_cast<uint32_t *>(p);
                HIP_TRY(ctx, hipMalloc(&p, (size_t)n * sizeof(int64_t)));
                ctx->d_ov_frags = static_cast<int64_t *>(p);
                ctx->amb_cap = n;
            }
            ctx->d_amb_count = reinterpret_cast<uint32_t *>(ctx->fix_count) + 56;  // +224 B of the 256-byte scratch block
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_amb_count, 0, sizeof(uint32_t), s_main));
            A.amb_list = ctx->d_amb;
            A.amb_count = ctx->d_amb_count;
        }
        tl.scan_first = heavy && !ahead;
        HIP_TRY(ctx, mark(0, s_setup));
        if (ahead && A.light == 1 && main_lds_bytes(M) + iss::setup_lds_bytes(M.n_isize, M.ev_ns, true) > (size_t)158 * 1024)
            A.light = 2;  // (k_main's tables leave no room for the event tables beside them: read in place, off the critical path)
        {
            const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 8 * (int64_t)ctx->n_cu);
            hipLaunchKernelGGL(iss::k_setup, dim3(blocks), dim3(256), iss::setup_lds_bytes(M.n_isize, M.ev_ns, A.light == 1 && M.n_scan > 0), s_setup, M, dg, A, desc);
        }
        if (ctx->has_frag) {  // (one stream: s_setup == s_main)
            // fragment lengths the device could not decide (|x - round(x)| < guard): libm on the host, then redo those pairs
            uint32_t n_amb = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&n_amb, ctx->d_amb_count, sizeof n_amb, hipMemcpyDeviceToHost, s_main));
            HIP_TRY(ctx, hipStreamSynchronize(s_main));
            if (n_amb) {
                std::vector<iss::FragAmb> amb(n_amb);
                HIP_TRY(ctx, hipMemcpy(amb.data(), ctx->d_amb, n_amb * sizeof(iss::FragAmb), hipMemcpyDeviceToHost));
                std::vector<uint32_t> pairs(n_amb);
                std::vector<int64_t> frags(n_amb);
                for (uint32_t k = 0; k < n_amb; ++k) {
                    pairs[k] = amb[k].pair;
                    frags[k] = host_int_normal(amb[k].x1, amb[k].x2, false, ctx->frag_mu, ctx->frag_sd);
                }
                HIP_TRY(ctx, hipMemcpy(ctx->d_ov_pairs, pairs.data(), n_amb * sizeof(uint32_t), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(ctx->d_ov_frags, frags.data(), n_amb * sizeof(int64_t), hipMemcpyHostToDevice));
                A.ov_pairs = ctx->d_ov_pairs;
                A.ov_frags = ctx->d_ov_frags;
                A.n_ov = n_amb;
                hipLaunchKernelGGL(iss::k_setup_override, dim3((n_amb + 63) / 64), dim3(64), 0, s_main, M, dg, A, desc);
            }
        }
        if (heavy) {
            // the event lists of all reads, one lane per read, then the edit scripts of the reads that have an event
            if (!ahead) HIP_TRY(ctx, mark(3, s_setup));
            hipLaunchKernelGGL(iss::k_indel_scan, dim3(A.scan_wgs), dim3(iss::SCAN_THREADS), iss::scan_lds_bytes(M.ev_ns), s_setup, M, A, desc);
            {
                const size_t lds = iss::script_lds_bytes(M.RL, M.pitch, false), lds1 = iss::script_lds_bytes(M.RL, M.pitch, true);
                const int64_t per_wg = (int64_t)iss::SC_WAVES * 64, per_wg1 = (int64_t)iss::SC_WAVES1 * 64;  // reads per workgroup pass; at most 2 n reads
                const dim3 grid((unsigned)std::min<int64_t>((int64_t)iss::SC_WGS_PER_CU * ctx->n_cu, (2 * n + per_wg - 1) / per_wg)), block(64 * iss::SC_WAVES);
                const dim3 grid1((unsigned)std::min<int64_t>((int64_t)iss::SC_WGS_PER_CU * ctx->n_cu, (2 * n + per_wg1 - 1) / per_wg1)), block1(64 * iss::SC_WAVES1);
                const bool narrow = iss::ap_ww(M.pitch) <= 12;  // (window words a lane prefetches in registers)
                // (two launches: the reads with one event step -- straight-line code --, then the reads with more)
                // On the setup stream the two run SIDE BY SIDE (the second on the auxiliary stream, forked behind the scan and joined
                // in front of k_main): both spend half of their time waiting for loads, and one workgroup of each fits a CU.
                hipStream_t s_multi = s_setup;
                if (ahead) {
                    s_multi = ctx->indel_stream;
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork[slot_i], s_setup));
                    HIP_TRY(ctx, hipStreamWaitEvent(s_multi, ctx->ev_fork[slot_i], 0));
                }
#define ISS_LAUNCH_SCRIPT(MUT, WW)                                                                                                     \
    do {                                                                                                                               \
        hipLaunchKernelGGL((iss::k_indel_script<MUT, WW, true>), grid1, block1, lds1, s_setup, M, dg, A, desc, ctx->stats);             \
        hipLaunchKernelGGL((iss::k_indel_script<MUT, WW, false>), grid, block, lds, s_multi, M, dg, A, desc, ctx->stats);               \
    } while (0)
                if (A.mut) { if (narrow) ISS_LAUNCH_SCRIPT(true, 12); else ISS_LAUNCH_SCRIPT(true, 26); }
                else { if (narrow) ISS_LAUNCH_SCRIPT(false, 12); else ISS_LAUNCH_SCRIPT(false, 26); }
#undef ISS_LAUNCH_SCRIPT
                if (ahead) {
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_join[slot_i], s_multi));
                    HIP_TRY(ctx, hipStreamWaitEvent(s_setup, ctx->ev_join[slot_i], 0));
                }
            }
            if (!ahead) HIP_TRY(ctx, mark(4, s_setup));
        }
        if (ahead) {  // k_main (and what follows it) waits for this chunk's setup-stream kernels
            if (ctx->timing && !ctx->timing_main_only) { HIP_TRY(ctx, hipEventCreate(&tl.ev[7])); HIP_TRY(ctx, hipEventRecord(tl.ev[7], s_setup)); }
            HIP_TRY(ctx, hipEventRecord(ctx->ev_setup_done[slot_i], s_setup));
            HIP_TRY(ctx, hipStreamWaitEvent(s_main, ctx->ev_setup_done[slot_i], 0));
        }
        HIP_TRY(ctx, mark(1, s_main));
        {
            const uint64_t passes = ((uint64_t)n + iss::MAIN_PAIRS - 1) / iss::MAIN_PAIRS;  // a workgroup pass = 256 pairs
            // persistent grid, split over the position tiles in proportion to the tiles' work per pass -- a fixed part
            // (descriptor, addresses) + one part per iteration of 4 superitems, whether or not all four lanes of a pair have one
            // (the last tile may be short) -- at most one workgroup per pass of a tile
            const unsigned wg_per_tile_cap = 8192;  // (tile_wg0 is 16 bits wide)
            unsigned total = 0;
            for (int t = 0; t < M.n_tiles; ++t) {
                const unsigned weight = 1u + (unsigned)(std::min(M.TS, M.S - t * M.TS) + 3) / 4u;
                unsigned w = std::max(1u, (unsigned)((uint64_t)budget_all * weight / weight_all));
                w = (unsigned)std::min<uint64_t>(std::min<uint64_t>(w, wg_per_tile_cap), passes);
                A.tile_wg0[t] = (uint16_t)total;
                total += w;
            }
            A.tile_wg0[M.n_tiles] = (uint16_t)total;
            const dim3 grid(total), block(iss::MAIN_THREADS);
            const bool plain = !any_exceptions && !ctx->has_frag;
#define ISS_LAUNCH_MAIN(MUT, PLAIN)                                                                                      \
    do {                                                                                                                 \
        if (heavy) hipLaunchKernelGGL((iss::k_main<MUT, PLAIN, true>), grid, block, lds_bytes, s_main, M, dg, A, desc);   \
        else hipLaunchKernelGGL((iss::k_main<MUT, PLAIN, false>), grid, block, lds_bytes, s_main, M, dg, A, desc);        \
    } while (0)
            // plain launches of models with a short pass: k_main_g -- the rows of a group of passes wait in registers until the
            // group's deferred bases are settled, their byte patches follow the rows out in time (iss_kernels.hip.h)
            bool grouped = false;
            if (plain && !heavy && !A.mut && ctx->env_group != 0) {
                const int ni = (M.TS + 3) / 4;
                const int np = main_group_passes(M, ni, ctx->env_group);
                const uint64_t span = (uint64_t)(np - 1) * total * iss::MAIN_PAIRS * (uint64_t)M.row + (uint64_t)iss::MAIN_PAIRS * M.row + 4096;
                const uint32_t min_round = ctx->env_group_min > 0 ? (uint32_t)ctx->env_group_min : MAIN_GROUP_MIN_ROUND;
#define ISS_MAIN_G_LAUNCH(NI_, NP_)                                                                                                   \
    if (!grouped && ni == NI_ && np == NP_ && span < ((uint64_t)1 << 32)) {                                                          \
        hipLaunchKernelGGL((iss::k_main_g<true, NI_, NP_>), grid, block, lds_bytes, s_main, M, dg, A, desc, min_round);                \
        ctx->main_kernel = "k_main_g<" #NI_ ", " #NP_ ">";                                                                           \
        grouped = true;                                                                                                              \
    }
                ISS_MAIN_G_LIST(ISS_MAIN_G_LAUNCH)
#undef ISS_MAIN_G_LAUNCH
            }
            if (!grouped) ctx->main_kernel = std::string("k_main<") + (A.mut ? "true" : "false") + ", " + (plain ? "true" : "false") + ", " + (heavy ? "true" : "false") + ">";
            if (grouped) { /* launched */ }
            else if (A.mut) { if (plain) ISS_LAUNCH_MAIN(true, true); else ISS_LAUNCH_MAIN(true, false); }
            else { if (plain) ISS_LAUNCH_MAIN(false, true); else ISS_LAUNCH_MAIN(false, false); }
#undef ISS_LAUNCH_MAIN
        }
        HIP_TRY(ctx, mark(2, s_main));
        if (M.n_scan > 0 || ctx->has_frag) {
            // the rest (irregular pairs, reads whose script does not fit, every read with an event of a light model): one
            // wavefront per read, behind k_main (it takes the read's phreds from the row and rewrites its letters)
            HIP_TRY(ctx, mark(5, s_main));
            const unsigned blocks = (unsigned)std::min<int64_t>(8 * ctx->n_cu, (2 * n + iss::FIX_WAVES - 1) / iss::FIX_WAVES);
            hipLaunchKernelGGL(iss::k_indel_fixup, dim3(blocks), dim3(64 * iss::FIX_WAVES), iss::fix_lds_bytes(M.RL), s_main, M, dg, A, desc,
                               fix_list, counter, ctx->stats);
            HIP_TRY(ctx, mark(6, s_main));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_slot_done[slot_i], s_main));
        ctx->ev_slot_valid[slot_i] = true;
        HIP_TRY(ctx, hipGetLastError());
        if (ctx->timing) ctx->timed.push_back(tl);
        done += n;
    }
    ctx->n_launches += 1;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_call_done[par], ctx->stream));
    ctx->ev_call_valid[par] = true;
    return 0;
}

int iss_synchronize(iss_ctx *ctx) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    return sync_all(ctx);
}

int iss_output_download(iss_ctx *ctx, int64_t first_pair, int64_t n_pairs, uint8_t *r1_base, uint8_t *r1_qual,
                        uint8_t *r2_base, uint8_t *r2_qual) {
    if (!ctx || first_pair < 0 || n_pairs < 0 || first_pair + n_pairs > ctx->capacity)
        return fail(ctx, ISS_E_INVALID, "iss_output_download: rows out of range");
    uint8_t *host[4] = {r1_base, r1_qual, r2_base, r2_qual};
    const size_t pitch = (size_t)ctx->M.pitch;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    if (!n_pairs) return 0;
    // the device rows are interleaved (iss::xp): four plain [n_pairs][pitch] arrays are formed on the device, then copied
    const size_t need = 4 * pitch * (size_t)n_pairs;
    if (ctx->stage_cap < need) {
        if (ctx->d_stage) (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr; ctx->stage_cap = 0;
        void *q = nullptr;
        HIP_TRY(ctx, hipMalloc(&q, need));
        ctx->d_stage = static_cast<uint8_t *>(q);
        ctx->stage_cap = need;
    }
    {
        hipLaunchKernelGGL(iss::k_rows_to_arrays, dim3((unsigned)((n_pairs + 3) / 4)), dim3(64, 4), 0, ctx->stream,
                           ctx->out[0] + (size_t)first_pair * ctx->M.row, ctx->d_stage, n_pairs, ctx->M.S, ctx->M.row);
        HIP_TRY(ctx, hipGetLastError());
    }
    for (int k = 0; k < 4; ++k)
        if (host[k])
            HIP_TRY(ctx, hipMemcpyAsync(host[k], ctx->d_stage + (size_t)k * pitch * (size_t)n_pairs, pitch * (size_t)n_pairs,
                                        hipMemcpyDeviceToHost, ctx->stream));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    return 0;
}

static void free_item_tables(iss_ctx *ctx) {
    for (int k = 0; k < 2; ++k) {
        if (ctx->d_items[k]) (void)hipFree(ctx->d_items[k]);
        if (ctx->d_item_first[k]) (void)hipFree(ctx->d_item_first[k]);
        if (ctx->h_items[k]) (void)hipHostFree(ctx->h_items[k]);
        if (ctx->h_item_first[k]) (void)hipHostFree(ctx->h_item_first[k]);
        if (ctx->ev_items[k]) (void)hipEventDestroy(ctx->ev_items[k]);
        ctx->d_items[k] = ctx->h_items[k] = nullptr;
        ctx->d_item_first[k] = ctx->h_item_first[k] = nullptr;
        ctx->ev_items[k] = nullptr;
    }
    ctx->d_items_cap = 0;
}

static void free_community(iss_ctx *ctx) {
    if (ctx->comm_packed) (void)hipFree(ctx->comm_packed);
    if (ctx->comm_mask) (void)hipFree(ctx->comm_mask);
    if (ctx->comm_ascii) (void)hipFree(ctx->comm_ascii);
    ctx->comm_packed = ctx->comm_mask = nullptr;
    ctx->comm_ascii = nullptr;
    ctx->comm_cap = 0;
    ctx->comm_ids.clear();
    ctx->comm_items.clear();
}

int iss_generate_batch(iss_ctx *ctx, int32_t n_items, const int32_t *genome_ids, const int64_t *n_pairs, uint64_t first_ordinal,
                       uint64_t seed, int32_t sequence_type, int32_t gc_bias, int64_t out_first_pair) {
    if (!ctx || !ctx->have_model) return fail(ctx, ISS_E_INVALID, "iss_generate_batch: upload a model first");
    if (n_items < 0 || (n_items && (!genome_ids || !n_pairs))) return fail(ctx, ISS_E_INVALID, "iss_generate_batch: bad argument");
    if (sequence_type != ISS_SEQ_METAGENOMICS && sequence_type != ISS_SEQ_AMPLICON)
        return fail(ctx, ISS_E_INVALID, "sequence type is not supported");
    const iss::DevModel &M = ctx->M;
    int64_t total = 0;
    std::vector<int64_t> first((size_t)n_items + 1, 0);
    for (int32_t k = 0; k < n_items; ++k) {
        if (genome_ids[k] < 0 || genome_ids[k] >= (int32_t)ctx->genomes.size()) return fail(ctx, ISS_E_INVALID, "unknown genome id");
        if (n_pairs[k] < 0) return fail(ctx, ISS_E_INVALID, "negative pair count");
        if (!(M.RL < ctx->genomes[genome_ids[k]].L))
            return fail(ctx, ISS_E_SHORT_RECORD, "record shorter than read length for this ErrorModel");
        total += n_pairs[k];
        first[(size_t)k + 1] = total;
    }
    if (out_first_pair < 0 || out_first_pair + total > ctx->capacity)
        return fail(ctx, ISS_E_INVALID, "output rows out of the reserved range");
    if (total == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const std::vector<int32_t> ids(genome_ids, genome_ids + n_items);
    bool single = true;
    for (int32_t k = 1; k < n_items; ++k) single &= ids[(size_t)k] == ids[0];
    std::vector<iss::BatchItem> call_items;
    iss::DevGenome dg{};
    bool any_exceptions = false;
    if (single) {
        // one record (a batch cut inside a long work item): its own buffers are the "arena", at offset 0
        const Genome &G = ctx->genomes[ids[0]];
        call_items.assign((size_t)n_items, iss::BatchItem{0, G.L, G.has_exceptions ? 1 : 0, 0});
        dg = iss::DevGenome{G.packed, G.mask, G.ascii, G.L, G.has_exceptions ? 1 : 0};
        any_exceptions = G.has_exceptions;
    } else {
        // ---- the records side by side in one arena (kept until another list of records is asked for; the buffers are
        // kept as long as they are large enough -- refilling them is ordered on the stream behind their last readers)
        if (ids != ctx->comm_ids) {
            std::vector<iss::BatchItem> items((size_t)n_items);
            std::vector<int64_t> place(ctx->genomes.size(), -1);  // a record used by several items stands once
            int64_t coord = 64;
            bool exceptions = false;
            for (int32_t k = 0; k < n_items; ++k) {
                const Genome &G = ctx->genomes[ids[k]];
                if (place[ids[k]] < 0) {
                    place[ids[k]] = coord;
                    coord += ((G.L + 31) / 32) * 32 + 64;  // zero padding between records (k_main's windows overhang by a few bases)
                }
                items[(size_t)k] = iss::BatchItem{place[ids[k]], G.L, G.has_exceptions ? 1 : 0, 0};
                exceptions |= G.has_exceptions;
            }
            // (arena coordinates are the pair descriptors' 36-bit coordinates and k_main's 32-bit word numbers, like a single
            //  record's: round 5 -- until then the records of a call had to stay below 2^31 bases)
            if (coord >= iss::MAX_RECORD) return fail(ctx, ISS_E_INVALID, "iss_generate_batch: the records of one call must stay below 2^34 - 4096 bases");
            if (coord > ctx->comm_cap) {
                { int rc_ = sync_all(ctx); if (rc_) return rc_; }
                free_community(ctx);
                const int64_t cap = coord + coord / 4;
                void *p = nullptr;
                HIP_TRY(ctx, hipMalloc(&p, ((size_t)cap / 16 + 8) * 4));
                ctx->comm_packed = static_cast<uint32_t *>(p);
                HIP_TRY(ctx, hipMalloc(&p, ((size_t)cap / 32 + 8) * 4));
                ctx->comm_mask = static_cast<uint32_t *>(p);
                HIP_TRY(ctx, hipMalloc(&p, (size_t)cap + 64));
                ctx->comm_ascii = static_cast<uint8_t *>(p);
                ctx->comm_cap = cap;
            }
            ctx->comm_ids.clear();  // (not valid while it is being refilled)
            HIP_TRY(ctx, hipMemsetAsync(ctx->comm_packed, 0, ((size_t)coord / 16 + 8) * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->comm_mask, 0, ((size_t)coord / 32 + 8) * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->comm_ascii, 'A', (size_t)coord + 64, ctx->stream));
            for (size_t g = 0; g < place.size(); ++g) {
                if (place[g] < 0) continue;
                const Genome &G = ctx->genomes[g];
                const size_t w_mk = (size_t)(G.L + 31) / 32;
                HIP_TRY(ctx, hipMemcpyAsync(ctx->comm_packed + 2 + place[g] / 16, G.packed, 2 * w_mk * 4, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->comm_mask + 2 + place[g] / 32, G.mask, w_mk * 4, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->comm_ascii + place[g], G.ascii, (size_t)G.L, hipMemcpyDeviceToDevice, ctx->stream));
            }
            ctx->comm_ids = ids;
            ctx->comm_items = items;
            ctx->comm_exceptions = exceptions;
            HIP_TRY(ctx, hipEventRecord(ctx->ev_inputs, ctx->stream));  // (k_setup may run on the setup stream: it waits for the arena)
            ctx->inputs_pending = true;
        }
        call_items = ctx->comm_items;
        dg = iss::DevGenome{ctx->comm_packed + 2, ctx->comm_mask + 2, ctx->comm_ascii, 0, ctx->comm_exceptions ? 1 : 0};
        any_exceptions = ctx->comm_exceptions;
    }
    if ((size_t)n_items + 1 > ctx->d_items_cap) {
        { int rc_ = sync_all(ctx); if (rc_) return rc_; }
        free_item_tables(ctx);
        const size_t cap = (size_t)n_items + 1 + 64;
        for (int k = 0; k < 2; ++k) {
            void *p = nullptr;
            HIP_TRY(ctx, hipMalloc(&p, cap * sizeof(iss::BatchItem)));
            ctx->d_items[k] = static_cast<iss::BatchItem *>(p);
            HIP_TRY(ctx, hipMalloc(&p, cap * sizeof(int64_t)));
            ctx->d_item_first[k] = static_cast<int64_t *>(p);
            HIP_TRY(ctx, hipHostMalloc(&p, cap * sizeof(iss::BatchItem), hipHostMallocDefault));
            ctx->h_items[k] = static_cast<iss::BatchItem *>(p);
            HIP_TRY(ctx, hipHostMalloc(&p, cap * sizeof(int64_t), hipHostMallocDefault));
            ctx->h_item_first[k] = static_cast<int64_t *>(p);
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_items[k], hipEventDisableTiming));
        }
        ctx->d_items_cap = cap;
        ctx->batch_seq = 0;
    }
    const int set = (int)(ctx->batch_seq & 1u);
    if (ctx->batch_seq >= 2) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_items[set]));  // the call before last is done with this set
    memcpy(ctx->h_items[set], call_items.data(), (size_t)n_items * sizeof(iss::BatchItem));
    memcpy(ctx->h_item_first[set], first.data(), ((size_t)n_items + 1) * sizeof(int64_t));
    // (on the stream k_setup runs on: beside the previous call's kernels, not behind them)
    hipStream_t s_in = setup_runs_ahead(ctx) ? ctx->setup_stream : ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_items[set], ctx->h_items[set], (size_t)n_items * sizeof(iss::BatchItem), hipMemcpyHostToDevice, s_in));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_item_first[set], ctx->h_item_first[set], ((size_t)n_items + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s_in));
    const int rc = generate_core(ctx, dg, any_exceptions, ctx->d_items[set], ctx->d_item_first[set], n_items, total, first_ordinal,
                                 seed, sequence_type, gc_bias, out_first_pair);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_items[set], ctx->stream));
    ++ctx->batch_seq;
    ctx->last_first.assign(first.begin(), first.end());
    ctx->last_off.resize((size_t)n_items);
    for (int32_t k = 0; k < n_items; ++k) ctx->last_off[(size_t)k] = call_items[(size_t)k].off;
    return 0;
}

// ---- the inner plugin surface (ErrorModel methods), batched: see iss_units.hip.h and include/iss_mi355x.h
namespace {
struct DevBuf {  // a device allocation freed at scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
int unit_prologue(iss_ctx *ctx, int32_t orientation, int64_t n, const char *what) {
    if (!ctx || !ctx->have_model) return fail(ctx, ISS_E_INVALID, std::string(what) + ": upload a model first");
    if (ctx->M.quality_mode != 0) return fail(ctx, ISS_E_INVALID, std::string(what) + ": KDErrorModel tables only");
    if ((orientation != 0 && orientation != 1) || n < 0 || n > (int64_t)0x7fffffff) return fail(ctx, ISS_E_INVALID, std::string(what) + ": bad argument");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, ISS_E_HIP, hipGetErrorString(e));
    return 0;
}
}  // namespace

int iss_gen_phred_scores(iss_ctx *ctx, int32_t orientation, int64_t n, uint64_t first_ordinal, uint64_t seed, uint8_t *quality) {
    if (int rc = unit_prologue(ctx, orientation, n, "iss_gen_phred_scores")) return rc;
    if (!n) return 0;
    if (!quality) return fail(ctx, ISS_E_INVALID, "iss_gen_phred_scores: NULL output");
    const size_t bytes = (size_t)n * ctx->M.RL;
    DevBuf d;
    HIP_TRY(ctx, hipMalloc(&d.p, bytes));
    const iss::UnitArgs U{seed, first_ordinal, orientation, (int32_t)n};
    hipLaunchKernelGGL(iss::k_unit_phred, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->M, U, static_cast<uint8_t *>(d.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(quality, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int iss_mut_sequence(iss_ctx *ctx, int32_t orientation, int64_t n, uint64_t first_ordinal, uint64_t seed, uint8_t *seq,
                     const uint8_t *quality, int32_t *status) {
    if (int rc = unit_prologue(ctx, orientation, n, "iss_mut_sequence")) return rc;
    if (!n) return 0;
    if (!seq || !quality || !status) return fail(ctx, ISS_E_INVALID, "iss_mut_sequence: NULL argument");
    const size_t bytes = (size_t)n * ctx->M.RL;
    for (size_t k = 0; k < bytes; ++k)
        if (quality[k] > (uint8_t)ctx->M.n_q) return fail(ctx, ISS_E_INVALID, "iss_mut_sequence: phred score outside the model's table");
    DevBuf ds, dq, dst;
    HIP_TRY(ctx, hipMalloc(&ds.p, bytes));
    HIP_TRY(ctx, hipMalloc(&dq.p, bytes));
    HIP_TRY(ctx, hipMalloc(&dst.p, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(ds.p, seq, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dq.p, quality, bytes, hipMemcpyHostToDevice, ctx->stream));
    const iss::UnitArgs U{seed, first_ordinal, orientation, (int32_t)n};
    hipLaunchKernelGGL(iss::k_unit_mut, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->M, U, static_cast<uint8_t *>(ds.p),
                       static_cast<const uint8_t *>(dq.p), static_cast<int32_t *>(dst.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(seq, ds.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(status, dst.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int iss_random_insert_size(iss_ctx *ctx, int64_t n, uint64_t first_ordinal, uint64_t seed, int64_t *insert_size) {
    if (int rc = unit_prologue(ctx, 0, n, "iss_random_insert_size")) return rc;
    if (!n) return 0;
    if (!insert_size) return fail(ctx, ISS_E_INVALID, "iss_random_insert_size: NULL output");
    DevBuf d;
    HIP_TRY(ctx, hipMalloc(&d.p, (size_t)n * sizeof(int64_t)));
    const iss::UnitArgs U{seed, first_ordinal, 0, (int32_t)n};
    hipLaunchKernelGGL(iss::k_unit_isize, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->M, U, static_cast<int64_t *>(d.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(insert_size, d.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int iss_ev_step(iss_ctx *ctx, int32_t orientation, int64_t n, const int32_t *cur, const uint64_t *m53, const uint64_t *v53,
                int32_t *next, int32_t *slot, uint8_t *mask) {
    if (int rc = unit_prologue(ctx, orientation, n, "iss_ev_step")) return rc;
    if (!n) return 0;
    if (!cur || !m53 || !v53 || !next || !slot || !mask) return fail(ctx, ISS_E_INVALID, "iss_ev_step: NULL argument");
    for (int64_t i = 0; i < n; ++i)
        if (cur[i] < -1 || cur[i] > ctx->M.ev_ns - 2 || (m53[i] >> 53) || (v53[i] >> 53))
            return fail(ctx, ISS_E_INVALID, "iss_ev_step: state or numerator out of range");
    DevBuf dc, dm, dv, dn, ds, dk;
    HIP_TRY(ctx, hipMalloc(&dc.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&dm.p, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc(&dv.p, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc(&dn.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&ds.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&dk.p, (size_t)n));
    HIP_TRY(ctx, hipMemcpyAsync(dc.p, cur, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dm.p, m53, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dv.p, v53, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(iss::k_unit_ev_step, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->M, orientation, (int32_t)n,
                       static_cast<const int32_t *>(dc.p), static_cast<const uint64_t *>(dm.p), static_cast<const uint64_t *>(dv.p),
                       static_cast<int32_t *>(dn.p), static_cast<int32_t *>(ds.p), static_cast<uint8_t *>(dk.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(next, dn.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(slot, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mask, dk.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int iss_introduce_indels(iss_ctx *ctx, int32_t orientation, int64_t n, uint64_t first_ordinal, uint64_t seed, const uint8_t *seq,
                         const int32_t *seq_len, const uint8_t *full_seq, int64_t full_len, const int64_t *bounds, uint8_t *out,
                         int32_t *status) {
    if (int rc = unit_prologue(ctx, orientation, n, "iss_introduce_indels")) return rc;
    if (!n) return 0;
    if (!seq || !seq_len || !full_seq || !bounds || !out || !status || full_len < 1)
        return fail(ctx, ISS_E_INVALID, "iss_introduce_indels: NULL argument");
    const int RL = ctx->M.RL;
    for (int64_t i = 0; i < n; ++i)
        if (seq_len[i] < 0 || seq_len[i] > RL) return fail(ctx, ISS_E_INVALID, "iss_introduce_indels: a read longer than read_length");
    for (int64_t i = 0; i < n; ++i)  // (read_start, read_end) index full_seq in adjust_seq_length; beyond its end is handled
        if (bounds[2 * i] < 0 || bounds[2 * i + 1] < 0)  // ('A' / IndexError as in the reference), a negative bound is not
            return fail(ctx, ISS_E_INVALID, "iss_introduce_indels: negative read bounds");
    const int32_t cap = 6 * RL + 8;  // letters (<= 5 RL + 8) + the event masks of the steps
    const size_t bytes = (size_t)n * RL;
    DevBuf ds, dl, dg, db, dw, dout, dst;
    HIP_TRY(ctx, hipMalloc(&ds.p, bytes));
    HIP_TRY(ctx, hipMalloc(&dl.p, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(&dg.p, (size_t)full_len));
    HIP_TRY(ctx, hipMalloc(&db.p, (size_t)n * 2 * sizeof(int64_t)));
    HIP_TRY(ctx, hipMalloc(&dw.p, (size_t)n * cap));
    HIP_TRY(ctx, hipMalloc(&dout.p, bytes));
    HIP_TRY(ctx, hipMalloc(&dst.p, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(ds.p, seq, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dl.p, seq_len, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dg.p, full_seq, (size_t)full_len, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(db.p, bounds, (size_t)n * 2 * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    const iss::UnitArgs U{seed, first_ordinal, orientation, (int32_t)n};
    hipLaunchKernelGGL(iss::k_unit_indels, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->M, U, static_cast<const uint8_t *>(ds.p),
                       static_cast<const int32_t *>(dl.p), static_cast<const uint8_t *>(dg.p), full_len, static_cast<const int64_t *>(db.p),
                       static_cast<uint8_t *>(dw.p), cap, static_cast<uint8_t *>(dout.p), static_cast<int32_t *>(dst.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(status, dst.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int iss_output_download_coords(iss_ctx *ctx, int64_t first_pair, int64_t n_pairs, int64_t *coords) {
    if (!ctx || !coords || first_pair < 0 || n_pairs < 0 || first_pair + n_pairs > ctx->capacity)
        return fail(ctx, ISS_E_INVALID, "iss_output_download_coords: rows out of range");
    std::vector<iss::PairDesc> tmp((size_t)n_pairs);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_pairs)
        HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ctx->desc + first_pair, sizeof(iss::PairDesc) * (size_t)n_pairs,
                                    hipMemcpyDeviceToHost, ctx->stream));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    for (int64_t i = 0; i < n_pairs; ++i) {
        int64_t off = 0;  // rows of a batch call carry arena coordinates: back to the record's own
        const int64_t r = first_pair + i - ctx->last_row0;
        if (!ctx->last_first.empty() && r >= 0 && r < ctx->last_n) {
            const size_t k = (size_t)(std::upper_bound(ctx->last_first.begin(), ctx->last_first.end(), r) - ctx->last_first.begin()) - 1;
            off = ctx->last_off[k];
        }
        coords[4 * i + 0] = iss::desc_fs(tmp[i]) - off;
        coords[4 * i + 1] = iss::desc_re(tmp[i]) - off - ctx->M.RL;
        coords[4 * i + 2] = iss::desc_re(tmp[i]) - off;
        coords[4 * i + 3] = tmp[i].isz;
    }
    return 0;
}

int iss_timing_enable(iss_ctx *ctx, int enable) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    int rc = settle_timing(ctx);
    ctx->timing = enable != 0;
    ctx->timing_main_only = enable == 2;
    ctx->timing_all = enable != 0 && enable != 2;  // (a split by kernel needs the kernels one after the other: every value but 2)
    return rc;
}

int iss_timing_read(iss_ctx *ctx, double ms[4], int64_t *n_launches) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    int rc = settle_timing(ctx);
    if (rc) return rc;
    for (int k = 0; k < 4; ++k) { if (ms) ms[k] = ctx->ms_acc[k]; ctx->ms_acc[k] = 0; }
    if (n_launches) *n_launches = ctx->n_launches;
    ctx->n_launches = 0;
    return 0;
}

int iss_stats_read(iss_ctx *ctx, int64_t *n_fixup_reads, int64_t *n_scripted_reads) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    uint64_t v[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpy(v, ctx->stats, sizeof v, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemset(ctx->stats, 0, sizeof v));
    if (n_fixup_reads) *n_fixup_reads = (int64_t)v[0];
    if (n_scripted_reads) *n_scripted_reads = (int64_t)v[1];
    return 0;
}

int iss_main_kernel(iss_ctx *ctx, char *name, int capacity) {
    if (!ctx || !name || capacity < 1) return fail(ctx, ISS_E_INVALID, "iss_main_kernel: ctx / name is NULL or capacity < 1");
    const size_t n = std::min(ctx->main_kernel.size(), (size_t)capacity - 1);
    memcpy(name, ctx->main_kernel.data(), n);
    name[n] = 0;
    return (int)n;
}

// ------------------------------------------------------------------ reference-compatible MT mode
int iss_mt_seed(iss_ctx *ctx, uint64_t seed) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    if (seed > 0xffffffffull) return fail(ctx, ISS_E_INVALID, "seed must be < 2^32 (numpy's legacy seeding raises)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    auto &m = ctx->mt;
    if (!m.d_state) {
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, 2 * sizeof(iss::MtState)));
        m.d_state = static_cast<iss::MtState *>(p);
        HIP_TRY(ctx, hipMalloc(&p, sizeof(iss::MtWalkResult)));
        m.d_res = static_cast<iss::MtWalkResult *>(p);
        HIP_TRY(ctx, hipMalloc(&p, sizeof(iss::MtGauss)));
        m.d_gauss = static_cast<iss::MtGauss *>(p);
    }
    HIP_TRY(ctx, hipMemset(m.d_gauss, 0, sizeof(iss::MtGauss)));  // np.random.seed() drops the cached gaussian
    iss::MtState st[2];
    const uint32_t key[1] = {(uint32_t)seed};
    mt_init_by_array(st[0].mt, key, 1);       // random.seed(seed)
    mt_init_genrand(st[1].mt, (uint32_t)seed);  // np.random.seed(seed)
    HIP_TRY(ctx, hipMemcpy(m.d_state, st, sizeof st, hipMemcpyHostToDevice));
    m.fill[0] = m.fill[1] = m.used[0] = m.used[1] = 0;
    m.seeded = true;
    return 0;
}

static int mt_reserve(iss_ctx *ctx, size_t cap_py, size_t cap_np) {
    auto &m = ctx->mt;
    const size_t want[2] = {cap_py, cap_np};
    for (int s = 0; s < 2; ++s) {
        if (m.cap[s] >= want[s]) continue;
        if (m.fill[s] != m.used[s]) {  // keep the unconsumed words
            std::vector<uint32_t> keep(m.fill[s] - m.used[s]);
            HIP_TRY(ctx, hipMemcpy(keep.data(), m.buf[s][m.cur[s]] + m.used[s], keep.size() * 4, hipMemcpyDeviceToHost));
            for (auto &b : m.buf[s]) { if (b) (void)hipFree(b); b = nullptr; }
            for (auto &b : m.buf[s]) { void *p = nullptr; HIP_TRY(ctx, hipMalloc(&p, want[s] * 4)); b = static_cast<uint32_t *>(p); }
            HIP_TRY(ctx, hipMemcpy(m.buf[s][0], keep.data(), keep.size() * 4, hipMemcpyHostToDevice));
            m.fill[s] = keep.size();
        } else {
            for (auto &b : m.buf[s]) { if (b) (void)hipFree(b); b = nullptr; }
            for (auto &b : m.buf[s]) { void *p = nullptr; HIP_TRY(ctx, hipMalloc(&p, want[s] * 4)); b = static_cast<uint32_t *>(p); }
            m.fill[s] = 0;
        }
        m.cur[s] = 0;
        m.used[s] = 0;
        m.cap[s] = want[s];
    }
    return 0;
}

int iss_generate_mt(iss_ctx *ctx, int32_t genome_id, int64_t n_pairs, int32_t sequence_type, int32_t gc_bias,
                    int64_t out_first_pair, int64_t *n_done) {
    if (n_done) *n_done = 0;
    if (!ctx || !ctx->have_model) return fail(ctx, ISS_E_INVALID, "iss_generate_mt: upload a model first");
    if (!ctx->mt.seeded) return fail(ctx, ISS_E_INVALID, "iss_generate_mt: call iss_mt_seed first");
    if (genome_id < 0 || genome_id >= (int32_t)ctx->genomes.size()) return fail(ctx, ISS_E_INVALID, "unknown genome id");
    if (sequence_type != ISS_SEQ_METAGENOMICS && sequence_type != ISS_SEQ_AMPLICON)
        return fail(ctx, ISS_E_INVALID, "sequence type is not supported");
    if (n_pairs < 0 || out_first_pair < 0 || out_first_pair + n_pairs > ctx->capacity)
        return fail(ctx, ISS_E_INVALID, "output rows out of the reserved range");
    const Genome &G = ctx->genomes[genome_id];
    const iss::DevModel &M = ctx->M;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    auto &m = ctx->mt;
    const int64_t CH = m.pool_ch ? m.pool_ch : 8192;  // (a worker of a set, lent for this call: its own turn length and buffers)
    const bool basic = M.quality_mode == 1;
    const size_t py_need = iss::mt_py_need(M.RL), np_need = iss::mt_np_need(M.RL, basic);
    if (!m.pool_ch) { int rc_ = mt_reserve(ctx, 3 * ((size_t)(CH + 1) * py_need + 1248), 3 * ((size_t)(CH + 1) * np_need + 1248)); if (rc_) return rc_; }
    if (!(M.RL < G.L)) {
        // the reference draws the insert size BEFORE its assertion fails (generator.py:121-126, 130)
        if (m.has_frag) {
            // np.random.normal(mu, sd) (generator.py:122): numpy's legacy polar Box-Muller -- a cached second value is used up,
            // else candidates of two doubles each are drawn until 0 < r2 < 1 and f * x1 is cached -- replayed on the host (libm)
            iss::MtGauss gs;
            HIP_TRY(ctx, hipMemcpy(&gs, m.d_gauss, sizeof gs, hipMemcpyDeviceToHost));
            if (gs.has_gauss) {
                gs.has_gauss = 0;
            } else {
                for (size_t used = 0;;) {
                    const size_t want[2] = {0, used + 256};
                    { int rc_ = mt_ensure(ctx, want); if (rc_) return rc_; }
                    uint32_t w[256];
                    // (on the context's stream, which mt_ensure has made wait for the refill: the streams are non-blocking, a copy
                    //  on the null stream would not be ordered behind the fill kernel and the leftover copy)
                    HIP_TRY(ctx, hipMemcpyAsync(w, m.buf[1][m.cur[1]] + m.used[1] + used, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    bool done = false;
                    for (int c = 0; c < 64 && !done; ++c) {
                        auto res53 = [](uint32_t a, uint32_t b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0); };
                        volatile double x1 = 2.0 * res53(w[4 * c], w[4 * c + 1]) - 1.0, x2 = 2.0 * res53(w[4 * c + 2], w[4 * c + 3]) - 1.0;
                        volatile double a2 = x1 * x1, b2 = x2 * x2;
                        volatile double r2 = a2 + b2;
                        used += 4;
                        if (r2 >= 1.0 || r2 == 0.0) continue;
                        volatile double f = -2.0 * log(r2);
                        f = f / r2;
                        f = sqrt(f);
                        gs.gauss = f * x1;
                        gs.has_gauss = 1;
                        gs.x1 = x1;
                        gs.x2 = x2;
                        done = true;
                    }
                    if (done) { m.used[1] += used; break; }
                }
            }
            HIP_TRY(ctx, hipMemcpyAsync(m.d_gauss, &gs, sizeof gs, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return fail(ctx, ISS_E_SHORT_RECORD, "record shorter than read length for this ErrorModel");
        }
        if (!basic) {  // (BasicErrorModel.random_insert_size is a constant: nothing is drawn)
            const size_t want[2] = {0, 2};
            { int rc_ = mt_ensure(ctx, want); if (rc_) return rc_; }
            m.used[1] += 2;
        }
        return fail(ctx, ISS_E_SHORT_RECORD, "record shorter than read length for this ErrorModel");
    }
    if (n_pairs == 0) return 0;
    const iss::DevGenome dg{G.packed, G.mask, G.ascii, G.L, G.has_exceptions ? 1 : 0};
    const size_t fixed_lds = iss::mt_walk_fixed_lds_bytes(M.RL);
    const size_t rows_bytes = (((size_t)2 * M.NB * M.RL * M.mt_row_w + 1) & ~(size_t)1) * 4;  // 16-bit digit rows
    const bool use_rows = !basic && rows_bytes + fixed_lds <= 150 * 1024;
    const size_t lds_bytes = fixed_lds + (use_rows ? rows_bytes : 0);
    // Resolver path (k_mt_resolve + k_mt_emit) for plain runs; the sequential walker for indel-heavy models, the
    // BasicErrorModel, and for the single pairs the resolver hands back.
    typedef void (*resolve_fn)(iss::DevModel, iss::DevGenome, iss::MtResolveArgs, iss::PairDesc *);
    resolve_fn resolve = nullptr;
    size_t resolve_lds = 0;
    {
        const char *force = getenv("ISS_MT_PATH");  // "walk": sequential walker only (testing aid)
        const bool allowed = !(force && !strcmp(force, "walk")) && ctx->mt_bounce_rate < 0.05 &&
                             M.n_isize <= 4096 && !basic;
        const size_t budget = 160 * 1024 - 256;
        const uint32_t need_py = iss::mt_res_need_py(M.RL), need_np = iss::mt_res_need_np(M.RL);
        struct Cand { int pyv, npv; bool rows; resolve_fn fn; };
        const Cand cands[8] = {  // digit rows in LDS first, then the smallest rings that show a whole pair
            {8, 2, true, iss::k_mt_resolve<8, 2, true>},   {4, 2, true, iss::k_mt_resolve<4, 2, true>},
            {8, 4, true, iss::k_mt_resolve<8, 4, true>},   {4, 4, true, iss::k_mt_resolve<4, 4, true>},
            {8, 2, false, iss::k_mt_resolve<8, 2, false>}, {4, 2, false, iss::k_mt_resolve<4, 2, false>},
            {8, 4, false, iss::k_mt_resolve<8, 4, false>}, {4, 4, false, iss::k_mt_resolve<4, 4, false>}};
        for (const Cand &c : cands) {
            if (!allowed || resolve) break;
            if (need_py > (uint32_t)c.pyv * 1024u || need_np > (uint32_t)c.npv * 1024u) continue;
            const size_t b = iss::mt_res_lds_bytes(M, c.pyv, c.npv, c.rows);
            if (b > budget) continue;
            resolve = c.fn;
            resolve_lds = b;
        }
        if (resolve) {
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)budget));
            if (!m.d_rec) {
                void *p = nullptr;
                HIP_TRY(ctx, hipMalloc(&p, (size_t)CH * sizeof(iss::MtPairRec)));
                m.d_rec = static_cast<iss::MtPairRec *>(p);
            }
            if (!m.d_mut_cnt) {  // (sized for the longest turn: a lent chain brings its own, shorter, d_rec)
                void *p = nullptr;
                HIP_TRY(ctx, hipMalloc(&p, (size_t)2 * 8192 * sizeof(int32_t)));
                m.d_mut_cnt = static_cast<int32_t *>(p);
                HIP_TRY(ctx, hipMalloc(&p, (size_t)2 * 8192 * sizeof(int64_t)));
                m.d_mut_off = static_cast<int64_t *>(p);
            }
        }
    }
    if (basic && !m.d_amb) {  // phreds the host has to round, and its answers
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, 2 * iss::MT_AMB_CAP * sizeof(iss::MtPhredAmb)));
        m.d_amb = static_cast<iss::MtPhredAmb *>(p);
    }
    std::vector<iss::MtPhredAmb> ovq;  // answers for the pair that restarts
    int64_t done = 0;
    m.mut_n = 0;
    bool ov_valid = false, walk_one = false;
    int64_t ov_frag = 0;
    // words wanted for a turn: those of n + 1 pairs, plus `boost` more when a turn made no progress on them -- with
    // gc_bias every rejected candidate pair (generator.py:82-92) consumes a whole pair's draws, and a turn of one pair that
    // meets three rejections in a row needs more than two pairs' worth
    int64_t boost = gc_bias ? 4 : 0;
    while (done < n_pairs) {
        const int64_t n = walk_one ? 1 : std::min(CH, n_pairs - done);
        const size_t want[2] = {std::min(m.cap[0] / 624 * 624 - 624, (size_t)(n + 1 + boost) * py_need),
                                std::min(m.cap[1] / 624 * 624 - 624, (size_t)(n + 1 + boost) * np_need)};
        { int rc_ = mt_ensure(ctx, want); if (rc_) return rc_; }
        MtPrefetch pf;
        if (!walk_one && done + n < n_pairs) {  // produce the next chunk's words while this chunk runs
            const int64_t n_next = std::min(CH, n_pairs - done - n);
            const size_t want_next[2] = {(size_t)(n_next + 1) * py_need, (size_t)(n_next + 1) * np_need};
            { int rc_ = mt_prefetch_begin(ctx, want, want_next, &pf); if (rc_) return rc_; }
        }
        const int64_t row0 = out_first_pair + done;
        iss::MtWalkResult res{};
        if (resolve && !walk_one) {
            iss::MtResolveArgs R{};
            R.py_base = m.buf[0][m.cur[0]];
            R.np_base = m.buf[1][m.cur[1]];
            R.py_off = (uint32_t)m.used[0];
            R.np_off = (uint32_t)m.used[1];
            R.py_fill = (uint32_t)m.fill[0];
            R.np_fill = (uint32_t)m.fill[1];
            R.py_cap = (uint32_t)m.cap[0];
            R.np_cap = (uint32_t)m.cap[1];
            R.n_pairs = n;
            R.sequence_type = sequence_type;
            R.gc_bias = gc_bias ? 1 : 0;
            R.gc_thr = 8106479329266893ull;
            R.res = m.d_res;
            R.rec = m.d_rec;
            R.has_frag = m.has_frag ? 1 : 0;
            R.frag_mu = m.frag_mu;
            R.frag_sd = m.frag_sd;
            R.guard = getenv("ISS_MT_GUARD") ? atof(getenv("ISS_MT_GUARD")) : 1e-6;
            R.gauss = m.d_gauss;
            hipLaunchKernelGGL(resolve, dim3(1), dim3(iss::RES_THREADS), resolve_lds, ctx->stream, M, dg, R, ctx->desc + row0);
            HIP_TRY(ctx, hipMemcpyAsync(&res, m.d_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipGetLastError());
            if (res.n_done > 0) {
                auto emit = [&](const iss::MtEmitMut &E) {
                    hipLaunchKernelGGL(iss::k_mt_emit, dim3((unsigned)((2 * res.n_done + 3) / 4)), dim3(256), 0, ctx->stream, M, dg,
                                       R.py_base, R.np_base, res.n_done, ctx->desc + row0, m.d_rec,
                                       ctx->out[0] + (size_t)row0 * M.row, ctx->out[1] + (size_t)row0 * M.row,
                                       ctx->out[2] + (size_t)row0 * M.row, ctx->out[3] + (size_t)row0 * M.row, E);
                };
                iss::MtEmitMut E{};
                if (!m.d_mut) {
                    emit(E);
                } else {
                    // --store_mutations: count the rows of every mate, place them with a prefix sum, write them in order
                    const size_t items = (size_t)(2 * res.n_done);
                    E.mut_cnt = m.d_mut_cnt;
                    emit(E);
                    std::vector<int32_t> cnt(items);
                    HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), m.d_mut_cnt, items * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    std::vector<int64_t> off(items);
                    int64_t at = m.mut_n;
                    for (size_t k = 0; k < items; ++k) { off[k] = at; at += cnt[k]; }
                    HIP_TRY(ctx, hipMemcpyAsync(m.d_mut_off, off.data(), items * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
                    E.mut_cnt = nullptr;
                    E.mut_off = m.d_mut_off;
                    E.mut = m.d_mut;
                    E.mut_cap = m.mut_cap;
                    E.pair_base = done;
                    emit(E);
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // `off` is pageable host memory
                    m.mut_n = at;
                }
            }
            m.used[0] += res.py_used;
            m.used[1] += res.np_used;
            { int rc_ = mt_prefetch_commit(ctx, pf); if (rc_) return rc_; }
            done += res.n_done;
            m.n_resolved += res.n_done;
            if (res.pad) { walk_one = true; continue; }  // the next pair is not plain: one turn of the walker
            if (res.n_done == 0 && res.starved && (size_t)(R.py_fill - R.py_off) >= want[0] &&
                (size_t)(R.np_fill - R.np_off) >= want[1]) {
                if (boost >= 256) return fail(ctx, ISS_E_INVALID, "MT stream buffers too small for one read pair");
                boost = 2 * boost + 4;
            }
            continue;
        }
        iss::MtWalkArgs A{};
        A.py = m.buf[0][m.cur[0]] + m.used[0];
        A.np = m.buf[1][m.cur[1]] + m.used[1];
        A.py_avail = (uint32_t)(m.fill[0] - m.used[0]);
        A.np_avail = (uint32_t)(m.fill[1] - m.used[1]);
        A.n_pairs = n;
        A.sequence_type = sequence_type;
        A.gc_bias = gc_bias ? 1 : 0;
        A.gc_thr = 8106479329266893ull;
        for (int k = 0; k < 4; ++k) A.out[k] = ctx->out[k] + (size_t)row0 * M.row;
        A.res = m.d_res;
        A.use_rows = use_rows && n > 64 ? 1 : 0;  // staging the rows (one wavefront, tens of KB) only pays for a real batch
        A.mut = m.d_mut;
        A.mut_cap = m.mut_cap;
        A.mut_base = m.mut_n;
        A.pair_base = done;
        A.has_frag = m.has_frag ? 1 : 0;
        A.frag_mu = m.frag_mu;
        A.frag_sd = m.frag_sd;
        A.ov_valid = ov_valid ? 1 : 0;
        A.ov_frag = ov_frag;
        A.guard = getenv("ISS_MT_GUARD") ? atof(getenv("ISS_MT_GUARD")) : 1e-6;
        if (basic && A.guard > 0.45) A.guard = 0.45;  // (a test aid: > 0.5 would make every phred "ambiguous" twice over)
        A.gauss = m.d_gauss;
        A.amb = m.d_amb;
        A.ovq = m.d_amb ? m.d_amb + iss::MT_AMB_CAP : nullptr;
        A.n_ovq = (int32_t)ovq.size();
        if (!ovq.empty())
            HIP_TRY(ctx, hipMemcpyAsync(m.d_amb + iss::MT_AMB_CAP, ovq.data(), ovq.size() * sizeof(iss::MtPhredAmb),
                                        hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(iss::k_mt_walk, dim3(1), dim3(64), A.use_rows ? lds_bytes : fixed_lds, ctx->stream, M, dg, A,
                           ctx->desc + row0);
        HIP_TRY(ctx, hipMemcpyAsync(&res, m.d_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipGetLastError());
        m.used[0] += res.py_used;
        m.used[1] += res.np_used;
        { int rc_ = mt_prefetch_commit(ctx, pf); if (rc_) return rc_; }
        done += res.n_done;
        m.n_walked += res.n_done;
        m.mut_n += res.n_mut;
        // host answers (phreds, fragment length) belong to the attempt that started the launch: they stay only if the
        // walk stopped again at that very attempt (gc_bias rejections move on to a new attempt of the same pair)
        const bool same_attempt = res.n_done == 0 && res.py_used == 0 && res.np_used == 0;
        if (!same_attempt) ovq.clear();
        if (res.need_host == 2) {
            // BasicErrorModel: phreds within the guard of a rounding boundary -- evaluated here exactly as numpy / the
            // reference do (libm): legacy_gauss f = sqrt(-2*log(r2)/r2); loc + scale*g; min(q, 0.9999);
            // int(round(-10 * log10(1 - p)))  (basic.py:52-53, util.py:44); the same pair restarts with the answers
            const int n_amb = std::min<int>(res.n_amb, iss::MT_AMB_CAP);
            std::vector<iss::MtPhredAmb> amb((size_t)n_amb);
            HIP_TRY(ctx, hipMemcpy(amb.data(), m.d_amb, amb.size() * sizeof(iss::MtPhredAmb), hipMemcpyDeviceToHost));
            for (auto &e : amb) {
                e.q = host_basic_phred(e.x1, e.x2, e.cached != 0, M.basic_mean, M.basic_sd, M.basic_cap);
                ovq.push_back(e);
            }
            if (ovq.size() > (size_t)iss::MT_AMB_CAP) return fail(ctx, ISS_E_INVALID, "too many undecidable phred scores in one pair");
            if (!same_attempt) ov_valid = false;  // (a restart of the SAME attempt keeps its host-evaluated fragment length)
            continue;
        }
        ov_valid = false;
        if (res.need_host) {
            // int(np.random.normal(mu, sd)) of the next pair with the host's libm, exactly as numpy's legacy_gauss:
            // f = sqrt(-2*log(r2)/r2); fresh value f*x2, cached value f*x1; loc + scale*g; int() truncates
            ov_frag = host_int_normal(res.host_x1, res.host_x2, res.host_cached != 0, m.frag_mu, m.frag_sd);
            ov_valid = true;
            continue;
        }
        if (res.n_done == 0 && res.starved && A.py_avail >= want[0] && A.np_avail >= want[1]) {
            if (boost >= 256) return fail(ctx, ISS_E_INVALID, "MT stream buffers too small for one read pair");
            boost = 2 * boost + 4;
        }
        if (res.n_done > 0) walk_one = false;
    }
    if (n_done) *n_done = done;
    return 0;
}

// ------------------------------------------------------------------ MT mode: W workers per launch (round 5)
// The reference's own parallelism is N workers, each a sequential chain over ITS two MT19937 streams seeded seed + cpu_number
// (iss/generator.py:234-236, iss/app.py:81-106).  One chain keeps one workgroup busy (k_mt_resolve: 2.3 us per NovaSeq pair);
// a set of W workers is W chains side by side: per turn ONE launch of each kernel of the path with one workgroup (k_mt_fill_w,
// k_mt_resolve_w, k_mt_walk_w) or one grid row (k_mt_emit_w) per worker, the jobs in tables in HBM.  Every worker's rows and
// stream positions are exactly those of iss_mt_seed(seed_w) + iss_generate_mt(...) in a context of its own.
int iss_mt_workers_seed(iss_ctx *ctx, int32_t n_workers, const uint64_t *seeds) {
    if (!ctx || n_workers < 1 || n_workers > 1024 || !seeds) return fail(ctx, ISS_E_INVALID, "iss_mt_workers_seed: 1 .. 1024 workers");
    for (int32_t w = 0; w < n_workers; ++w)
        if (seeds[w] > 0xffffffffull) return fail(ctx, ISS_E_INVALID, "seed must be < 2^32 (numpy's legacy seeding raises)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    free_mt_set(ctx);
    auto &t = ctx->mts;
    const size_t W = (size_t)n_workers;
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, 2 * W * sizeof(iss::MtState)));
    t.d_state = static_cast<iss::MtState *>(p);
    HIP_TRY(ctx, hipMalloc(&p, W * sizeof(iss::MtWalkResult)));
    t.d_res = static_cast<iss::MtWalkResult *>(p);
    HIP_TRY(ctx, hipMalloc(&p, W * sizeof(iss::MtGauss)));
    t.d_gauss = static_cast<iss::MtGauss *>(p);
    HIP_TRY(ctx, hipMemset(t.d_gauss, 0, W * sizeof(iss::MtGauss)));  // np.random.seed() drops the cached gaussian
    HIP_TRY(ctx, hipHostMalloc(&p, W * sizeof(iss::MtWalkResult), hipHostMallocDefault));
    t.h_res = static_cast<iss::MtWalkResult *>(p);
    std::vector<iss::MtState> st(2 * W);
    for (size_t w = 0; w < W; ++w) {
        const uint32_t key[1] = {(uint32_t)seeds[w]};
        mt_init_by_array(st[2 * w].mt, key, 1);               // random.seed(seed)
        mt_init_genrand(st[2 * w + 1].mt, (uint32_t)seeds[w]);  // np.random.seed(seed)
    }
    HIP_TRY(ctx, hipMemcpy(t.d_state, st.data(), st.size() * sizeof(iss::MtState), hipMemcpyHostToDevice));
    t.W = n_workers;
    t.started = t.poisoned = false;
    t.cur.assign(2 * W, 0);
    t.fill.assign(2 * W, 0);
    t.used.assign(2 * W, 0);
    t.last_read.assign(6 * W, -1);
    t.n_resolved = t.n_walked = 0;
    return 0;
}

namespace {

constexpr int MT_SET_BUFS = 2;  // stream buffers per (worker, stream) in rotation (see iss_ctx::MtSet::buf)

// stream buffers, pair records and job tables of the set, sized for the model (called by every generate call; a model with longer
// reads than the buffers were cut for is refused: seed the set again)
int mt_set_reserve(iss_ctx *ctx) {
    auto &t = ctx->mts;
    const iss::DevModel &M = ctx->M;
    const size_t W = (size_t)t.W;
    const bool basic = M.quality_mode == 1;
    const size_t need[2] = {iss::mt_py_need(M.RL), iss::mt_np_need(M.RL, basic)};
    if (!t.ch) {
        const char *e = getenv("ISS_MT_SET_TURN");  // pairs per worker and turn (tests: many turns)
        // (98 304 / W within 512 .. 4096: a worker whose resolver meets a pair for the walker loses the rest of its turn, a turn costs
        //  ~0.4 ms beside its resolver -- measured flat between 1024 and 1536 at W = 64, 512 and 768 at W = 256; 4096 against 8192
        //  at W = 8: + 7 %)
        t.ch = e ? std::max<int64_t>(1, std::min<int64_t>(8192, atoll(e))) : std::max<int64_t>(512, std::min<int64_t>(4096, 98304 / (int64_t)W));
    }
    // A buffer holds K turns' words (worst case): the words produced ahead are APPENDED behind a stream's valid words while there
    // is room, and only at a buffer's end the stream moves to the other buffer, its unconsumed words copied in front (round 5: with
    // K = 3 and a move every turn, the moves of the workers whose turn had ended early -- nearly a whole turn's words each, ~ 400 MB
    // per turn at W = 64 -- were 1 ms of a 6.5 ms turn, on the critical path).  K = 8 where 32 GB (and half of the free memory) hold it, 3 at least.
    const size_t turn_words[2] = {(size_t)(t.ch + 1) * need[0] + 1248, (size_t)(t.ch + 1) * need[1] + 1248};
    if (!t.buf_turns) {
        const char *e = getenv("ISS_MT_SET_BUF_TURNS");  // (tests: 3 = a move every second turn)
        const size_t per_k = W * MT_SET_BUFS * (turn_words[0] + turn_words[1]) * sizeof(uint32_t);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)64 << 30; }
        const size_t budget = std::min((size_t)32 << 30, free_b / 2);  // (half of what is free at most: other engines share the device)
        t.buf_turns = e ? std::max(3, std::min(8, atoi(e))) : (int)std::max<size_t>(3, std::min<size_t>(8, budget / per_k));
    }
    const size_t want[2] = {(size_t)t.buf_turns * turn_words[0], (size_t)t.buf_turns * turn_words[1]};
    if (t.cap[0] && (t.cap[0] < want[0] || t.cap[1] < want[1]))
        return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: the set's stream buffers were sized for a model with shorter reads (seed the set again)");
    if (!t.cap[0]) {
        // (all or nothing: a reservation that failed half way leaves nothing behind and can be repeated -- cap[] marks it as made)
        auto undo = [&]() {
            for (auto &st : t.buf) for (auto &b : st) { if (b) (void)hipFree(b); b = nullptr; }
            if (t.d_rec) (void)hipFree(t.d_rec);
            if (t.h_jobs) (void)hipHostFree(t.h_jobs);
            if (t.d_jobs) (void)hipFree(t.d_jobs);
            t.d_rec = nullptr; t.h_jobs = nullptr; t.d_jobs = nullptr;
            for (auto &e : t.ev_emit) { if (e) (void)hipEventDestroy(e); e = nullptr; }
            if (t.ev_side) (void)hipEventDestroy(t.ev_side);
            if (t.ev_turn) (void)hipEventDestroy(t.ev_turn);
            t.ev_side = t.ev_turn = nullptr;
            (void)hipGetLastError();
        };
        const size_t jobs_bytes = (((4 * 2 * W) * std::max(sizeof(iss::MtFillJob), sizeof(iss::MtMoveJob)) +
                                    W * (sizeof(iss::MtResolveJob) + sizeof(iss::MtWalkJob) + sizeof(iss::MtEmitJob))) + 255) & ~(size_t)255;
        bool ok = true;
        for (int s = 0; s < 2 && ok; ++s)
            for (int b = 0; b < MT_SET_BUFS && ok; ++b) {
                void *p = nullptr;
                ok = hipMalloc(&p, W * want[s] * sizeof(uint32_t)) == hipSuccess;
                t.buf[s][b] = ok ? static_cast<uint32_t *>(p) : nullptr;
            }
        void *p = nullptr;
        if (ok && (ok = hipMalloc(&p, 2 * W * (size_t)t.ch * sizeof(iss::MtPairRec)) == hipSuccess)) t.d_rec = static_cast<iss::MtPairRec *>(p);
        if (ok && (ok = hipHostMalloc(&p, 2 * jobs_bytes, hipHostMallocDefault) == hipSuccess)) t.h_jobs = static_cast<uint8_t *>(p);
        if (ok && (ok = hipMalloc(&p, 2 * jobs_bytes) == hipSuccess)) t.d_jobs = static_cast<uint8_t *>(p);
        if (!ok) {
            undo();
            return fail(ctx, ISS_E_NOMEM, "iss_generate_mt_workers: no memory for the workers' stream buffers (W x " + std::to_string((want[0] + want[1]) * MT_SET_BUFS * 4) + " bytes)");
        }
        hipError_t e = hipSuccess;
        for (auto &ev : t.ev_emit) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t.ev_side, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t.ev_turn, hipEventDisableTiming);
        if (e != hipSuccess) { undo(); HIP_TRY(ctx, e); }
        t.jobs_bytes = jobs_bytes;
        t.cap[0] = want[0];
        t.cap[1] = want[1];
    }
    return 0;
}

// One worker of the set through the single-worker path (iss_generate_mt): its chain is lent to ctx->mt for the call.  For what
// the side-by-side loop below does not do itself: records shorter than a read (the reference draws before its assertion
// fails), custom fragment lengths and the BasicErrorModel (draws the host's libm has to settle).
struct MtChainLoan {
    struct Chain {
        bool seeded; iss::MtState *d_state; uint32_t *buf[2][2]; int cur[2]; size_t cap[2], fill[2], used[2];
        iss::MtWalkResult *d_res; iss::MtGauss *d_gauss; iss::MtPairRec *d_rec; int64_t pool_ch;
    };
    iss_ctx *ctx;
    int w;
    Chain own;
    int64_t r0, w0;
    int base[2];
    typedef decltype(iss_ctx::mt) MtLegacy;
    static Chain save(const MtLegacy &m) {
        Chain c;
        c.seeded = m.seeded; c.d_state = m.d_state; c.d_res = m.d_res; c.d_gauss = m.d_gauss; c.d_rec = m.d_rec; c.pool_ch = m.pool_ch;
        for (int s = 0; s < 2; ++s) {
            c.cur[s] = m.cur[s]; c.cap[s] = m.cap[s]; c.fill[s] = m.fill[s]; c.used[s] = m.used[s];
            for (int b = 0; b < 2; ++b) c.buf[s][b] = m.buf[s][b];
        }
        return c;
    }
    static void load(MtLegacy &m, const Chain &c) {
        m.seeded = c.seeded; m.d_state = c.d_state; m.d_res = c.d_res; m.d_gauss = c.d_gauss; m.d_rec = c.d_rec; m.pool_ch = c.pool_ch;
        for (int s = 0; s < 2; ++s) {
            m.cur[s] = c.cur[s]; m.cap[s] = c.cap[s]; m.fill[s] = c.fill[s]; m.used[s] = c.used[s];
            for (int b = 0; b < 2; ++b) m.buf[s][b] = c.buf[s][b];
        }
    }
    MtChainLoan(iss_ctx *ctx_, int w_) : ctx(ctx_), w(w_) {
        auto &t = ctx->mts;
        auto &m = ctx->mt;
        own = save(m);
        r0 = m.n_resolved; w0 = m.n_walked;
        Chain c;
        c.seeded = true; c.d_state = t.d_state + 2 * (size_t)w; c.d_res = t.d_res + w; c.d_gauss = t.d_gauss + w;
        c.d_rec = t.d_rec + (size_t)w * (size_t)t.ch; c.pool_ch = t.ch;  // (the first of its two sets of pair records)
        for (int s = 0; s < 2; ++s) {  // (the single-worker path ping-pongs between the current buffer and the next of the rotation)
            base[s] = t.cur[2 * w + s];
            c.cur[s] = 0; c.cap[s] = t.cap[s]; c.fill[s] = t.fill[2 * w + s]; c.used[s] = t.used[2 * w + s];
            for (int b = 0; b < 2; ++b) c.buf[s][b] = t.buf[s][(base[s] + b) % MT_SET_BUFS] + (size_t)w * t.cap[s];
        }
        load(m, c);
    }
    ~MtChainLoan() {
        auto &t = ctx->mts;
        auto &m = ctx->mt;
        for (int s = 0; s < 2; ++s) { t.cur[2 * w + s] = (base[s] + m.cur[s]) % MT_SET_BUFS; t.fill[2 * w + s] = m.fill[s]; t.used[2 * w + s] = m.used[s]; }
        t.n_resolved += m.n_resolved - r0;
        t.n_walked += m.n_walked - w0;
        m.n_resolved = r0;
        m.n_walked = w0;
        load(m, own);
    }
};
int mt_set_single(iss_ctx *ctx, int w, int32_t genome_id, int64_t n_pairs, int32_t sequence_type, int32_t gc_bias, int64_t out_first_pair,
                  int64_t *n_done) {
    MtChainLoan loan(ctx, w);
    return iss_generate_mt(ctx, genome_id, n_pairs, sequence_type, gc_bias, out_first_pair, n_done);
}

}  // namespace

static int mt_workers_generate(iss_ctx *ctx, int32_t n_workers, const int32_t *genome_ids, const int64_t *n_pairs, const int64_t *out_first_pair,
                               int32_t sequence_type, int32_t gc_bias, int64_t *n_done, int32_t *status);

// A call that fails once it has begun (a HIP error, a draw the side-by-side path cannot take, stream buffers too small) returns in
// the middle of a turn: some workers' streams have advanced, rows are partly written, n_done / status say nothing for the others.
// The set is then POISONED -- every later call fails until iss_mt_workers_seed starts the workers anew -- instead of carrying on
// from undefined stream positions.  (A short record is not a failure: status[w] = ISS_E_SHORT_RECORD, the set goes on.)
int iss_generate_mt_workers(iss_ctx *ctx, int32_t n_workers, const int32_t *genome_ids, const int64_t *n_pairs, const int64_t *out_first_pair,
                            int32_t sequence_type, int32_t gc_bias, int64_t *n_done, int32_t *status) {
    if (ctx && ctx->mts.poisoned)
        return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: an earlier call failed half way (the workers' streams and rows are undefined): call iss_mt_workers_seed again");
    if (ctx) ctx->mts.started = false;
    const int rc = mt_workers_generate(ctx, n_workers, genome_ids, n_pairs, out_first_pair, sequence_type, gc_bias, n_done, status);
    if (rc < 0 && ctx && ctx->mts.started) ctx->mts.poisoned = true;
    return rc;
}

static int mt_workers_generate(iss_ctx *ctx, int32_t n_workers, const int32_t *genome_ids, const int64_t *n_pairs, const int64_t *out_first_pair,
                               int32_t sequence_type, int32_t gc_bias, int64_t *n_done, int32_t *status) {
    if (!ctx || !ctx->have_model) return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: upload a model first");
    auto &t = ctx->mts;
    if (n_workers < 1 || n_workers != t.W || !genome_ids || !n_pairs || !out_first_pair)
        return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: call iss_mt_workers_seed for this many workers first");
    if (sequence_type != ISS_SEQ_METAGENOMICS && sequence_type != ISS_SEQ_AMPLICON)
        return fail(ctx, ISS_E_INVALID, "sequence type is not supported");
    if (ctx->mt.d_mut) return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: --store_mutations rows are per context (one context per worker)");
    const int W = n_workers;
    const iss::DevModel &M = ctx->M;
    for (int w = 0; w < W; ++w) {
        if (n_done) n_done[w] = 0;
        if (status) status[w] = 0;
        if (n_pairs[w] == 0) continue;
        if (genome_ids[w] < 0 || genome_ids[w] >= (int32_t)ctx->genomes.size()) return fail(ctx, ISS_E_INVALID, "unknown genome id");
        if (n_pairs[w] < 0 || out_first_pair[w] < 0 || out_first_pair[w] + n_pairs[w] > ctx->capacity)
            return fail(ctx, ISS_E_INVALID, "output rows out of the reserved range");
        for (int v = 0; v < w; ++v)  // (the workers' rows must not overlap)
            if (n_pairs[v] > 0 && out_first_pair[w] < out_first_pair[v] + n_pairs[v] && out_first_pair[v] < out_first_pair[w] + n_pairs[w])
                return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: two workers' output rows overlap");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    { int rc_ = mt_set_reserve(ctx); if (rc_) return rc_; }
    t.started = true;  // (from here on a failure leaves the set undefined)
    auto &m = ctx->mt;
    const bool basic = M.quality_mode == 1;
    // the resolver (k_mt_resolve_w + k_mt_emit_w) for plain runs, the walker for indel-heavy models and for the single pairs the
    // resolver hands back -- the choice of iss_generate_mt
    typedef void (*resolve_fn)(iss::DevModel, const iss::MtResolveJob *);
    resolve_fn resolve = nullptr;
    size_t resolve_lds = 0;
    {
        const char *force = getenv("ISS_MT_PATH");  // "walk": sequential walker only (testing aid)
        const bool allowed = !(force && !strcmp(force, "walk")) && ctx->mt_bounce_rate < 0.05 && M.n_isize <= 4096 && !basic;
        const size_t budget = 160 * 1024 - 256;
        const uint32_t need_py = iss::mt_res_need_py(M.RL), need_np = iss::mt_res_need_np(M.RL);
        struct Cand { int pyv, npv; bool rows; resolve_fn fn; };
        const Cand cands[8] = {
            {8, 2, true, iss::k_mt_resolve_w<8, 2, true>},   {4, 2, true, iss::k_mt_resolve_w<4, 2, true>},
            {8, 4, true, iss::k_mt_resolve_w<8, 4, true>},   {4, 4, true, iss::k_mt_resolve_w<4, 4, true>},
            {8, 2, false, iss::k_mt_resolve_w<8, 2, false>}, {4, 2, false, iss::k_mt_resolve_w<4, 2, false>},
            {8, 4, false, iss::k_mt_resolve_w<8, 4, false>}, {4, 4, false, iss::k_mt_resolve_w<4, 4, false>}};
        for (const Cand &c : cands) {
            if (!allowed || resolve) break;
            if (need_py > (uint32_t)c.pyv * 1024u || need_np > (uint32_t)c.npv * 1024u) continue;
            const size_t b = iss::mt_res_lds_bytes(M, c.pyv, c.npv, c.rows);
            if (b > budget) continue;
            resolve = c.fn;
            resolve_lds = b;
        }
        if (resolve) HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(resolve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(iss::k_mt_walk_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
    }
    // ---- what the side-by-side loop does not do itself goes through the single-worker path, one worker after the other
    struct WS { int64_t n = 0, done = 0, row0 = 0; int32_t gid = 0; bool walk_one = false; int64_t boost = 0; };
    std::vector<WS> ws((size_t)W);
    const bool one_by_one = m.has_frag || basic;
    for (int w = 0; w < W; ++w) {
        if (n_pairs[w] == 0) continue;
        const Genome &G = ctx->genomes[genome_ids[w]];
        if (one_by_one || !(M.RL < G.L)) {
            int64_t dn = 0;
            const int rc = mt_set_single(ctx, w, genome_ids[w], n_pairs[w], sequence_type, gc_bias, out_first_pair[w], &dn);
            if (n_done) n_done[w] = dn;
            if (rc == ISS_E_SHORT_RECORD) { if (status) status[w] = rc; continue; }
            if (rc) return rc;
            continue;
        }
        ws[w].n = n_pairs[w];
        ws[w].row0 = out_first_pair[w];
        ws[w].gid = genome_ids[w];
        ws[w].boost = gc_bias ? 4 : 0;
    }
    const size_t need[2] = {iss::mt_py_need(M.RL), iss::mt_np_need(M.RL, basic)};
    // Words a turn is given: `need` is the most ONE attempt at a pair can consume (the kernels stop in front of a pair they
    // might not finish: "starved"), but a turn of n pairs consumes n times the USUAL amount -- a plain pair takes 2 x (10 (RL - 1)
    // + 2 RL) + ~2 words of `random` and 2 + 2 x (2 + 2 RL + 2 per substitution) (+ 2) of numpy, a gc_bias rejection a whole
    // pair's more (10 %) -- and what is left over is moved in front of the next turn's words: sized for the usual amount (+ 3 %,
    // + a few whole attempts), a turn leaves a few per cent of its words instead of half of numpy's.  A turn that runs out early
    // ends early, with its pairs done; the next one carries on.
    const double gcf = gc_bias ? 1.15 : 1.0;
    const double est[2] = {gcf * 1.03 * (2.0 * (10.0 * (M.RL - 1) + 2.0 * M.RL) + 4.0),
                           gcf * 1.03 * (2.0 + 2.0 * (2.0 + 2.0 * M.RL) + 8.0 * (0.02 * 2.0 * M.RL) + 4.0)};  // (2 words per substitution pick and mate... 2 % of the bases substituted: generous for every shipped model)
    auto words_for = [&](int s, int64_t n, int64_t boost) {
        return std::min((size_t)(n + 1 + boost) * need[s], (size_t)((double)n * est[s]) + (size_t)(4 + boost) * need[s]);
    };
    const size_t fixed_lds = iss::mt_walk_fixed_lds_bytes(M.RL);
    const size_t rows_bytes = (((size_t)2 * M.NB * M.RL * M.mt_row_w + 1) & ~(size_t)1) * 4;
    const bool use_rows = !basic && rows_bytes + fixed_lds <= 150 * 1024;
    const double guard = getenv("ISS_MT_GUARD") ? atof(getenv("ISS_MT_GUARD")) : 1e-6;
    if (!m.ev_main) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&m.ev_main, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&m.ev_fill, hipEventDisableTiming));
    }
    auto bufp = [&](int w, int s, int b) { return t.buf[s][b] + (size_t)w * t.cap[s]; };
    const size_t fm_sz = std::max(sizeof(iss::MtFillJob), sizeof(iss::MtMoveJob));
    std::vector<int64_t> n_w((size_t)W);
    std::vector<size_t> want(2 * (size_t)W);
    struct PF { bool on = false, append = false; size_t at = 0; uint32_t blocks = 0; };
    std::vector<PF> pf(2 * (size_t)W);
    std::vector<int> res_buf(2 * (size_t)W);
    const bool dbg = getenv("ISS_MT_SET_DEBUG") != nullptr;  // per call: turns, words produced / moved, pairs handed to the walker
    uint64_t dbg_turns = 0, dbg_moved[2] = {0, 0}, dbg_filled[2] = {0, 0}, dbg_bounce = 0, dbg_moves = 0, dbg_big = 0, dbg_pairs = 0, dbg_appends = 0, dbg_starved = 0, dbg_own = 0, dbg_skip = 0, dbg_ensure = 0;
    std::fill(t.last_read.begin(), t.last_read.end(), (int64_t)-1);  // (everything before this call has been waited for: sync_all above)
    for (;;) {
        bool any = false;
        for (int w = 0; w < W; ++w) {
            n_w[w] = ws[w].done < ws[w].n ? (ws[w].walk_one ? 1 : std::min(t.ch, ws[w].n - ws[w].done)) : 0;
            any |= n_w[w] > 0;
        }
        if (!any) break;
        const int64_t turn = ++t.turns;  // (>= 1)
        const int par = (int)(turn & 1);
        // a buffer about to be WRITTEN (produced into, moved into) may still be read by an emitter: the one of two turns ago has
        // been waited for at the top of the turn, the one of the turn before only if a target says so
        auto read_by_last_turn = [&](int k, int b) { return t.last_read[(size_t)k * 3 + b] == turn - 1; };
        uint8_t *hj = t.h_jobs + (size_t)par * t.jobs_bytes, *dj = t.d_jobs + (size_t)par * t.jobs_bytes;
        auto tab = [&](size_t k, uint8_t *base) { return base + k * 2 * (size_t)W * fm_sz; };  // tables 0..3 (fill / move), then the rest
        iss::MtFillJob *h_fill_e = reinterpret_cast<iss::MtFillJob *>(tab(0, hj)), *h_fill_a = reinterpret_cast<iss::MtFillJob *>(tab(1, hj));
        iss::MtMoveJob *h_move_e = reinterpret_cast<iss::MtMoveJob *>(tab(2, hj)), *h_move_c = reinterpret_cast<iss::MtMoveJob *>(tab(3, hj));
        uint8_t *h_rest = tab(4, hj), *d_rest = tab(4, dj);
        iss::MtResolveJob *h_rj = reinterpret_cast<iss::MtResolveJob *>(h_rest);
        iss::MtWalkJob *h_wj = reinterpret_cast<iss::MtWalkJob *>(h_rest + (size_t)W * sizeof(iss::MtResolveJob));
        iss::MtEmitJob *h_ej = reinterpret_cast<iss::MtEmitJob *>(h_rest + (size_t)W * (sizeof(iss::MtResolveJob) + sizeof(iss::MtWalkJob)));
        auto dev_of = [&](const void *h) { return dj + (reinterpret_cast<const uint8_t *>(h) - hj); };
        hipStream_t s_side = ctx->setup_stream;  // the walker beside the resolver, the emitter beside the NEXT turn's resolver
        hipStream_t s_emit = ctx->emit_stream;
        // ---- (a) every worker of the turn has the words of n + 1 pairs (+ boost) in front of it: mt_ensure, for all at once
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, m.ev_fill, 0));  // (words produced ahead during the turn before)
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, t.ev_emit[par], 0));  // (this parity's pair records: their last reader, two turns ago)
        bool fill_e = false, move_e = false, wait_prev_e = false;
        for (int w = 0; w < W; ++w)
            for (int s = 0; s < 2; ++s) {
                const int k = 2 * w + s;
                h_fill_e[k] = iss::MtFillJob{nullptr, nullptr, 0u, 0u};
                h_move_e[k] = iss::MtMoveJob{nullptr, nullptr, 0u, 0u};
                want[k] = n_w[w] ? std::min(t.cap[s] / 624 * 624 - 624, words_for(s, n_w[w], ws[w].boost)) : 0;
                const size_t left = t.fill[k] - t.used[k];
                if (left >= want[k]) continue;
                const size_t missing = (want[k] - left + 623) / 624;
                ++dbg_ensure;
                if (t.fill[k] + missing * 624 <= t.cap[s]) {  // appended in place: nothing moves, nobody reads behind `fill`
                    h_fill_e[k] = iss::MtFillJob{t.d_state + k, bufp(w, s, t.cur[k]) + t.fill[k], (uint32_t)missing, 0u};
                    t.fill[k] += missing * 624;
                    fill_e = true;
                    continue;
                }
                move_e = true;
                const int nxt = (t.cur[k] + 1) % MT_SET_BUFS;
                h_move_e[k] = iss::MtMoveJob{bufp(w, s, t.cur[k]) + t.used[k], bufp(w, s, nxt), (uint32_t)left, 0u};
                const size_t room = (t.cap[s] - left) / 624;
                const uint32_t blocks = (uint32_t)std::min(room, (want[k] - left + 623) / 624);
                h_fill_e[k] = iss::MtFillJob{t.d_state + k, bufp(w, s, nxt) + left, blocks, 0u};
                wait_prev_e |= read_by_last_turn(k, nxt);
                t.cur[k] = nxt;
                t.used[k] = 0;
                t.fill[k] = left + (size_t)blocks * 624;
                fill_e = true;
            }
        if (fill_e) {
            if (wait_prev_e) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, t.ev_emit[par ^ 1], 0));  // (the emitter of the turn before reads a buffer written now)
            if (move_e) {
                HIP_TRY(ctx, hipMemcpyAsync(dev_of(h_move_e), h_move_e, 2 * (size_t)W * sizeof(iss::MtMoveJob), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(iss::k_mt_move_w, dim3(2 * W, iss::MOVE_BLOCKS), dim3(256), 0, ctx->stream, reinterpret_cast<const iss::MtMoveJob *>(dev_of(h_move_e)));
            }
            HIP_TRY(ctx, hipEventRecord(m.ev_main, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->fill_stream, m.ev_main, 0));  // (incl. the main stream's wait for that emitter)
            HIP_TRY(ctx, hipMemcpyAsync(dev_of(h_fill_e), h_fill_e, 2 * (size_t)W * sizeof(iss::MtFillJob), hipMemcpyHostToDevice, ctx->fill_stream));
            hipLaunchKernelGGL(iss::k_mt_fill_w, dim3(2 * W), dim3(iss::FILL_THREADS), 0, ctx->fill_stream, reinterpret_cast<const iss::MtFillJob *>(dev_of(h_fill_e)));
            HIP_TRY(ctx, hipEventRecord(m.ev_fill, ctx->fill_stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, m.ev_fill, 0));
        }
        // ---- (b) the words of the turn AFTER this one are produced beside it (mt_prefetch_begin)
        bool fill_a = false;
        for (int w = 0; w < W; ++w)
            for (int s = 0; s < 2; ++s) {
                const int k = 2 * w + s;
                pf[k] = PF{};
                h_fill_a[k] = iss::MtFillJob{nullptr, nullptr, 0u, 0u};
                if (!n_w[w] || ws[w].walk_one || ws[w].done + n_w[w] >= ws[w].n) continue;
                const int64_t n_next = std::min(t.ch, ws[w].n - ws[w].done - n_w[w]);
                const size_t want_next = words_for(s, n_next, 0);
                const size_t avail = t.fill[k] - t.used[k];
                if (avail >= want[k] + want_next) continue;
                // (only what is missing: everything in front of the turn is moved behind it -- a backlog would be copied every turn)
                const size_t blocks = (want[k] + want_next - avail + 623) / 624;
                if (t.fill[k] + blocks * 624 <= t.cap[s]) {  // appended in place (committed in (f) by moving `fill` on: no copy)
                    pf[k].on = true; pf[k].append = true; pf[k].blocks = (uint32_t)blocks;
                    h_fill_a[k] = iss::MtFillJob{t.d_state + k, bufp(w, s, t.cur[k]) + t.fill[k], (uint32_t)blocks, 0u};
                    fill_a = true;
                    continue;
                }
                if (avail + blocks * 624 > t.cap[s]) continue;
                pf[k].on = true; pf[k].at = avail; pf[k].blocks = (uint32_t)blocks;
                h_fill_a[k] = iss::MtFillJob{t.d_state + k, bufp(w, s, (t.cur[k] + 1) % MT_SET_BUFS) + avail, (uint32_t)blocks, 0u};
                fill_a = true;
            }
        if (fill_a) {  // (behind everything queued on the main stream so far -- incl. its wait for the emitter of two turns ago -- and
                       //  ALWAYS behind the emitter of the turn before, whether that one still reads a target (the other buffer of a
                       //  stream at its buffer's end) or not (words appended): fill, emitter and resolver all three together is what
                       //  the resolver -- the chain the turn waits for -- loses by: 1.53 against 1.79e7 pairs/s at W = 64, 3.3
                       //  against 4.7e7 at W = 256 with the wait left out for appended words)
            HIP_TRY(ctx, hipEventRecord(m.ev_main, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->fill_stream, m.ev_main, 0));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->fill_stream, t.ev_emit[par ^ 1], 0));
            HIP_TRY(ctx, hipMemcpyAsync(dev_of(h_fill_a), h_fill_a, 2 * (size_t)W * sizeof(iss::MtFillJob), hipMemcpyHostToDevice, ctx->fill_stream));
            hipLaunchKernelGGL(iss::k_mt_fill_w, dim3(2 * W), dim3(iss::FILL_THREADS), 0, ctx->fill_stream, reinterpret_cast<const iss::MtFillJob *>(dev_of(h_fill_a)));
            HIP_TRY(ctx, hipEventRecord(m.ev_fill, ctx->fill_stream));
        }
        // ---- (c) the turn: the resolver for the workers on the fast path, the walker for the others
        bool any_res = false, any_walk = false, walk_rows = false;
        for (int w = 0; w < W; ++w) {
            const Genome &G = ctx->genomes[ws[w].gid];
            const iss::DevGenome dg{G.packed, G.mask, G.ascii, G.L, G.has_exceptions ? 1 : 0};
            const int64_t row0 = ws[w].row0 + ws[w].done;
            const bool walker = n_w[w] > 0 && (!resolve || ws[w].walk_one);
            iss::MtResolveJob &rj = h_rj[w];
            rj = iss::MtResolveJob{};
            iss::MtWalkJob &wj = h_wj[w];
            wj = iss::MtWalkJob{};
            if (n_w[w] > 0 && !walker) {
                iss::MtResolveArgs &R = rj.A;
                R.py_base = bufp(w, 0, t.cur[2 * w]);
                R.np_base = bufp(w, 1, t.cur[2 * w + 1]);
                res_buf[2 * w] = t.cur[2 * w];
                res_buf[2 * w + 1] = t.cur[2 * w + 1];
                R.py_off = (uint32_t)t.used[2 * w];
                R.np_off = (uint32_t)t.used[2 * w + 1];
                R.py_fill = (uint32_t)t.fill[2 * w];
                R.np_fill = (uint32_t)t.fill[2 * w + 1];
                R.py_cap = (uint32_t)t.cap[0];
                R.np_cap = (uint32_t)t.cap[1];
                R.n_pairs = n_w[w];
                R.sequence_type = sequence_type;
                R.gc_bias = gc_bias ? 1 : 0;
                R.gc_thr = 8106479329266893ull;
                R.res = t.d_res + w;
                R.rec = t.d_rec + ((size_t)par * (size_t)W + (size_t)w) * (size_t)t.ch;
                R.has_frag = 0;
                R.guard = guard;
                R.gauss = t.d_gauss + w;
                rj.g = dg;
                rj.desc = ctx->desc + row0;
                any_res = true;
            } else if (walker) {
                iss::MtWalkArgs &A = wj.A;
                A.py = bufp(w, 0, t.cur[2 * w]) + t.used[2 * w];
                A.np = bufp(w, 1, t.cur[2 * w + 1]) + t.used[2 * w + 1];
                A.py_avail = (uint32_t)(t.fill[2 * w] - t.used[2 * w]);
                A.np_avail = (uint32_t)(t.fill[2 * w + 1] - t.used[2 * w + 1]);
                A.n_pairs = n_w[w];
                A.sequence_type = sequence_type;
                A.gc_bias = gc_bias ? 1 : 0;
                A.gc_thr = 8106479329266893ull;
                for (int k = 0; k < 4; ++k) A.out[k] = ctx->out[k] + (size_t)row0 * M.row;
                A.res = t.d_res + w;
                A.use_rows = use_rows && n_w[w] > 64 ? 1 : 0;
                A.pair_base = ws[w].done;
                A.guard = guard;
                A.gauss = t.d_gauss + w;
                wj.g = dg;
                wj.desc = ctx->desc + row0;
                any_walk = true;
                walk_rows |= A.use_rows != 0;
            }
        }
        HIP_TRY(ctx, hipMemcpyAsync(d_rest, h_rest, (size_t)W * (sizeof(iss::MtResolveJob) + sizeof(iss::MtWalkJob)), hipMemcpyHostToDevice, ctx->stream));
        if (any_walk) {  // (beside the resolver: other workers; behind the words and tables the main stream has waited for / copied)
            HIP_TRY(ctx, hipEventRecord(t.ev_turn, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(s_side, t.ev_turn, 0));
            hipLaunchKernelGGL(iss::k_mt_walk_w, dim3(W), dim3(64), walk_rows ? fixed_lds + rows_bytes : fixed_lds, s_side, M, reinterpret_cast<const iss::MtWalkJob *>(dev_of(h_wj)));
            HIP_TRY(ctx, hipEventRecord(t.ev_side, s_side));
        }
        if (any_res) hipLaunchKernelGGL(resolve, dim3(W), dim3(iss::RES_THREADS), resolve_lds, ctx->stream, M, reinterpret_cast<const iss::MtResolveJob *>(dev_of(h_rj)));
        if (any_walk) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, t.ev_side, 0));
        HIP_TRY(ctx, hipMemcpyAsync(t.h_res, t.d_res, (size_t)W * sizeof(iss::MtWalkResult), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipGetLastError());
        // ---- (d) the reads of the resolved pairs, all workers in one launch
        int64_t emit_max = 0;
        for (int w = 0; w < W; ++w) {
            iss::MtEmitJob &ej = h_ej[w];
            ej = iss::MtEmitJob{};
            if (!(n_w[w] > 0 && h_rj[w].A.n_pairs > 0) || t.h_res[w].n_done <= 0) continue;
            const int64_t row0 = ws[w].row0 + ws[w].done;
            ej.py = h_rj[w].A.py_base;
            ej.np = h_rj[w].A.np_base;
            ej.n_pairs = t.h_res[w].n_done;
            ej.desc = ctx->desc + row0;
            ej.rec = h_rj[w].A.rec;
            for (int k = 0; k < 4; ++k) ej.out[k] = ctx->out[k] + (size_t)row0 * M.row;
            ej.g = h_rj[w].g;
            emit_max = std::max(emit_max, ej.n_pairs);
            for (int s = 0; s < 2; ++s) t.last_read[(size_t)(2 * w + s) * 3 + res_buf[2 * w + s]] = turn;
        }
        if (emit_max > 0) {  // (on its own stream: the next turn's resolver does not wait for it.  Launched here, in front of the walk
                             //  behind the turn, not after it: the walk would run on a quiet chip -- 0.19 ms beside the emitter --
                             //  but the emitter would reach 0.1 ms further into the next resolver: 1.89 against 1.92e7 pairs/s at
                             //  W = 64, 4.96 against 5.12e7 at W = 256)
            HIP_TRY(ctx, hipMemcpyAsync(dev_of(h_ej), h_ej, (size_t)W * sizeof(iss::MtEmitJob), hipMemcpyHostToDevice, s_emit));
            hipLaunchKernelGGL(iss::k_mt_emit_w, dim3((unsigned)((2 * emit_max + 3) / 4), (unsigned)W), dim3(256), 0, s_emit, M,
                               reinterpret_cast<const iss::MtEmitJob *>(dev_of(h_ej)));
        }
        HIP_TRY(ctx, hipEventRecord(t.ev_emit[par], s_emit));
        // ---- (e) what the turn consumed and produced; a resolver that stopped in front of a pair for the walker (an indel candidate,
        //      a letter outside ACGT, a genome end in a template) gets that ONE pair walked right here, behind the turn, so that
        //      its worker is back on the fast path with the next turn (a turn of its own for one pair cost a worker 1.5 turns
        //      per such pair: 21 turns instead of 16 for a call of 16 full ones at W = 64)
        bool any_odd = false;
        for (int w = 0; w < W; ++w) {
            if (!n_w[w]) continue;
            const iss::MtWalkResult &res = t.h_res[w];
            const bool walker = h_wj[w].A.n_pairs > 0;
            if (res.need_host) return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: a draw for the host's libm on the side-by-side path");
            t.used[2 * w] += res.py_used;
            t.used[2 * w + 1] += res.np_used;
            ws[w].done += res.n_done;
            if (dbg) { dbg_starved += res.starved != 0; dbg_own += ws[w].walk_one; }
            if (walker) {
                t.n_walked += res.n_done;
                if (res.n_done == 0 && res.starved && h_wj[w].A.py_avail >= want[2 * w] && h_wj[w].A.np_avail >= want[2 * w + 1]) {
                    if (ws[w].boost >= 256) return fail(ctx, ISS_E_INVALID, "MT stream buffers too small for one read pair");
                    ws[w].boost = 2 * ws[w].boost + 4;
                }
                if (res.n_done > 0) ws[w].walk_one = false;
            } else {
                t.n_resolved += res.n_done;
                if (res.pad) {
                    ++dbg_bounce;
                    ws[w].walk_one = true;  // (unless the walk behind this turn takes it)
                    any_odd = true;
                } else if (res.n_done == 0 && res.starved && (size_t)(h_rj[w].A.py_fill - h_rj[w].A.py_off) >= want[2 * w] &&
                           (size_t)(h_rj[w].A.np_fill - h_rj[w].A.np_off) >= want[2 * w + 1]) {
                    if (ws[w].boost >= 256) return fail(ctx, ISS_E_INVALID, "MT stream buffers too small for one read pair");
                    ws[w].boost = 2 * ws[w].boost + 4;
                }
            }
        }
        if (any_odd) {
            int n_odd = 0;
            for (int w = 0; w < W; ++w) {
                const bool odd = n_w[w] > 0 && h_rj[w].A.n_pairs > 0 && t.h_res[w].pad && ws[w].done < ws[w].n;
                iss::MtWalkJob &wj = h_wj[w];
                wj = iss::MtWalkJob{};
                // (the words of one attempt at a pair must stand in front of the walker: else the pair waits for its own turn)
                if (dbg && odd && (t.fill[2 * w] - t.used[2 * w] < 2 * need[0] || t.fill[2 * w + 1] - t.used[2 * w + 1] < 2 * need[1])) ++dbg_skip;
                if (!odd || t.fill[2 * w] - t.used[2 * w] < 2 * need[0] || t.fill[2 * w + 1] - t.used[2 * w + 1] < 2 * need[1]) continue;
                const Genome &G = ctx->genomes[ws[w].gid];
                const int64_t row0 = ws[w].row0 + ws[w].done;
                iss::MtWalkArgs &A = wj.A;
                A.py = bufp(w, 0, t.cur[2 * w]) + t.used[2 * w];
                A.np = bufp(w, 1, t.cur[2 * w + 1]) + t.used[2 * w + 1];
                A.py_avail = (uint32_t)(t.fill[2 * w] - t.used[2 * w]);
                A.np_avail = (uint32_t)(t.fill[2 * w + 1] - t.used[2 * w + 1]);
                A.n_pairs = 1;
                A.sequence_type = sequence_type;
                A.gc_bias = gc_bias ? 1 : 0;
                A.gc_thr = 8106479329266893ull;
                for (int k = 0; k < 4; ++k) A.out[k] = ctx->out[k] + (size_t)row0 * M.row;
                A.res = t.d_res + w;
                A.pair_base = ws[w].done;
                A.guard = guard;
                A.gauss = t.d_gauss + w;
                wj.g = iss::DevGenome{G.packed, G.mask, G.ascii, G.L, G.has_exceptions ? 1 : 0};
                wj.desc = ctx->desc + row0;
                ++n_odd;
            }
            if (n_odd) {
                HIP_TRY(ctx, hipMemcpyAsync(dev_of(h_wj), h_wj, (size_t)W * sizeof(iss::MtWalkJob), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(iss::k_mt_walk_w, dim3(W), dim3(64), fixed_lds, ctx->stream, M, reinterpret_cast<const iss::MtWalkJob *>(dev_of(h_wj)));
                HIP_TRY(ctx, hipMemcpyAsync(t.h_res, t.d_res, (size_t)W * sizeof(iss::MtWalkResult), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                HIP_TRY(ctx, hipGetLastError());
                for (int w = 0; w < W; ++w) {
                    if (h_wj[w].A.n_pairs <= 0) continue;
                    const iss::MtWalkResult &res = t.h_res[w];
                    if (res.need_host) return fail(ctx, ISS_E_INVALID, "iss_generate_mt_workers: a draw for the host's libm on the side-by-side path");
                    t.used[2 * w] += res.py_used;
                    t.used[2 * w + 1] += res.np_used;
                    ws[w].done += res.n_done;
                    t.n_walked += res.n_done;
                    if (res.n_done > 0) ws[w].walk_one = false;  // (a gc_bias rejection, or starved: the pair takes a turn of its own)
                }
            }
        }
        // ---- (f) the streams move on (mt_prefetch_commit: the unconsumed words in front of those produced ahead)
        bool move_c = false, wait_prev_c = false;
        for (int w = 0; w < W; ++w)
            for (int s = 0; s < 2; ++s) {
                const int k = 2 * w + s;
                h_move_c[k] = iss::MtMoveJob{nullptr, nullptr, 0u, 0u};
                if (!n_w[w] || !pf[k].on) continue;
                if (pf[k].append) {
                    t.fill[k] += (size_t)pf[k].blocks * 624;
                    if (dbg) { dbg_filled[s] += (uint64_t)pf[k].blocks * 624; ++dbg_appends; }
                    continue;
                }
                const size_t left = t.fill[k] - t.used[k];  // <= pf.at
                const int nxt = (t.cur[k] + 1) % MT_SET_BUFS;
                h_move_c[k] = iss::MtMoveJob{bufp(w, s, t.cur[k]) + t.used[k], bufp(w, s, nxt) + (pf[k].at - left), (uint32_t)left, 0u};
                wait_prev_c |= read_by_last_turn(k, nxt);
                if (dbg) { dbg_moved[s] += left; dbg_filled[s] += (uint64_t)pf[k].blocks * 624; ++dbg_moves; dbg_big += left > want[k] / 4; }
                t.cur[k] = nxt;
                t.used[k] = pf[k].at - left;
                t.fill[k] = pf[k].at + (size_t)pf[k].blocks * 624;
                move_c = true;
            }
        if (move_c) {
            if (wait_prev_c) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, t.ev_emit[par ^ 1], 0));  // (a target the emitter of the turn before reads)
            HIP_TRY(ctx, hipMemcpyAsync(dev_of(h_move_c), h_move_c, 2 * (size_t)W * sizeof(iss::MtMoveJob), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(iss::k_mt_move_w, dim3(2 * W, iss::MOVE_BLOCKS), dim3(256), 0, ctx->stream, reinterpret_cast<const iss::MtMoveJob *>(dev_of(h_move_c)));
        }
        HIP_TRY(ctx, hipGetLastError());
        ++dbg_turns;
    }
    if (dbg) {
        for (int w = 0; w < W; ++w) dbg_pairs += (uint64_t)ws[w].done;
        fprintf(stderr, "[mt set] turns ended starved %llu, one-pair walker turns %llu, walks behind a turn skipped for words %llu, fills in front of a turn %llu\n",
                (unsigned long long)dbg_starved, (unsigned long long)dbg_own, (unsigned long long)dbg_skip, (unsigned long long)dbg_ensure);
        fprintf(stderr, "[mt set] W %d turn %lld: %llu turns, %llu pairs, %llu to the walker; %llu appended, commits with a move %llu (%llu moved > want / 4); words moved py %llu np %llu, "
                        "produced ahead py %llu np %llu\n", W, (long long)t.ch, (unsigned long long)dbg_turns, (unsigned long long)dbg_pairs,
                (unsigned long long)dbg_bounce, (unsigned long long)dbg_appends, (unsigned long long)dbg_moves, (unsigned long long)dbg_big, (unsigned long long)dbg_moved[0],
                (unsigned long long)dbg_moved[1], (unsigned long long)dbg_filled[0], (unsigned long long)dbg_filled[1]);
    }
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, m.ev_fill, 0));
    for (auto &e : t.ev_emit) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, e, 0));  // (the rows are complete once the main stream is)
    for (int w = 0; w < W; ++w)
        if (n_done && ws[w].n) n_done[w] = ws[w].done;
    return 0;
}

/* iss_mt_peek for worker w of the set (tests: the stream positions after a run) */
int iss_mt_workers_peek(iss_ctx *ctx, int32_t worker, uint32_t *py_words, uint32_t *np_words, int32_t n) {
    if (!ctx || worker < 0 || worker >= ctx->mts.W || n < 0 || n > 624) return fail(ctx, ISS_E_INVALID, "iss_mt_workers_peek: bad argument");
    if (!ctx->have_model) return fail(ctx, ISS_E_INVALID, "iss_mt_workers_peek: upload a model first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    { int rc_ = mt_set_reserve(ctx); if (rc_) return rc_; }
    MtChainLoan loan(ctx, worker);
    return iss_mt_peek(ctx, py_words, np_words, n);
}

int iss_set_fragment(iss_ctx *ctx, int32_t enabled, double fragment_length, double fragment_sd) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    ctx->has_frag = enabled != 0;
    ctx->frag_mu = fragment_length;
    ctx->frag_sd = fragment_sd;
    return 0;
}

int iss_mutations_reserve(iss_ctx *ctx, int64_t capacity) {
    if (!ctx || capacity < 0 || capacity > 0x7fffffff) return fail(ctx, ISS_E_INVALID, "iss_mutations_reserve: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    if (ctx->d_pmut) (void)hipFree(ctx->d_pmut);
    ctx->d_pmut = nullptr;
    ctx->pmut_cap = 0;
    if (capacity) {
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, (size_t)capacity * sizeof(iss::MutRecord)));
        ctx->d_pmut = static_cast<iss::MutRecord *>(p);
        ctx->pmut_cap = capacity;
    }
    return 0;
}

int iss_mutations_download(iss_ctx *ctx, iss_mutation *out, int64_t capacity, int64_t *n_rows) {
    if (n_rows) *n_rows = 0;
    if (!ctx || capacity < 0) return fail(ctx, ISS_E_INVALID, "iss_mutations_download: bad argument");
    if (!ctx->d_pmut) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    uint32_t reserved = 0;
    HIP_TRY(ctx, hipMemcpy(&reserved, ctx->d_pmut_count, sizeof reserved, hipMemcpyDeviceToHost));
    if ((int64_t)reserved > ctx->pmut_cap) {
        if (n_rows) *n_rows = (int64_t)reserved;  // (the slots the call asked for: what a retry has to reserve)
        return fail(ctx, ISS_E_NOMEM, "mutation buffer too small for this call (reserve more with iss_mutations_reserve)");
    }
    std::vector<iss::MutRecord> rows(reserved);
    std::vector<uint32_t> flags((size_t)ctx->last_n);
    if (reserved) HIP_TRY(ctx, hipMemcpy(rows.data(), ctx->d_pmut, (size_t)reserved * sizeof(iss::MutRecord), hipMemcpyDeviceToHost));
    if (ctx->last_n)
        HIP_TRY(ctx, hipMemcpy(flags.data(), ctx->flags + ctx->last_row0, (size_t)ctx->last_n * sizeof(uint32_t),
                               hipMemcpyDeviceToHost));
    // keep: used slots; k_main's rows only for mates the fix-up did not rebuild.  Order: pair, mate, indel rows in
    // loop order (step, insertion slot / deletion) before the substitution rows in position order.
    std::vector<std::pair<uint64_t, uint32_t>> keyed;
    keyed.reserve(rows.size());
    for (uint32_t i = 0; i < rows.size(); ++i) {
        const iss::MutRecord &r = rows[i];
        if (r.pair < 0) continue;
        const int t = (uint8_t)r.type;
        const bool from_fixup = (t & 32) != 0;
        if (!from_fixup && (((flags[(size_t)r.pair] >> r.mate) | (flags[(size_t)r.pair] >> (2 + r.mate))) & 1u)) continue;
        const uint64_t phase = (t & 3) == 0 ? 1 : 0;
        const uint64_t key = ((uint64_t)(uint32_t)r.pair << 32) | ((uint64_t)(r.mate & 1) << 31) | (phase << 30) |
                             ((uint64_t)(uint16_t)r.position << 8) | (uint64_t)((t >> 2) & 7);
        keyed.emplace_back(key, i);
    }
    std::sort(keyed.begin(), keyed.end());
    if (n_rows) *n_rows = (int64_t)keyed.size();
    const int64_t n = std::min<int64_t>((int64_t)keyed.size(), capacity);
    for (int64_t i = 0; i < n && out; ++i) {
        const iss::MutRecord &r = rows[keyed[(size_t)i].second];
        out[i].pair = r.pair; out[i].mate = r.mate; out[i].type = (int8_t)(r.type & 3); out[i].position = r.position;
        out[i].ref = r.ref; out[i].alt = r.alt; out[i].quality = r.quality;
    }
    return 0;
}

int iss_mt_set_fragment(iss_ctx *ctx, int32_t enabled, double fragment_length, double fragment_sd) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    ctx->mt.has_frag = enabled != 0;
    ctx->mt.frag_mu = fragment_length;
    ctx->mt.frag_sd = fragment_sd;
    return 0;
}

int iss_mt_mutations_reserve(iss_ctx *ctx, int64_t capacity) {
    if (!ctx || capacity < 0) return fail(ctx, ISS_E_INVALID, "iss_mt_mutations_reserve: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = sync_all(ctx); if (rc_) return rc_; }
    auto &m = ctx->mt;
    if (m.d_mut) (void)hipFree(m.d_mut);
    m.d_mut = nullptr;
    m.mut_cap = m.mut_n = 0;
    if (capacity) {
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, (size_t)capacity * sizeof(iss::MutRecord)));
        m.d_mut = static_cast<iss::MutRecord *>(p);
        m.mut_cap = capacity;
    }
    return 0;
}

int iss_mt_mutations_download(iss_ctx *ctx, iss_mutation *out, int64_t capacity, int64_t *n_total) {
    if (!ctx || capacity < 0) return fail(ctx, ISS_E_INVALID, "iss_mt_mutations_download: bad argument");
    static_assert(sizeof(iss_mutation) == sizeof(iss::MutRecord), "ABI and device mutation records differ");
    auto &m = ctx->mt;
    if (n_total) *n_total = m.mut_n;
    const int64_t n = std::min(std::min(m.mut_n, m.mut_cap), capacity);
    if (n > 0 && out) HIP_TRY(ctx, hipMemcpy(out, m.d_mut, (size_t)n * sizeof(iss::MutRecord), hipMemcpyDeviceToHost));
    return 0;
}

int iss_mt_path_counts(iss_ctx *ctx, int64_t *n_resolved, int64_t *n_walked) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    if (n_resolved) *n_resolved = ctx->mt.n_resolved + ctx->mts.n_resolved;  // (single-worker calls + the worker set)
    if (n_walked) *n_walked = ctx->mt.n_walked + ctx->mts.n_walked;
    return 0;
}

int iss_mt_peek(iss_ctx *ctx, uint32_t *py_words, uint32_t *np_words, int32_t n) {
    if (!ctx || !ctx->mt.seeded || n < 0 || n > 624) return fail(ctx, ISS_E_INVALID, "iss_mt_peek: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc_ = mt_reserve(ctx, 4 * 624, 4 * 624); if (rc_) return rc_; }
    const size_t want[2] = {(size_t)n, (size_t)n};
    { int rc_ = mt_ensure(ctx, want); if (rc_) return rc_; }
    auto &m = ctx->mt;
    if (py_words) HIP_TRY(ctx, hipMemcpyAsync(py_words, m.buf[0][m.cur[0]] + m.used[0], (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (np_words) HIP_TRY(ctx, hipMemcpyAsync(np_words, m.buf[1][m.cur[1]] + m.used[1], (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ------------------------------------------------------------------ FASTQ formatting (host)
// The rows of n_items work items -> FASTQ text (or gzip members) on their way to the two files.
// cpu_numbers / file_off (iss_fastq_emit_scatter): per item its worker's number and where its text goes in both files; else every
// item is worker cpu_number's and the text follows what the files hold.
static int fastq_emit_core(iss_ctx *ctx, int fd_r1, int fd_r2, int32_t n_items, const char *const *record_ids, const int64_t *first_i,
                           const int64_t *first_pair, const int64_t *n_pairs, int32_t cpu_number, int32_t n_threads,
                           const int32_t *cpu_numbers = nullptr, const int64_t *file_off = nullptr) {
    if (!ctx || !ctx->have_model || n_items < 0 || cpu_number < 0 || fd_r1 < 0 || fd_r2 < 0)
        return fail(ctx, ISS_E_INVALID, "iss_fastq_emit: bad argument");
    if (file_off && ctx->fq.gzip) return fail(ctx, ISS_E_INVALID, "iss_fastq_emit_scatter: text mode only (a gzip member's size is not known up front)");
    const iss::DevModel &M = ctx->M;
    iss::FastqArgs A{};
    std::vector<int64_t> scatter;
    A.row = M.row;
    A.RL = M.RL;
    std::vector<iss::FastqItem> items;
    std::string ids;
    size_t bytes = 0, rec_len = 0;  // rec_len: record length of the item with the most pairs (the distance of its "previous record")
    int64_t n_records = 0, most = 0;
    for (int32_t k = 0; k < n_items; ++k) {
        if (!record_ids[k] || first_i[k] < 0 || first_pair[k] < 0 || n_pairs[k] < 0 || first_pair[k] + n_pairs[k] > ctx->capacity)
            return fail(ctx, ISS_E_INVALID, "iss_fastq_emit: bad argument");
        const size_t idlen = strlen(record_ids[k]);
        if (idlen > FASTQ_ID_MAX) return fail(ctx, ISS_E_INVALID, "iss_fastq_emit: record id longer than 4096 bytes");
        if (cpu_numbers && cpu_numbers[k] < 0) return fail(ctx, ISS_E_INVALID, "iss_fastq_emit: bad argument");
        if (file_off && file_off[k] < 0) return fail(ctx, ISS_E_INVALID, "iss_fastq_emit: bad argument");
        if (n_pairs[k] == 0) continue;
        iss::FastqItem it{};
        it.cpu_len = (int32_t)snprintf(it.cpu, sizeof it.cpu, "%d", cpu_numbers ? cpu_numbers[k] : cpu_number);
        if (file_off) scatter.push_back(file_off[k]);
        it.first_i = (uint64_t)first_i[k];
        it.before_first = iss::digits_before(it.first_i);
        it.text_off = bytes;
        it.first_pair = first_pair[k];
        it.rec_first = n_records;
        it.id_off = (uint32_t)ids.size();
        it.id_len = (int32_t)idlen;
        ids.append(record_ids[k], idlen);
        const size_t C = idlen + (size_t)it.cpu_len + 2 * (size_t)M.RL + 10;
        bytes += (size_t)n_pairs[k] * C + (size_t)(iss::digits_before(it.first_i + (uint64_t)n_pairs[k]) - it.before_first);
        n_records += n_pairs[k];
        if (n_pairs[k] > most) {
            most = n_pairs[k];
            int dg = 1;
            for (uint64_t v = it.first_i + (uint64_t)n_pairs[k] - 1; v >= 10; v /= 10) ++dg;
            rec_len = C + (size_t)dg;
        }
        items.push_back(it);
    }
    if (items.empty()) return 0;
    A.n_items = (int32_t)items.size();
    A.n_records = n_records;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FastqPipe &q = ctx->fq;
    if (!q.ready) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&q.copy_stream, hipStreamNonBlocking));
        for (auto &e : q.ev_fmt) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : q.ev_copy) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        q.stop = false;
        q.writer = std::thread(fastq_writer_loop, ctx);
        q.ready = true;
    }
    if (q.fd[0] != fd_r1 || q.fd[1] != fd_r2) {
        { int rc_ = fastq_flush(ctx); if (rc_) return rc_; }
        q.fd[0] = fd_r1; q.fd[1] = fd_r2;
        for (int m = 0; m < 2; ++m) {
            const off_t at = lseek(q.fd[m], 0, SEEK_CUR);
            if (at < 0) return fail(ctx, ISS_E_IO, std::string("lseek failed: ") + strerror(errno));
            q.off[m] = q.attached_off[m] = at;
            q.accounted[m] = 0;
        }
    }
    const uint32_t n_blocks = (uint32_t)((bytes + iss::DEFLATE_BLOCK - 1) / iss::DEFLATE_BLOCK);
    // compressed bytes of a batch: its own Huffman code never needs more than 8 bits per byte plus rounding; the
    // smoothing of the counts (every symbol keeps a code) and the block headers are covered by the margin
    auto comp_bytes = [](size_t text, size_t blocks) { return text + text / 8 + blocks * 320 + 64; };
    if (bytes > q.cap || (q.gzip && (comp_bytes(bytes, n_blocks) > q.comp_cap || n_blocks > q.blocks_cap))) {
        { int rc_ = fastq_flush_keep(ctx); if (rc_) return rc_; }
        fastq_free_buffers(ctx);
        // (pinned allocations are slow: leave room for longer ids and pair numbers instead of growing batch by batch)
        const size_t cap = bytes + bytes / 8 + (1u << 20);
        const size_t cap_blocks = (cap + iss::DEFLATE_BLOCK - 1) / iss::DEFLATE_BLOCK;
        const size_t comp_cap = comp_bytes(cap, cap_blocks);
        const size_t host_bytes = q.gzip ? comp_cap : cap;
        for (auto &sl : q.d_text) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipMalloc(&v, cap + 16)); p = static_cast<uint8_t *>(v); }
        for (auto &sl : q.h_text) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipHostMalloc(&v, host_bytes, hipHostMallocDefault)); p = static_cast<uint8_t *>(v); }
        q.cap = cap;
        if (q.gzip) {
            for (auto &sl : q.d_comp) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipMalloc(&v, comp_cap)); p = static_cast<uint8_t *>(v); }
            for (auto &sl : q.d_bbytes) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipMalloc(&v, cap_blocks * 4)); p = static_cast<uint32_t *>(v); }
            for (auto &sl : q.d_bcrc) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipMalloc(&v, cap_blocks * 4)); p = static_cast<uint32_t *>(v); }
            for (auto &sl : q.d_boff) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipMalloc(&v, (cap_blocks + 1) * 8)); p = static_cast<uint64_t *>(v); }
            for (auto &sl : q.h_bcrc) for (auto &p : sl) { void *v = nullptr; HIP_TRY(ctx, hipHostMalloc(&v, cap_blocks * 4, hipHostMallocDefault)); p = static_cast<uint32_t *>(v); }
            q.comp_cap = comp_cap;
            q.blocks_cap = (uint32_t)cap_blocks;
        }
    }
    if (q.gzip && !q.d_code[0][0]) {  // fixed-size state of the compressed mode, once
        HIP_TRY(ctx, hipStreamCreateWithFlags(&q.data_stream, hipStreamNonBlocking));
        iss::crc_shift_operator(iss::DEFLATE_BLOCK, q.op_block);
        iss::DeflateCode init{};
        for (int k = 0; k < 8; ++k) iss::crc_shift_operator((uint64_t)128 << k, init.crc_shift[k]);
        for (int sl = 0; sl < 2; ++sl)
            for (int m = 0; m < 2; ++m) {
                void *v = nullptr;
                HIP_TRY(ctx, hipMalloc(&v, (iss::DEFLATE_SYMS + 7) * 4));
                q.d_hist[sl][m] = static_cast<uint32_t *>(v);
                HIP_TRY(ctx, hipMalloc(&v, sizeof(iss::DeflateCode)));
                q.d_code[sl][m] = static_cast<iss::DeflateCode *>(v);
                HIP_TRY(ctx, hipMemcpy(v, &init, sizeof init, hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipHostMalloc(&v, 64, hipHostMallocDefault));
                q.h_total[sl][m] = static_cast<uint64_t *>(v);
            }
    }
    const int slot = q.next;
    {
        std::unique_lock<std::mutex> lk(q.mu);
        q.cv.wait(lk, [&] { return !q.busy[slot]; });
        if (!q.error.empty()) { const std::string e = q.error; q.error.clear(); return fail(ctx, ISS_E_IO, e); }
    }
    if (items.size() > q.items_cap[slot] || ids.size() + 1 > q.ids_cap[slot]) {  // (the slot is free: nothing reads its tables)
        if (q.h_items[slot]) (void)hipHostFree(q.h_items[slot]);
        if (q.d_items[slot]) (void)hipFree(q.d_items[slot]);
        if (q.h_ids[slot]) (void)hipHostFree(q.h_ids[slot]);
        if (q.d_ids[slot]) (void)hipFree(q.d_ids[slot]);
        q.h_items[slot] = q.d_items[slot] = nullptr;
        q.h_ids[slot] = q.d_ids[slot] = nullptr;
        const size_t ic = std::max<size_t>(64, 2 * items.size()), dc = std::max<size_t>(8192, 2 * (ids.size() + 1));
        void *v = nullptr;
        HIP_TRY(ctx, hipHostMalloc(&v, ic * sizeof(iss::FastqItem), hipHostMallocDefault));
        q.h_items[slot] = static_cast<iss::FastqItem *>(v);
        HIP_TRY(ctx, hipMalloc(&v, ic * sizeof(iss::FastqItem)));
        q.d_items[slot] = static_cast<iss::FastqItem *>(v);
        HIP_TRY(ctx, hipHostMalloc(&v, dc, hipHostMallocDefault));
        q.h_ids[slot] = static_cast<char *>(v);
        HIP_TRY(ctx, hipMalloc(&v, dc));
        q.d_ids[slot] = static_cast<char *>(v);
        q.items_cap[slot] = ic;
        q.ids_cap[slot] = dc;
    }
    memcpy(q.h_items[slot], items.data(), items.size() * sizeof(iss::FastqItem));
    memcpy(q.h_ids[slot], ids.data(), ids.size());
    HIP_TRY(ctx, hipMemcpyAsync(q.d_items[slot], q.h_items[slot], items.size() * sizeof(iss::FastqItem), hipMemcpyHostToDevice, ctx->stream));
    if (!ids.empty()) HIP_TRY(ctx, hipMemcpyAsync(q.d_ids[slot], q.h_ids[slot], ids.size(), hipMemcpyHostToDevice, ctx->stream));
    A.items = q.d_items[slot];
    A.ids = q.d_ids[slot];
    for (int m = 0; m < 2; ++m) {
        A.base[m] = ctx->out[2 * m];
        A.qual[m] = ctx->out[2 * m + 1];
        A.text[m] = q.d_text[slot][m];
    }
    hipLaunchKernelGGL(iss::k_fastq_format, dim3((unsigned)((n_records + iss::FASTQ_WAVES - 1) / iss::FASTQ_WAVES), 2),
                       dim3(64 * iss::FASTQ_WAVES), 0, ctx->stream, A);
    if (q.gzip) {  // the text stays on the device: histogram -> code -> block sizes + CRCs -> offsets -> bits (iss_deflate.hip.h)
        iss::DeflateArgs D{};
        D.n_bytes = bytes;
        D.n_blocks = n_blocks;
        D.out_cap = q.comp_cap;
        // the record length most records of this call have: the distance of the "previous record" matches
        if (rec_len >= 8 && rec_len <= 32768 && !getenv("ISS_DEFLATE_RUNS_ONLY")) {
            D.dist = (uint32_t)rec_len;
            iss::deflate_dist_code(D.dist, &D.dist_sym, &D.dist_ebits, &D.dist_eval);
        }
        for (int m = 0; m < 2; ++m) {
            D.text[m] = q.d_text[slot][m];
            D.hist[m] = q.d_hist[slot][m];
            D.code[m] = q.d_code[slot][m];
            D.block_bytes[m] = q.d_bbytes[slot][m];
            D.block_crc[m] = q.d_bcrc[slot][m];
            D.block_off[m] = q.d_boff[slot][m];
            D.out[m] = q.d_comp[slot][m];
            HIP_TRY(ctx, hipMemsetAsync(q.d_hist[slot][m], 0, iss::DEFLATE_SYMS * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(q.d_comp[slot][m], 0, std::min(q.comp_cap, comp_bytes(bytes, n_blocks)), ctx->stream));
        }
        const unsigned hist_grid = (unsigned)std::min<uint64_t>(2048, (bytes / 16 + iss::DEFLATE_THREADS - 1) / iss::DEFLATE_THREADS + 1);
        hipLaunchKernelGGL(iss::k_deflate_hist, dim3(hist_grid, 2), dim3(iss::DEFLATE_THREADS), 0, ctx->stream, D);
        hipLaunchKernelGGL(iss::k_deflate_build, dim3(2), dim3(64), 0, ctx->stream, D);
        hipLaunchKernelGGL(iss::k_deflate_len, dim3(n_blocks, 2), dim3(iss::DEFLATE_THREADS), 0, ctx->stream, D);
        hipLaunchKernelGGL(iss::k_deflate_scan, dim3(2), dim3(1024), 0, ctx->stream, D);
        hipLaunchKernelGGL(iss::k_deflate_encode, dim3(n_blocks, 2), dim3(iss::DEFLATE_THREADS), 0, ctx->stream, D);
    }
    HIP_TRY(ctx, hipEventRecord(q.ev_fmt[slot], ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(q.copy_stream, q.ev_fmt[slot], 0));
    for (int m = 0; m < 2; ++m) {
        if (q.gzip) {  // sizes and CRCs now; the writer thread fetches the bytes once it knows how many there are
            HIP_TRY(ctx, hipMemcpyAsync(q.h_total[slot][m], q.d_boff[slot][m] + n_blocks, 8, hipMemcpyDeviceToHost, q.copy_stream));
            HIP_TRY(ctx, hipMemcpyAsync(q.h_bcrc[slot][m], q.d_bcrc[slot][m], (size_t)n_blocks * 4, hipMemcpyDeviceToHost, q.copy_stream));
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(q.h_text[slot][m], q.d_text[slot][m], bytes, hipMemcpyDeviceToHost, q.copy_stream));
        }
    }
    HIP_TRY(ctx, hipEventRecord(q.ev_copy[slot], q.copy_stream));
    {
        std::lock_guard<std::mutex> lk(q.mu);
        if (const char *e = getenv("ISS_FASTQ_PIECES")) n_threads = atoi(e);  // tuning aid
        FastqJob job{slot, bytes, {q.fd[0], q.fd[1]}, {q.off[0], q.off[1]}, std::max(1, std::min<int>(n_threads, 128)),
                     q.gzip != 0, n_blocks, {}};
        if (!q.gzip) {  // (compressed members: the writer thread advances the offsets by what it wrote)
            for (const auto &it : items) job.item_off.push_back(it.text_off);
            // (scattered items lie where the caller says: the files' running offsets stay where they are)
            if (scatter.empty()) for (int m = 0; m < 2; ++m) { q.off[m] += (int64_t)bytes; q.accounted[m] += (int64_t)bytes; }
            job.item_file_off = std::move(scatter);
        }
        q.jobs.push_back(std::move(job));
        q.busy[slot] = true;
    }
    q.cv.notify_all();
    q.next ^= 1;
    return 0;
}

int iss_fastq_emit(iss_ctx *ctx, int fd_r1, int fd_r2, const char *record_id, int64_t first_i, int32_t cpu_number,
                   int64_t first_pair, int64_t n_pairs, int32_t n_threads) {
    if (!record_id) return fail(ctx, ISS_E_INVALID, "iss_fastq_emit: bad argument");
    return fastq_emit_core(ctx, fd_r1, fd_r2, 1, &record_id, &first_i, &first_pair, &n_pairs, cpu_number, n_threads);
}

int iss_fastq_emit_batch(iss_ctx *ctx, int fd_r1, int fd_r2, int32_t n_items, const char *const *record_ids, const int64_t *first_i,
                         const int64_t *first_pair, const int64_t *n_pairs, int32_t cpu_number) {
    if (n_items && (!record_ids || !first_i || !first_pair || !n_pairs)) return fail(ctx, ISS_E_INVALID, "iss_fastq_emit_batch: bad argument");
    return fastq_emit_core(ctx, fd_r1, fd_r2, n_items, record_ids, first_i, first_pair, n_pairs, cpu_number, 1);
}

int iss_fastq_emit_scatter(iss_ctx *ctx, int fd_r1, int fd_r2, int32_t n_items, const char *const *record_ids, const int64_t *first_i,
                           const int64_t *first_pair, const int64_t *n_pairs, const int32_t *cpu_numbers, const int64_t *file_off,
                           int32_t n_threads) {
    if (n_items && (!record_ids || !first_i || !first_pair || !n_pairs || !cpu_numbers || !file_off))
        return fail(ctx, ISS_E_INVALID, "iss_fastq_emit_scatter: bad argument");
    return fastq_emit_core(ctx, fd_r1, fd_r2, n_items, record_ids, first_i, first_pair, n_pairs, 0, n_threads, cpu_numbers, file_off);
}

int iss_fastq_flush(iss_ctx *ctx) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    return fastq_flush(ctx);
}

int iss_fastq_compress(iss_ctx *ctx, int32_t mode) {
    if (!ctx) return fail(nullptr, ISS_E_INVALID, "ctx is NULL");
    if (mode != 0 && mode != 1) return fail(ctx, ISS_E_INVALID, "iss_fastq_compress: mode must be 0 (text) or 1 (gzip members)");
    if (ctx->fq.gzip == mode) return 0;
    { int rc_ = fastq_flush(ctx); if (rc_) return rc_; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fastq_free_buffers(ctx);  // the host buffers have another size in the other mode
    ctx->fq.gzip = mode;
    return 0;
}

int iss_deflate_code_build(const uint32_t *hist, uint32_t record_distance, uint32_t *entry, uint32_t *hdr_bits,
                           uint32_t *hdr_words, uint32_t *dist_code) {
    if (!hist || !entry || !hdr_bits || !hdr_words || !dist_code || record_distance > 32768) return ISS_E_INVALID;
    dist_code[0] = dist_code[1] = dist_code[2] = 0;
    if (record_distance) iss::deflate_dist_code(record_distance, &dist_code[0], &dist_code[1], &dist_code[2]);
    static iss::DeflateCode c;  // (large for a stack frame; the function is a test hook, not re-entrant)
    static iss::DeflateWork ws;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    iss::deflate_build_code(hist, &ws, dist_code[0], 0, 1, iss::DeflateNoSync());
    iss::deflate_store_code(&ws, &c, 0, 1);
    memcpy(entry, c.entry, sizeof c.entry);
    *hdr_bits = c.hdr_bits;
    memcpy(hdr_words, c.hdr, sizeof c.hdr);
    return 0;
}

static int write_all(int fd, const char *p, size_t n) {
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; return -1; }
        p += w; n -= (size_t)w;
    }
    return 0;
}

static size_t fmt_u64(char *dst, uint64_t v) {
    char tmp[24];
    size_t n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (size_t i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}

int iss_fastq_write(int fd_r1, int fd_r2, const char *record_id, int64_t first_i, int32_t cpu_number, int64_t n_pairs,
                    int32_t read_length, int32_t pitch, const uint8_t *r1_base, const uint8_t *r1_qual,
                    const uint8_t *r2_base, const uint8_t *r2_qual, int32_t n_threads) {
    if (!record_id || n_pairs < 0 || read_length < 1 || pitch < read_length || cpu_number < 0 || first_i < 0)
        return fail(nullptr, ISS_E_INVALID, "iss_fastq_write: bad argument");
    const size_t idlen = strlen(record_id);
    const size_t max_rec = 1 + idlen + 1 + 20 + 1 + 11 + 2 + 1 + (size_t)read_length + 3 + (size_t)read_length + 1;
    const int64_t chunk = 1 << 14;
    n_threads = std::max(1, std::min<int32_t>(n_threads, 64));
    char cpu_txt[16];
    const size_t cpu_len = fmt_u64(cpu_txt, (uint64_t)cpu_number);
    for (int64_t base = 0; base < n_pairs; base += chunk * n_threads) {
        const int nt = (int)std::min<int64_t>(n_threads, (n_pairs - base + chunk - 1) / chunk);
        std::vector<std::vector<char>> buf(2 * nt);
        std::vector<size_t> used(2 * nt, 0);
        auto work = [&](int t) {
            const int64_t lo = base + (int64_t)t * chunk, hi = std::min(n_pairs, lo + chunk);
            for (int mate = 0; mate < 2; ++mate) {
                std::vector<char> &b = buf[2 * t + mate];
                b.resize((size_t)(hi - lo) * max_rec);
                char *w = b.data();
                const uint8_t *bases = mate ? r2_base : r1_base, *quals = mate ? r2_qual : r1_qual;
                for (int64_t i = lo; i < hi; ++i) {
                    *w++ = '@';
                    memcpy(w, record_id, idlen); w += idlen;
                    *w++ = '_';
                    w += fmt_u64(w, (uint64_t)(first_i + i));
                    *w++ = '_';
                    memcpy(w, cpu_txt, cpu_len); w += cpu_len;
                    *w++ = '/'; *w++ = (char)('1' + mate); *w++ = '\n';
                    memcpy(w, bases + (size_t)i * pitch, (size_t)read_length); w += read_length;
                    *w++ = '\n'; *w++ = '+'; *w++ = '\n';
                    const uint8_t *q = quals + (size_t)i * pitch;
                    for (int k = 0; k < read_length; ++k) w[k] = (char)(33 + q[k]);
                    w += read_length;
                    *w++ = '\n';
                }
                used[2 * t + mate] = (size_t)(w - b.data());
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < nt; ++t) {
            if (write_all(fd_r1, buf[2 * t].data(), used[2 * t]) || write_all(fd_r2, buf[2 * t + 1].data(), used[2 * t + 1]))
                return fail(nullptr, ISS_E_IO, std::string("write failed: ") + strerror(errno));
        }
    }
    return 0;
}

}  // extern "C"
