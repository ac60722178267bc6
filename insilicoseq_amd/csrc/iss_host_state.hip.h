// iss_host_state.hip.h -- host-side state of the engine: the FASTQ pipeline's job / pipe records, timed launches, and struct iss_ctx
// (one per GPU: streams, uploaded model and genomes, output rows, MT-mode chains and worker sets).  Included by iss_mi355x.hip.
#pragma once

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
