// iss_api_generate.hip.h -- C ABI: the Philox path -- iss_generate / iss_generate_batch (launch sequencing of k_setup, the indel kernels, k_main / k_main_g,
// k_indel_fixup), downloads, the ErrorModel methods as batched entries, timing and counters.
#pragma once

extern "C" {

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

}  // extern "C"
