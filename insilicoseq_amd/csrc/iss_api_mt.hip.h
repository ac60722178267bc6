// iss_api_mt.hip.h -- C ABI: the reference-identical mode (rng="mt") -- one worker per context (iss_generate_mt) and W workers side by side
// (iss_generate_mt_workers), custom fragment lengths, --store_mutations rows.
#pragma once

extern "C" {

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

}  // extern "C"
