// iss_api_model.hip.h -- C ABI: iss_model_upload (host tables -> the integer thresholds and compressed rows the kernels read, position tiles, guide bits),
// genome uploads (ASCII or 2-bit codes), output rows.
#pragma once

extern "C" {

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

}  // extern "C"
