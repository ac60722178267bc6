// iss_api_fastq.hip.h -- C ABI: FASTQ text / gzip members built on the device (iss_fastq_emit*, iss_fastq_flush, iss_fastq_compress) and the host
// formatter (iss_fastq_write).
#pragma once

extern "C" {

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
