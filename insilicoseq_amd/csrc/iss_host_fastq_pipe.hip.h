// iss_host_fastq_pipe.hip.h -- the FASTQ pipeline behind iss_fastq_emit*: writer thread (text or gzip members, pieces written
// with pwrite at their final offsets), flush with its invariants, buffers.
#pragma once

namespace {

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
            // scattered items (the workers of a set, every one at its own place of the final files): the text of item k is
            // [item_off[k], item_off[k + 1]).  ONE thread per file unless the caller asks for more: writes to one tmpfs file
            // serialise on its inode, and threads that queue there cost more than they add -- 64 M pairs of the whole command
            // `generate --rng mt --cpus 64`: 4.6 s with one thread per file, 5.8 with two, 7.6-8.0 with four or eight
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
