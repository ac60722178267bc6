// iss_host_mt_streams.hip.h -- reference-compatible MT mode, host side of the streams: MT19937 seeding as CPython / numpy do it,
// the fill kernel's launches one chunk ahead, libm evaluations the device hands back.
#pragma once

namespace {

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

}  // namespace
