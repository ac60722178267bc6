// iss_host_util.hip.h -- helpers every entry point uses: error reporting, checked uploads, the environment switches, what a
// context frees, the choice of the hot kernel's instantiation (k_main / k_main_g), its LDS size, timing, synchronisation.
#pragma once

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

// dynamic LDS of k_main: quality rows + deferred-work queues
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

}  // namespace
