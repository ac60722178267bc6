// iss_api_context.hip.h -- C ABI: version, build id, errors, context creation / destruction, the caller's stream.
#pragma once

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

}  // extern "C"
