// capi.inc.hpp -- extern "C": context management, options, profiling, device-resident, stage-level and host-pointer entry points
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

#ifdef MI_TEST_HOOKS
const char* mi_version(void) { return "mi_lumaeq 0.2 (gfx950) +test-hooks"; }
#else
const char* mi_version(void) { return "mi_lumaeq 0.2 (gfx950)"; }
#endif

const char* mi_status_str(mi_status s)
{
    switch (s) {
        case MI_OK: return "MI_OK";
        case MI_ERR_BAD_ARG: return "MI_ERR_BAD_ARG";
        case MI_ERR_UNSUPPORTED: return "MI_ERR_UNSUPPORTED";
        case MI_ERR_HIP: return "MI_ERR_HIP";
        case MI_ERR_OOM: return "MI_ERR_OOM";
        case MI_ERR_NO_DEVICE: return "MI_ERR_NO_DEVICE";
        case MI_ERR_BUSY: return "MI_ERR_BUSY";
    }
    return "MI_ERR_?";
}

const char* mi_kernel_name(int k)
{
    static const char* names[MI_K_COUNT] = {"hist_partial_kernel", "equalize_lut_kernel", "lut_apply_kernel",
                                            "tile_hist_kernel", "tile_lut_kernel", "clahe_interp_kernel", "equalize_fused_kernel", "color_kernel",
                                            "fused_finish_kernel", "analyze_diff_kernel"};
    return (k >= 0 && k < MI_K_COUNT) ? names[k] : "?";
}

int mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

mi_status mi_device_pci_bus_id(int device, char* buf, size_t buf_len)
{
    if (!buf || buf_len < 13) return MI_ERR_BAD_ARG;
    buf[0] = 0;
    const int n = mi_device_count();
    if (n <= 0 || device < 0 || device >= n) return MI_ERR_NO_DEVICE;
    if (hipDeviceGetPCIBusId(buf, (int)std::min<size_t>(buf_len, 64), device) != hipSuccess) { (void)hipGetLastError(); buf[0] = 0; return MI_ERR_HIP; }
    return MI_OK;
}

mi_status mi_thread_bind_near_device(int device, mi_numa_binding* out)
{
    mi_numa_binding b{};
    b.node = -1;
    auto finish = [&](mi_status st, const std::string& why) {
        snprintf(b.why, sizeof b.why, "%s", why.c_str());
        if (out) *out = b;
        return st;
    };
    char bdf[64];
    const mi_status st = mi_device_pci_bus_id(device, bdf, sizeof bdf);
    if (st) return finish(st, "no PCI address for this device");
    const char* off = getenv("MI_LUMAEQ_NUMA_BIND");
    const bool apply = !(off && atoi(off) == 0);
    const mi_host::NumaBinding r = mi_host::bind_thread_near_pci(bdf, "/sys", apply);
    b.node = r.node;
    b.cpus = apply ? r.cpus : 0;
    return finish(MI_OK, apply ? r.why : r.why + " (MI_LUMAEQ_NUMA_BIND=0: not applied)");
}

// Once per process: were the kernels built with separately rounded float steps?  (0 = not yet known, 1 = yes, -1 = no)
static std::atomic<int> g_contract_ok{0};
static int contract_probe()
{
    int v = g_contract_ok.load();
    if (v != 0) return v;
    float* d = nullptr;
    float* hp = nullptr;                                          // pinned, like every host buffer this library gives the runtime
    if (hipMalloc((void**)&d, sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipHostMalloc((void**)&hp, sizeof(float), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return 0; }
    *hp = 1.0f;
    const float a = 1.0f + 1.0f / 4096.0f;
    hipLaunchKernelGGL(contract_probe_kernel, dim3(1), dim3(1), 0, nullptr, a, a, -(1.0f + 1.0f / 2048.0f), d);
    const bool ran = hipGetLastError() == hipSuccess && hipMemcpy(hp, d, sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
    const float h = *hp;
    (void)hipFree(d);
    (void)hipHostFree(hp);
    if (!ran) { (void)hipGetLastError(); return 0; }
    v = h == 0.0f ? 1 : -1;
    g_contract_ok.store(v);
    return v;
}

mi_status mi_ctx_create(int device, mi_ctx** out)
{
    if (!out) return MI_ERR_BAD_ARG;
    *out = nullptr;
    const int n = mi_device_count();
    if (n <= 0 || device < 0 || device >= n) return MI_ERR_NO_DEVICE;
    if (hipSetDevice(device) == hipSuccess && contract_probe() < 0) {
        fprintf(stderr, "mi_lumaeq: this library was built without -ffp-contract=off (float multiply-adds are fused): its CLAHE / LUT "
                        "arithmetic would not be bit-exact; rebuild with opencv-opencl_amd/csrc/Makefile\n");
        return MI_ERR_UNSUPPORTED;
    }
    mi_ctx* c = new (std::nothrow) mi_ctx();
    if (!c) return MI_ERR_OOM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return MI_ERR_HIP;
    }
    // (the context's own stream is created by the first entry point that needs one -- ensure_stream(): a streaming worker's
    // context never does, and every stream a process creates competes for the runtime's few hardware queues, see pipe_stream_create)
    {   // one line of pinned, device-writable host memory: the finish kernel of the fused path reports repaired launches into it
        void* q = nullptr;
        if (hipHostMalloc(&q, 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess
            || hipEventCreateWithFlags(&c->ev_scratch, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (q) (void)hipHostFree(q);
            for (hipEvent_t e : {c->ev_scratch, c->ev_b, c->ev_k}) if (e) (void)hipEventDestroy(e);
            delete c;
            return MI_ERR_HIP;
        }
        c->h_mirror = (uint32_t*)q;
        memset(c->h_mirror, 0, 256);                             // [0..31] the fused path's words, [32..33] the 16-bit CLAHE's 14-bit-content hint
        c->h_mirror[33] = c->c16_seq;                            // "executed so far" starts far ahead of "such content last seen" (0): no mid kernel at first
    }
    // the 16-bit tile histogram uses 128 KiB of dynamic LDS (above the 64 KiB default limit)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_hist16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kHalf16 * (int)sizeof(uint32_t));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_hist12_kernel<kHist12Threads, kCopies12>), hipFuncAttributeMaxDynamicSharedMemorySize, kHist12Words * (int)sizeof(uint32_t));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(clahe_interp16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kInterp16Entries * (int)sizeof(uint2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(clahe_interp16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kInterp16Entries * (int)sizeof(uint2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(clahe_interp16_mid_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kInterp16MidEntries * (int)sizeof(uint2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(clahe_interp16_mid_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kInterp16MidEntries * (int)sizeof(uint2));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->cu_count = prop.multiProcessorCount;
    if (device < kMaxDevices) {
        if (g_fused_ctx_live[device].fetch_add(1) < kMaxFusedCtxPerDevice) c->fused_slot = true;
        else g_fused_ctx_live[device].fetch_sub(1);
    }
    if (const char* e = getenv("MI_LUMAEQ_FUSED")) c->fused_mode = atoi(e);
    if (const char* e = getenv("MI_LUMAEQ_FUSED_WGS_PER_CU")) c->fused_wgs_per_cu = std::max(1, std::min(8, atoi(e)));
    if (const char* e = getenv("MI_LUMAEQ_FUSED_VPT")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 20 || v == 24) c->fused_vpt = v; }
    if (const char* e = getenv("MI_LUMAEQ_FUSED_ACQUIRE")) c->fused_acquire = atoi(e) != 0;
    if (const char* e = getenv("MI_LUMAEQ_PIPE_COPY_STREAMS")) c->pipe_copy_streams = atoi(e) > 1 ? 2 : 1;      // A/B runs of nv12_stream
    if (const char* e = getenv("MI_LUMAEQ_PIPE_PRIVATE_STREAMS")) c->pipe_private_streams = atoi(e) != 0;
    if (const char* e = getenv("MI_LUMAEQ_HOST_COPY_STREAMS")) c->host_copy_streams = atoi(e) > 1 ? 2 : 1;
    if (const char* e = getenv("MI_LUMAEQ_HOST_COPY_THREADS")) c->host_copy_threads = atoi(e) > 1 ? 2 : 1;
    *out = c;
    return MI_OK;
}

void mi_ctx_destroy(mi_ctx* c)
{
    if (!c) return;
    if (c->fused_slot && c->device >= 0 && c->device < kMaxDevices) g_fused_ctx_live[c->device].fetch_sub(1);
    (void)hipSetDevice(c->device);
    // The device-resident forms are stream-ordered on streams the CALLER owns: a kernel of this context may still be reading the scratch
    // freed below or about to write the pinned mirror words (fused_finish_kernel -> h_mirror).  Wait for the whole device before the
    // first free -- explicitly, not through hipFree's implicit wait (contexts are destroyed at the end of a worker's life, never per frame).
    (void)hipDeviceSynchronize();
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    for (auto e : c->chunk_events) (void)hipEventDestroy(e);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_luts) (void)hipFree(c->d_luts);
    if (c->d_ghist) (void)hipFree(c->d_ghist);
    if (c->d_sync16) (void)hipFree(c->d_sync16);
    if (c->d_fused) (void)hipFree(c->d_fused);
    for (void* q : c->retired) (void)hipFree(q);
    if (c->d_planes) (void)hipFree(c->d_planes);
    if (c->d_c16) (void)hipFree(c->d_c16);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_mirror) (void)hipHostFree(c->h_mirror);
    for (hipEvent_t e : {c->ev_scratch, c->ev_b, c->ev_k}) if (e) (void)hipEventDestroy(e);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    delete c->crew;
    if (c->d_stage_in) (void)hipFree(c->d_stage_in);
    if (c->d_stage_out) (void)hipFree(c->d_stage_out);
    if (c->h_pin_in) (void)hipHostFree(c->h_pin_in);
    if (c->h_pin_out) (void)hipHostFree(c->h_pin_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mi_ctx_device(const mi_ctx* c) { return c ? c->device : -1; }
int mi_ctx_last_hip_error(const mi_ctx* c) { return c ? c->last_hip : 0; }
const char* mi_ctx_last_error_msg(const mi_ctx* c) { return c ? c->last_msg.c_str() : "null context"; }

mi_status mi_ctx_set_profiling(mi_ctx* c, int enabled)
{
    ENTER(c);
    c->profiling = enabled < 0 ? 0 : (enabled > 2 ? 1 : enabled);
    return MI_OK;
}

mi_status mi_ctx_profile_read(mi_ctx* c, mi_profile* out, int reset)
{
    ENTER(c);
    for (auto& p : c->pending) {
        HIPCHK(c, hipEventSynchronize(p.b));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, p.a, p.b));
        c->prof.total_ms[p.kernel] += ms;
        c->prof.launches[p.kernel] += 1;
        auto& sv = c->samples[p.kernel];
        if (sv.size() < 65536) sv.push_back(ms);
        else { sv[c->sample_pos[p.kernel]] = ms; c->sample_pos[p.kernel] = (c->sample_pos[p.kernel] + 1) & 65535; }
        c->free_events.push_back(p.a);
        c->free_events.push_back(p.b);
    }
    c->pending.clear();
    for (int k = 0; k < MI_K_COUNT; ++k) {
        std::vector<float> v = c->samples[k];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        auto q = [&](double f) { return (double)v[(size_t)std::min<double>((double)v.size() - 1, f * (double)(v.size() - 1) + 0.5)]; };
        c->prof.min_ms[k] = v.front(); c->prof.max_ms[k] = v.back();
        c->prof.p10_ms[k] = q(0.10); c->prof.p50_ms[k] = q(0.50); c->prof.p90_ms[k] = q(0.90);
    }
    if (out) *out = c->prof;
    if (reset) {
        c->prof = mi_profile{};
        for (int k = 0; k < MI_K_COUNT; ++k) { c->samples[k].clear(); c->sample_pos[k] = 0; }
    }
    return MI_OK;
}

mi_status mi_host_register(void* ptr, size_t bytes)
{
    if (!ptr || bytes == 0) return MI_ERR_BAD_ARG;
    if (mi_device_count() <= 0) return MI_ERR_NO_DEVICE;
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP; }
    g_pins.add(ptr, bytes);
    return MI_OK;
}

mi_status mi_host_unregister(void* ptr)
{
    if (!ptr) return MI_ERR_BAD_ARG;
    // hipHostUnregister may wait for the device: the registry calls it WITHOUT its lock (the range sits on the "being unpinned" list
    // meanwhile, so nobody judges it pinned), and keeps the range registered when the runtime refuses although the pages are still
    // pinned -- the caller can ask again.  A refusal because the runtime does not know the pages as pinned any more (the caller
    // unpinned them itself, a teardown did) drops the entry instead: kept, it could never be removed, and after the address was
    // freed and reused the registry would call memory pinned that nobody pinned.
    using R = mi_host::PinRegistry;
    switch (g_pins.remove(ptr, g_pending_dma, [](void* q) {
                const hipError_t e = hipHostUnregister(q);
                if (e == hipSuccess) return R::UNPINNED;
                (void)hipGetLastError();
                if (e == hipErrorHostMemoryNotRegistered) return R::NOT_PINNED_ANY_MORE;
                return runtime_says_pinned(q, 1) ? R::REFUSED : R::NOT_PINNED_ANY_MORE;      // any other error: ask what is true now
            })) {
        case R::REMOVED: return MI_OK;
        case R::ALREADY_UNPINNED: return MI_OK;      // the entry is gone and the pages are not pinned: what the caller asked for
        case R::NOT_REGISTERED: return MI_ERR_BAD_ARG;
        case R::BUSY: return MI_ERR_BUSY;
        default: return MI_ERR_HIP;
    }
}

// Options: include/mi_lumaeq.h documents the ones that change behaviour, include/mi_lumaeq_tuning.h the speed-only ones; the
// test hooks exist in libmi_lumaeq_test.so only (csrc/Makefile builds it with -DMI_TEST_HOOKS).
mi_status mi_ctx_set_option(mi_ctx* c, const char* name, int value)
{
    ENTER(c);
    if (!name) return fail(c, MI_ERR_BAD_ARG, "null option name");
    // ---- behaviour (mi_lumaeq.h)
    if (!strcmp(name, "fused")) { c->fused_mode = value; return MI_OK; }
    if (!strcmp(name, "fused_timeout_ms")) {
        c->fused_timeout_ms = std::max(1, value);
#ifdef MI_TEST_HOOKS
        c->fused_timeout_us = 0;
#endif
        return MI_OK;
    }
    if (!strcmp(name, "fused_demote_after")) { if (value < 0 || value > 32) return fail(c, MI_ERR_BAD_ARG, "fused_demote_after must be 0..32"); c->fused_demote_after = value; return MI_OK; }
    if (!strcmp(name, "fused_reprobe_ms")) { c->fused_reprobe_ms = c->fused_reprobe_ms_now = std::max(1, value); return MI_OK; }
    if (!strcmp(name, "clahe_fp_contract")) { c->clahe_fp_contract = value != 0; return MI_OK; }
    // ---- speed only (mi_lumaeq_tuning.h)
    if (!strcmp(name, "two_kernel_max_frames")) { if (value < 0 || value > 64) return fail(c, MI_ERR_BAD_ARG, "two_kernel_max_frames must be 0..64"); c->two_kernel_max_frames = value; return MI_OK; }
    if (!strcmp(name, "fused_wgs_per_cu")) { c->fused_wgs_per_cu = std::max(1, std::min(8, value)); return MI_OK; }
    if (!strcmp(name, "fused_vpt") && value == 0) { c->fused_vpt = kVPT; return MI_OK; }
    if (!strcmp(name, "fused_vpt")) { if (value != 8 && value != 16 && value != 20 && value != 24) return fail(c, MI_ERR_BAD_ARG, "fused_vpt must be 0 (default), 8, 16, 20 or 24"); c->fused_vpt = value; return MI_OK; }
    if (!strcmp(name, "fused_acquire")) { c->fused_acquire = value != 0; return MI_OK; }
    if (!strcmp(name, "bgr_fused")) { c->bgr_fused = value != 0; return MI_OK; }
    if (!strcmp(name, "clahe_hist_threads")) { if (value != 256 && value != 512) return fail(c, MI_ERR_BAD_ARG, "clahe_hist_threads must be 256 or 512"); c->clahe_hist_threads = value; return MI_OK; }
    if (!strcmp(name, "clahe_tiles_per_wg")) { if (value < 0 || value > 8) return fail(c, MI_ERR_BAD_ARG, "clahe_tiles_per_wg must be 0..8"); c->clahe_tiles_per_wg = value; return MI_OK; }
    if (!strcmp(name, "clahe_seg_pairs")) { if (value < 4 || value > 15) return fail(c, MI_ERR_BAD_ARG, "clahe_seg_pairs must be 4..15"); c->clahe_seg_pairs = value; return MI_OK; }
    if (!strcmp(name, "clahe_xcd_map")) { c->clahe_xcd_map = value != 0; return MI_OK; }
    if (!strcmp(name, "clahe_float_tables")) { c->clahe_float_tables = value != 0; return MI_OK; }
    if (!strcmp(name, "clahe16_fast12")) { c->clahe16_fast12 = value != 0; return MI_OK; }
    if (!strcmp(name, "clahe16_transposed")) { c->clahe16_transposed = value != 0; return MI_OK; }
    if (!strcmp(name, "clahe16_wide")) { c->clahe16_wide = value < 0 ? 0 : (value > 2 ? 2 : value); return MI_OK; }
    if (!strcmp(name, "pipe_copy_streams")) { if (value < 1 || value > 2) return fail(c, MI_ERR_BAD_ARG, "pipe_copy_streams must be 1 or 2"); c->pipe_copy_streams = value; return MI_OK; }
    if (!strcmp(name, "host_copy_streams")) { if (value < 1 || value > 2) return fail(c, MI_ERR_BAD_ARG, "host_copy_streams must be 1 or 2"); c->host_copy_streams = value; return MI_OK; }
    if (!strcmp(name, "host_copy_threads")) { if (value < 1 || value > 2) return fail(c, MI_ERR_BAD_ARG, "host_copy_threads must be 1 or 2"); c->host_copy_threads = value; return MI_OK; }
#ifdef MI_TEST_HOOKS
    // ---- test hooks (this is libmi_lumaeq_test.so)
    if (!strcmp(name, "fused_fault_inject")) { if (value < 0 || value > 3) return fail(c, MI_ERR_BAD_ARG, "fused_fault_inject must be 0..3"); c->fused_fault_inject = value; return MI_OK; }
    if (!strcmp(name, "fused_timeout_us")) { c->fused_timeout_us = std::max(0, value); return MI_OK; }
    if (!strcmp(name, "hip_fail_after")) { c->hip_fail_after = std::max(0, value); return MI_OK; }
#endif
    return fail(c, MI_ERR_BAD_ARG, "unknown option");
}

// Waits for `stream`.  A bounded-wait expiry of the fused kernel is NOT an error: the finish kernel that follows every fused
// launch has already redone the affected tickets on the device (see kernels/equalize_fused.hip.h) and the event is only counted
// (mi_ctx_get_stat).  MI_ERR_HIP is returned for the one case the repair refuses: a frame whose stamps contradict the protocol.
mi_status mi_ctx_synchronize(mi_ctx* c, void* stream)
{
    ENTER(c);
    if (mi_status st0 = ensure_stream(c)) return st0;
    hipStream_t s = pick_stream(c, stream);
    HIPCHK(c, hipStreamSynchronize(s));
    if (!c->d_fused) return MI_OK;
    uint64_t st4[4];
    mi_status st = fused_read_stats(c, s, st4);
    if (st) return st;
    if (st4[2] > c->fused_seen_hard) {
        c->fused_seen_hard = st4[2];
        c->last_hip = 0;
        return fail(c, MI_ERR_HIP, "fused equalize kernel: a frame could not be repaired after an expired inter-workgroup wait; output invalid");
    }
    return MI_OK;
}

// Statistics: "fused_fallbacks" (fused launches in which a bounded wait expired and the finish kernel redid the missing tickets),
// "fused_frames_repaired", "fused_hard_errors" (frames the repair refused), "fused_last_status" (1 = wait on a frame's LUT
// flag / histogram total, 2 = LUT checksum) -- these read device words with a blocking copy on the context's stream: call after the
// stream the work ran on has been synchronised.  Host-side counters: "fused_demotions" (times the context gave the fused path up
// for a while after repeated repairs), "fused_demoted" (1 while it is given up), "error_drains" (error exits that had to wait for
// a stream before returning), "host_copies_shared" (staging copies the helper thread took half of), "clahe16_mid_launches" (16-bit
// CLAHE calls that launched clahe_interp16_mid_kernel: by the pinned-memory hint, or always / never by option "clahe16_wide").
mi_status mi_ctx_get_stat(mi_ctx* c, const char* name, uint64_t* out)
{
    ENTER(c);
    if (!name || !out) return fail(c, MI_ERR_BAD_ARG, "null stat name / out");
    if (!strcmp(name, "fused_demotions")) { *out = c->fused_demotions; return MI_OK; }
    if (!strcmp(name, "fused_demoted")) { *out = c->fused_demoted ? 1 : 0; return MI_OK; }
    if (!strcmp(name, "error_drains")) { *out = c->error_drains; return MI_OK; }
    if (!strcmp(name, "host_planes_staged")) { *out = c->planes_staged; return MI_OK; }
    if (!strcmp(name, "host_planes_direct")) { *out = c->planes_direct; return MI_OK; }
    if (!strcmp(name, "host_copies_shared")) { *out = c->crew ? c->crew->shared_jobs() : 0; return MI_OK; }
    if (!strcmp(name, "clahe16_mid_launches")) { *out = c->c16_mid_launches; return MI_OK; }
    static const char* names[4] = {"fused_fallbacks", "fused_frames_repaired", "fused_hard_errors", "fused_last_status"};
    for (int k = 0; k < 4; ++k)
        if (!strcmp(name, names[k])) {
            uint64_t st4[4];
            mi_status st = ensure_stream(c);
            if (st) return st;
            if ((st = fused_read_stats(c, c->stream, st4))) return st;
            *out = st4[k];
            return MI_OK;
        }
    return fail(c, MI_ERR_BAD_ARG, "unknown stat");
}

// ---- device-resident batched forms ------------------------------------------------------------------
mi_status mi_equalize_hist_u8_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, void* stream)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{(const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride, width, height, n_frames};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    return equalize_dev(c, pick_stream(c, stream), a, nullptr);
}

mi_status mi_equalize_hist_nv12_batch_dev(mi_ctx* c, const void* d_in, void* d_out, int width, int height, int n_frames,
                                          mi_uv_mode uv_mode, void* stream)
{
    ENTER_COMPUTE(c);
    if (uv_mode != MI_UV_FILL128 && uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    const size_t frame = (size_t)width * height + ((size_t)width * height) / 2;
    PlaneArgs a{(const uint8_t*)d_in, (size_t)width, frame, (uint8_t*)d_out, (size_t)width, frame, width, height, n_frames};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    UVJob uv = nv12_uv((const uint8_t*)d_in, (uint8_t*)d_out, width, height, uv_mode);
    return equalize_dev(c, pick_stream(c, stream), a, &uv);
}

mi_status mi_clahe_u8_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                int width, int height, int n_frames, double clip_limit, int tiles_x, int tiles_y, void* stream)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{(const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride, width, height, n_frames};
    mi_status st = check_plane(c, a, true);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    return clahe_dev(c, pick_stream(c, stream), a, clip_limit, tiles_x, tiles_y, nullptr);
}

mi_status mi_clahe_nv12_batch_dev(mi_ctx* c, const void* d_in, void* d_out, int width, int height, int n_frames,
                                  mi_uv_mode uv_mode, double clip_limit, int tiles_x, int tiles_y, void* stream)
{
    ENTER_COMPUTE(c);
    if (uv_mode != MI_UV_FILL128 && uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    const size_t frame = (size_t)width * height + ((size_t)width * height) / 2;
    PlaneArgs a{(const uint8_t*)d_in, (size_t)width, frame, (uint8_t*)d_out, (size_t)width, frame, width, height, n_frames};
    mi_status st = check_plane(c, a, true);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    UVJob uv = nv12_uv((const uint8_t*)d_in, (uint8_t*)d_out, width, height, uv_mode);
    return clahe_dev(c, pick_stream(c, stream), a, clip_limit, tiles_x, tiles_y, &uv);
}

// ---- stage-level forms ---------------------------------------------------------------------------------
mi_status mi_hist_u8_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                               int width, int height, int n_frames, void* d_hist, void* stream)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{(const uint8_t*)d_src, src_step, src_frame_stride, nullptr, 0, 0, width, height, n_frames};
    mi_status st = check_plane(c, a, false);
    if (st) return st;
    if (!d_hist) return fail(c, MI_ERR_BAD_ARG, "null d_hist");
    hipStream_t s = pick_stream(c, stream);
    if (n_frames == 0) return MI_OK;
    if (width == 0 || height == 0) { HIPCHK(c, hipMemsetAsync(d_hist, 0, (size_t)n_frames * 1024, s)); return MI_OK; }
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        int nparts = 0;
        st = launch_hist_partials(c, s, a, f0, nf, &nparts);
        if (st) return st;
        LAUNCH(c, s, MI_K_EQ_LUT, equalize_lut_kernel, dim3(nf), dim3(kThreads), 0,
               (const uint32_t*)c->d_partial, nparts, 0, (uint8_t*)nullptr, (int32_t*)d_hist + (size_t)f0 * 256);
    }
    return MI_OK;
}

mi_status mi_equalize_lut_batch_dev(mi_ctx* c, const void* d_hist, int64_t total, int n_frames, void* d_lut, void* stream)
{
    ENTER_COMPUTE(c);
    if (!d_hist || !d_lut || n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "null pointer / negative count");
    if (total <= 0 || total > 0x7fffffffLL) return fail(c, MI_ERR_BAD_ARG, "total must be in [1, 2^31)");
    hipStream_t s = pick_stream(c, stream);
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        LAUNCH(c, s, MI_K_EQ_LUT, equalize_lut_kernel, dim3(nf), dim3(kThreads), 0,
               (const uint32_t*)d_hist + (size_t)f0 * 256, 1, (int)total, (uint8_t*)d_lut + (size_t)f0 * 256, (int32_t*)nullptr);
    }
    return MI_OK;
}

mi_status mi_lut_apply_u8_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                    void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                    int width, int height, int n_frames, const void* d_lut, void* stream)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{(const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride, width, height, n_frames};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    if (!d_lut) return fail(c, MI_ERR_BAD_ARG, "null d_lut");
    hipStream_t s = pick_stream(c, stream);
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        st = launch_apply(c, s, a, f0, nf, (const uint8_t*)d_lut + (size_t)f0 * 256, nullptr);
        if (st) return st;
    }
    return MI_OK;
}

mi_status mi_clahe_tile_luts_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                       int width, int height, int n_frames, double clip_limit, int tiles_x, int tiles_y,
                                       void* d_luts, void* stream)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{(const uint8_t*)d_src, src_step, src_frame_stride, nullptr, 0, 0, width, height, n_frames};
    mi_status st = check_plane(c, a, false);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    if (!d_luts) return fail(c, MI_ERR_BAD_ARG, "null d_luts");
    ClaheGeom g;
    st = clahe_geometry(c, width, height, clip_limit, tiles_x, tiles_y, &g);
    if (st) return st;
    hipStream_t s = pick_stream(c, stream);
    const size_t per_frame = (size_t)tiles_x * tiles_y * 256;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        st = launch_tile_luts(c, s, a, g, f0, nf, (uint8_t*)d_luts + (size_t)f0 * per_frame);
        if (st) return st;
    }
    return MI_OK;
}

// ---- host-pointer forms (the cv::Mat boundary) -----------------------------------------------------------
// Host plane -> (pinned staging unless the plane is pinned ->) H2D -> kernels -> D2H (-> pinned -> host rows), all on the
// context's stream, synchronous on return.  `nv12_mode` < 0: plain Y plane; otherwise whole NV12 frame.
//
// Host memory.  Pinned planes (mi_host_register, or pinned by the caller) are DMA'd as they are.  Everything else is packed
// through the context's own pinned staging buffers in chunks whose host copies overlap the DMA of the chunk before; the copies
// are shared with the context's helper thread (copy_crew.hpp), because one core copies slower than the link transfers.  The
// library never hands PAGEABLE memory to hipMemcpyAsync: the runtime pins such pages on the fly and keeps them pinned in a cache
// of its own after the call has returned (profiles/r02_v_pageable_path_log.txt), and both process aborts on record happened on
// that path (docs/experiments.md, "the silent abort").  The option that used to bring that path back (host_direct) is gone.
//
// Error exits.  From the first copy that touches caller memory on, every return that has not itself waited for the stream
// drains it first (StreamDrain): the caller may free or reuse src / dst as soon as the call returns, whatever it returns.
struct CrewCall {
    mi_host::CopyCrew* crew;
    CrewCall(mi_ctx* c, bool want) : crew(nullptr)
    {
        if (!want || c->host_copy_threads < 2) return;
        if (!c->crew) c->crew = new (std::nothrow) mi_host::CopyCrew();
        if (c->crew) { crew = c->crew; crew->begin(); }
    }
    ~CrewCall() { if (crew) crew->end(); }
    void copy(uint8_t* dst, size_t dstep, const uint8_t* src, size_t sstep, int width, int rows)
    {
        if (crew) crew->copy_rows(dst, dstep, src, sstep, (size_t)width, (size_t)rows);
        else copy_rows(dst, dstep, src, sstep, width, rows);
    }
};

static mi_status host_op(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                         int width, int height, int nv12_mode, bool is_clahe, double clip_limit, int tiles_x, int tiles_y)
{
    // Only the Y plane crosses PCIe.  The UV half of a host NV12 frame is not computed on: it is filled with 128 or
    // copied on the host (exactly the reference's memset/memcpy, OpenCVequalHist.cpp:160-162 / ColoropenCVCwqualHist.cpp:165)
    // while the GPU works on Y -- a third fewer bytes over the bus than shipping UV both ways.  (The device-resident
    // forms do the UV fill/copy on the GPU, fused into their launches.)
    const size_t ybytes = (size_t)width * height;
    const size_t uvbytes = nv12_mode >= 0 ? ybytes / 2 : 0;
    mi_status st;
    if ((st = grow_dev(c, &c->d_stage_in, &c->stage_in_bytes, ybytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, ybytes))) return st;
    if (!c->stream_b && c->host_copy_streams > 1) {
        // the second copy stream exists only in contexts that run host forms on unpinned planes (a streaming worker's context never
        // does: every stream a process creates competes for the runtime's few hardware queues, see pipe_stream_create)
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_k, hipEventDisableTiming));
    }
    hipStream_t s = c->stream, sb = c->stream_b ? c->stream_b : c->stream;
    const bool in_pinned = src_step == (size_t)width && host_range_pinned(src, ybytes, &c->pin_neg);
    const bool out_pinned = dst_step == (size_t)width && host_range_pinned(dst, ybytes, &c->pin_neg);
    if (!in_pinned && (st = grow_pinned(c, &c->h_pin_in, &c->pin_in_bytes, ybytes))) return st;
    if (!out_pinned && (st = grow_pinned(c, &c->h_pin_out, &c->pin_out_bytes, ybytes))) return st;
    for (bool direct : {in_pinned, out_pinned}) ++(direct ? c->planes_direct : c->planes_staged);
    CrewCall crew(c, !(in_pinned && out_pinned) && ybytes >= 4 * mi_host::CopyCrew::kMinBytes);
    StreamDrain drain(HipStreamSync{}, drain_counter(c));
    drain.watch(s); drain.watch(sb);
    // Staged planes move in chunks whose host copies overlap the DMA of the chunks before.  The chunks ALTERNATE between two
    // streams: the copy engine idles ~10-13 us between two dependent copies of one stream (measured: seven chunks took 268 us
    // instead of 157), and the other stream's transfer covers that gap.  Chunk sizes ramp 256 KiB -> 2 MiB on the way up (the
    // engine starts after a short first host copy) and back down at the end (the last host copy, which nothing overlaps, is a
    // short one): 256 KiB, 512 KiB, then 1 MiB chunks -- the best of a 4 x 5 sweep of chunk size x ramp on a 4K plane
    // (profiles/r03_b_host_chunk_sweep.txt: 0.406 ms per call against 0.464 with 2 MiB chunks ramped from 256 KiB).
    static const int chunk_kb = [] { const char* e = getenv("MI_LUMAEQ_HOST_CHUNK_KB"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 1024; }();
    static const int ramp0 = [] { const char* e = getenv("MI_LUMAEQ_HOST_CHUNK_RAMP"); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 4; }();
    const int rows_per_chunk = std::max(1, (int)(((size_t)chunk_kb << 10) / (size_t)width));
    const bool two = c->host_copy_streams > 1 && c->stream_b && c->ev_b && c->ev_k;
    if (in_pinned) {
        HIPCHK(c, hipMemcpyAsync(c->d_stage_in, src, ybytes, hipMemcpyHostToDevice, s));
    } else {
        int y0 = 0, i = 0;
        bool used_b = false;
        for (int ramp = ramp0; y0 < height; ramp = std::max(1, ramp / 2), ++i) {
            const int nr = std::min(std::max(1, rows_per_chunk / ramp), height - y0);
            const size_t off = (size_t)y0 * width;
            crew.copy(c->h_pin_in + off, (size_t)width, src + (size_t)y0 * src_step, src_step, width, nr);
            const bool on_b = two && (i & 1);
            HIPCHK(c, hipMemcpyAsync(c->d_stage_in + off, c->h_pin_in + off, (size_t)nr * width, hipMemcpyHostToDevice, on_b ? sb : s));
            used_b |= on_b;
            y0 += nr;
        }
        if (used_b) {                                            // the kernels (on s) need stream_b's chunks as well
            HIPCHK(c, hipEventRecord(c->ev_b, sb));
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_b, 0));
        }
    }
    PlaneArgs a{c->d_stage_in, (size_t)width, ybytes, c->d_stage_out, (size_t)width, ybytes, width, height, 1};
    st = is_clahe ? clahe_dev(c, s, a, clip_limit, tiles_x, tiles_y, nullptr) : equalize_dev(c, s, a, nullptr);
    if (st) return st;
    // "unrecoverable frame" counter of the fused path: the finish kernel mirrors it into pinned host memory (FusedJob::host_hard),
    // so it is simply read once the downloads -- which follow the finish kernel in stream order -- have completed: no extra copy
    const bool check_status = !is_clahe && c->d_fused;
    auto hard_error = [&]() {
        if (!check_status) return false;
        const uint64_t hard = fused_hard_seen(c);
        if (hard <= c->fused_seen_hard) return false;
        c->fused_seen_hard = hard;
        return true;
    };
    auto host_uv = [&]() {                                      // runs while the GPU / DMA engines are busy with Y
        if (!uvbytes) return;
        if (nv12_mode == MI_UV_FILL128) memset(dst + ybytes, 128, uvbytes);
        else if (dst != src) memmove(dst + ybytes, src + ybytes, uvbytes);
    };
    static const char* kHardMsg = "fused equalize kernel: a frame could not be repaired after an expired inter-workgroup wait; output invalid";
    if (out_pinned) {
        HIPCHK(c, hipMemcpyAsync(dst, c->d_stage_out, ybytes, hipMemcpyDeviceToHost, s));
        host_uv();                                              // the copy is asynchronous: the UV work overlaps the DMA itself
        HIPCHK(c, hipStreamSynchronize(s));
        if (!in_pinned && two) HIPCHK(c, hipStreamSynchronize(sb));   // (idle by now: the kernels waited for its last upload)
        drain.done();
        if (hard_error()) return fail(c, MI_ERR_HIP, kHardMsg);
        return MI_OK;
    }
    // device -> pinned in chunks, each followed by an event; then drain chunk by chunk into the caller's rows
    struct Chunk { size_t off, bytes; int y0, nr; };
    std::vector<Chunk> chunks;
    {
        std::vector<int> sizes;
        int left = height;
        for (int ramp = ramp0; left > 0; ramp = std::max(1, ramp / 2)) { const int nr = std::min(std::max(1, rows_per_chunk / ramp), left); sizes.push_back(nr); left -= nr; }
        int y0 = 0;
        for (size_t i = sizes.size(); i-- > 0;) { chunks.push_back({(size_t)y0 * width, (size_t)sizes[i] * width, y0, sizes[i]}); y0 += sizes[i]; }
    }
    while (c->chunk_events.size() < chunks.size()) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->chunk_events.push_back(e);
    }
    if (two && chunks.size() > 1) {                              // stream_b's downloads wait for the kernels
        HIPCHK(c, hipEventRecord(c->ev_k, s));
        HIPCHK(c, hipStreamWaitEvent(sb, c->ev_k, 0));
    }
    for (size_t i = 0; i < chunks.size(); ++i) {
        hipStream_t cs = (two && (i & 1)) ? sb : s;
        HIPCHK(c, hipMemcpyAsync(c->h_pin_out + chunks[i].off, c->d_stage_out + chunks[i].off, chunks[i].bytes, hipMemcpyDeviceToHost, cs));
        HIPCHK(c, hipEventRecord(c->chunk_events[i], cs));
    }
    host_uv();
    for (size_t i = 0; i < chunks.size(); ++i) {
        HIPCHK(c, hipEventSynchronize(c->chunk_events[i]));
        if (i == 0 && hard_error()) return fail(c, MI_ERR_HIP, kHardMsg);     // (the guard drains the remaining chunks)
        crew.copy(dst + (size_t)chunks[i].y0 * dst_step, dst_step, c->h_pin_out + chunks[i].off, (size_t)width, width, chunks[i].nr);
    }
    drain.done();                                               // every chunk's event has been waited for: both streams are idle
    return MI_OK;
}

mi_status mi_equalize_hist_u8(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0) return st;
    return host_op(c, src, src_step, dst, dst_step, width, height, -1, false, 0.0, 0, 0);
}

mi_status mi_clahe_u8(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height,
                      double clip_limit, int tiles_x, int tiles_y)
{
    ENTER_COMPUTE(c);
    PlaneArgs a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0) return MI_OK;
    return host_op(c, src, src_step, dst, dst_step, width, height, -1, true, clip_limit, tiles_x, tiles_y);
}

mi_status mi_equalize_hist_nv12(mi_ctx* c, const uint8_t* in, uint8_t* out, int width, int height, mi_uv_mode uv_mode)
{
    ENTER_COMPUTE(c);
    if (uv_mode != MI_UV_FILL128 && uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    PlaneArgs a{in, (size_t)std::max(width, 0), 0, out, (size_t)std::max(width, 0), 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0) return st;
    return host_op(c, in, (size_t)width, out, (size_t)width, width, height, (int)uv_mode, false, 0.0, 0, 0);
}

mi_status mi_clahe_nv12(mi_ctx* c, const uint8_t* in, uint8_t* out, int width, int height, mi_uv_mode uv_mode,
                        double clip_limit, int tiles_x, int tiles_y)
{
    ENTER_COMPUTE(c);
    if (uv_mode != MI_UV_FILL128 && uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    PlaneArgs a{in, (size_t)std::max(width, 0), 0, out, (size_t)std::max(width, 0), 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0) return MI_OK;
    return host_op(c, in, (size_t)width, out, (size_t)width, width, height, (int)uv_mode, true, clip_limit, tiles_x, tiles_y);
}

}  // extern "C"
