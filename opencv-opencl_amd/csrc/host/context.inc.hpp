// context.inc.hpp -- mi_ctx, error helpers, scratch growth, kernel launch bracket, grid heuristics, stage launchers (K1-K3)
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

namespace {

struct PendingEvent { hipEvent_t a, b; int kernel; };

}  // namespace

// process-wide registry of caller-pinned host ranges (mi_host_register): host/pin_registry.hpp holds the rules and the locking
static mi_host::PinRegistry g_pins;
// caller memory that a pipe's queued DMA still reads or writes (pending_ranges.hpp): mi_host_unregister answers MI_ERR_BUSY for it
static mi_host::PendingRanges g_pending_dma;
using PinnedNegCache = mi_host::PinnedNegCache;

// What the HIP runtime says about memory the caller pinned itself (hipHostMalloc / hipHostRegister, a pinned torch tensor): both ends
// of the range are host-pinned AND lie in ONE allocation.  Called by the registry WITHOUT its lock.
static bool runtime_says_pinned(const void* p, size_t bytes)
{
    // An unknown pointer is not an error worth keeping: only the error THIS query raised is cleared (a pending error of an earlier
    // asynchronous call is left for its own check).
    auto quiet = [](hipError_t before, hipError_t e) { if (e != hipSuccess && before == hipSuccess) (void)hipGetLastError(); return e == hipSuccess; };
    auto pinned_at = [&](const void* q) {
        hipPointerAttribute_t at{};
        const hipError_t before = hipPeekAtLastError();
        return quiet(before, hipPointerGetAttributes(&at, q)) && at.type == hipMemoryTypeHost;
    };
    // ... and both ends must lie in ONE allocation: two pinned buffers with pageable memory between them, or a registration that covers
    // only part of the plane, would otherwise be DMA'd as they are -- the pageable path this library does not take.  An allocation the
    // runtime cannot describe is treated as unpinned (the plane is staged: slower, never wrong).
    auto allocation_of = [&](const void* q, void** base, size_t* size) {
        const hipError_t before = hipPeekAtLastError();
        return quiet(before, hipMemGetAddressRange((hipDeviceptr_t*)base, size, (hipDeviceptr_t)const_cast<void*>(q)));
    };
    if (!(pinned_at(p) && pinned_at((const uint8_t*)p + bytes - 1))) return false;
    void *b0 = nullptr, *b1 = nullptr;
    size_t s0 = 0, s1 = 0;
    return allocation_of(p, &b0, &s0) && allocation_of((const uint8_t*)p + bytes - 1, &b1, &s1) && b0 == b1 && s0 == s1 && s0 >= bytes;
}

static bool host_range_pinned(const void* p, size_t bytes, PinnedNegCache* neg = nullptr)
{
    return g_pins.pinned(p, bytes, neg, runtime_says_pinned);
}

// Concurrency guard for the fused kernel.  Its workgroups wait for each other, so every slice of a frame (T
// workgroups) must be co-resident.  Several contexts may run fused launches on one GPU at the same time; each launch is then only
// guaranteed a share of the chip.  At most kMaxFusedCtxPerDevice live contexts per device get the fused path (later ones use the
// three-kernel path), and a frame is only fused when T <= (CUs * WGs/CU) / (2 * kMaxFusedCtxPerDevice), i.e. a launch that
// receives half of its fair share still has all of a frame's slices resident.  What this cannot see -- other processes on the
// GPU, a preempted queue -- is covered by the bounded waits and the finish kernel's repair (kernels/equalize_fused.hip.h).
constexpr int kMaxDevices = 64;
constexpr int kMaxFusedCtxPerDevice = 4;
static std::atomic<int> g_fused_ctx_live[kMaxDevices];

struct mi_ctx {
    int device = -1;
    bool fused_slot = false;                                     // this context holds one of the per-device fused slots
    hipStream_t stream = nullptr;
    hipStream_t stream_b = nullptr;                              // second copy stream of the host forms: chunk DMAs alternate between the two, so the
                                                                 // ~10 us the copy engine idles between two dependent copies of ONE stream is covered
                                                                 // by the other stream's transfer (profiles/r03_a_host_form_timeline.txt)
    hipEvent_t ev_b = nullptr, ev_k = nullptr;                   // stream_b's uploads done -> kernels; kernels done -> stream_b's downloads
    std::mutex mu;
    int last_hip = 0;
    std::string last_msg = "ok";
    int cu_count = 256;

    // device scratch, grown lazily ("allocate once per size", OpenCLequalHist.cpp:175-186)
    uint32_t* d_partial = nullptr; size_t partial_bytes = 0;     // histogram partials
    uint8_t*  d_luts = nullptr;    size_t luts_bytes = 0;        // per-frame / per-tile LUTs
    uint32_t* d_sync16 = nullptr;  size_t sync16_bytes = 0;      // per-frame arrival words of tile_hist12_kernel (zero between launches)
    uint32_t* d_ghist = nullptr;   size_t ghist_bytes = 0;       // global histograms + arrival counters of hist_lut_kernel (zero between launches)
    int two_kernel_max_frames = 8;                               // option "two_kernel_max_frames": batches up to this size take hist_lut_kernel + lut_apply_kernel
    uint32_t* d_fused = nullptr;   size_t fused_bytes = 0;       // hand-off block of the fused kernel (self-cleaning; all per-launch
                                                                 // state lives in it, so captured launches replay unchanged).  The
                                                                 // ticket stamps are PART of the block: stamps, epochs and checksums
                                                                 // share one lifetime (a retired block keeps its own stamps for the
                                                                 // graphs that still replay into it)
    size_t fused_cap = 0;                                        // frames the block is laid out for
    size_t fused_ticket_cap = 0;                                 // ticket stamps the block holds
    uint32_t fused_generation = 0;                               // blocks this context has allocated so far (selects the host mirror word)
    uint64_t fused_stat_base[4] = {};                            // statistics of hand-off blocks this context has since replaced
    uint64_t fused_seen_hard = 0;                                // unrecoverable frames already reported to the caller
    uint32_t* h_status = nullptr;                                // pinned mirror of the device statistics words (blocking reads)
    // Demotion of the fused path (equalize_fused.inc.hpp): the finish kernel also writes its "launches repaired" counter into a
    // word of pinned host memory, so the host learns about repaired launches without a copy or a synchronisation.
    uint32_t* h_mirror = nullptr;                                // pinned, device-written: [g % 16] repaired launches, [16 + g % 16] unrecoverable
                                                                 // frames of hand-off block generation g
    uint64_t fused_repaired_base = 0, fused_hard_base = 0;       // ... of blocks since replaced
    uint32_t* fused_hard_word = nullptr;                         // set by a pipe around its launch: pinned word of the frame's slot that receives
                                                                 // the block's "frames refused" count if this launch refuses any (instead of the mirror)
    uint64_t fused_window_start_repaired = 0;                    // repaired launches seen when the current observation window began
    uint32_t fused_window_launches = 0;                          // fused launches issued in the window
    uint64_t fused_demotions = 0;                                // statistic "fused_demotions"
    bool fused_demoted = false;                                  // launches take the three-kernel path until fused_reprobe_at
    bool fused_probing = false;                                  // the launch after a demotion period: one repair demotes again
    std::chrono::steady_clock::time_point fused_reprobe_at{};
    int fused_reprobe_ms = 1000, fused_reprobe_ms_now = 1000;    // option "fused_reprobe_ms": first demotion period (doubles, up to 64x)
    int fused_demote_after = 3;                                  // option "fused_demote_after": repaired launches per window that demote (0 = never)
    // pipes on this context (pipe.inc.hpp): one at a time; while frames are pending the other compute entry points answer MI_ERR_BUSY
    int pipes_open = 0, pipe_pending = 0;
    hipEvent_t ev_scratch = nullptr;                             // end of the last device-form call on a caller stream, while a pipe is open
    bool scratch_foreign = false;                                // ... recorded and not yet waited for by the pipe's compute stream
    hipStream_t cur_stream = nullptr;                            // stream of the device-form call in progress (pick_stream) ...
    bool cur_stream_set = false;                                 // ... valid (the null stream is a legitimate value)
    bool fused_pair_open = false;                                // a fused kernel was launched and its finish kernel was not (a failed launch in between):
                                                                 // the hand-off block's counters are in an unknown state and are reset before the next launch
    unsigned long long error_drains = 0;                                // statistic "error_drains": error exits that had to drain a stream first
    unsigned long long planes_staged = 0, planes_direct = 0;            // statistics "host_planes_staged" / "host_planes_direct": host planes packed through
                                                                        // the library's pinned staging / DMA'd as the caller pinned them
    PinnedNegCache pin_neg;
    mi_host::CopyCrew* crew = nullptr;                           // helper thread for staging copies of the host forms (created on first use)
    int pipe_private_streams = 0;                                // MI_LUMAEQ_PIPE_PRIVATE_STREAMS=1 (measurements): a pipe created from now on owns its streams instead of sharing the device's
    int pipe_copy_streams = 2;                                   // option "pipe_copy_streams": copy streams per direction of a pipe created from now on
    int host_copy_streams = 2;                                   // option "host_copy_streams": 1 = every chunk DMA of a host form on the one stream
    int host_copy_threads = 2;                                   // option "host_copy_threads": 1 = the calling thread copies alone
    bool capturing = false;                                      // the stream of the call in progress is being captured (hipGraph)
    bool graph_captured = false;                                 // a capture has been seen: scratch referenced by graph nodes is never freed
    std::vector<void*> retired;                                  // ... it is parked here until the context is destroyed
    int fused_mode = 1;                                          // MI_LUMAEQ_FUSED=0 forces the 3-kernel path
    int fused_wgs_per_cu = 4;                                    // MI_LUMAEQ_FUSED_WGS_PER_CU
    int fused_vpt = kVPT;                                        // MI_LUMAEQ_FUSED_VPT (8, 16, 20, 24)
    int fused_acquire = 1;                                       // MI_LUMAEQ_FUSED_ACQUIRE
#ifdef MI_TEST_HOOKS                                             // libmi_lumaeq_test.so only (csrc/Makefile): the product library has neither
    int fused_fault_inject = 0;                                  // option "fused_fault_inject": 0 off, 1..3 see kernels/equalize_fused.hip.h
    int fused_timeout_us = 0;                                    // option "fused_timeout_us": > 0 overrides fused_timeout_ms
    int hip_fail_after = 0;                                      // option "hip_fail_after": the n-th checked HIP call from now on reports a failure
                                                                 // (after it has been issued), 0 = off
#endif
    int fused_timeout_ms = 50;                                   // option "fused_timeout_ms": bound of every inter-workgroup wait
    int bgr_fused = 1;                                           // option "bgr_fused": 9 B/px two-pass BGR luma equalization / CLAHE
    int clahe_fp_contract = 0;                                   // option "clahe_fp_contract": CLAHE interpolation with GCC's FMA contraction (aarch64 OpenCV builds)
    int clahe16_fast12 = 1;                                      // option "clahe16_fast12": 4096-bin x 8-copy tile histograms with the LUT folded in (12-bit bet)
    int clahe16_transposed = 0;                                  // option "clahe16_transposed": value-major LUTs for 16-bit interpolation (tiles <= 64)
    int clahe16_wide = 1;                                        // option "clahe16_wide": the 16384-entry interpolation kernel for 14-bit rectangles (round 6): 0 never, 1 when such a rectangle was seen lately, 2 always
    uint32_t c16_seq = 1000;                                     // sequence number of 16-bit CLAHE launches (WideHint: the pinned words start at 0 = "long ago")
    uint64_t c16_mid_launches = 0;                               // statistic "clahe16_mid_launches"
    int clahe_hist_threads = 512;                                // option "clahe_hist_threads": 256 or 512 threads per tile-histogram workgroup
    int clahe_tiles_per_wg = 0;                                  // option "clahe_tiles_per_wg": tiles a tile-histogram workgroup walks in batches (0 = by tile size, 1, 2, 4, 8)
    int clahe_seg_pairs = 9;                                     // option "clahe_seg_pairs": pairs per float table when a wide grid is cut into column segments (4..15)
    int clahe_xcd_map = 1;                                       // option "clahe_xcd_map": XCD-aware tile order of the tile histogram pass
    int clahe_float_tables = 1;                                  // option "clahe_float_tables": f32 pair tables in LDS (tiles_x <= 14)
    uint8_t*  d_stage_in = nullptr;  size_t stage_in_bytes = 0;  // device frame for the host-pointer forms
    uint8_t*  d_stage_out = nullptr; size_t stage_out_bytes = 0;
    uint8_t*  d_c16 = nullptr;     size_t c16_bytes = 0;         // 16-bit CLAHE: tile histograms + ushort LUTs (N4)
    uint8_t*  d_planes = nullptr;  size_t planes_bytes = 0;      // Y,U,V,Y' planes of the BGR luma pipeline (N3)
    uint8_t*  h_pin_in = nullptr;  size_t pin_in_bytes = 0;      // pinned staging
    uint8_t*  h_pin_out = nullptr; size_t pin_out_bytes = 0;

    // profiling
    int profiling = 0;                                           // 0 off, 1 every kernel, 2 every kernel but the housekeeping ones
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> free_events;
    std::vector<hipEvent_t> chunk_events;                        // D2H chunk completion (host-pointer forms)
    mi_profile prof{};
    std::vector<float> samples[MI_K_COUNT];                      // per-launch durations since the last reset (ring of 65 536)
    size_t sample_pos[MI_K_COUNT] = {};
};

namespace {

mi_status fail_hip(mi_ctx* c, hipError_t e, const char* what)
{
    c->last_hip = (int)e;
    c->last_msg = std::string(what) + ": " + hipGetErrorString(e);
    return MI_ERR_HIP;
}
mi_status fail(mi_ctx* c, mi_status s, const char* msg)
{
    if (c) c->last_msg = msg;
    return s;
}

// Test hook (libmi_lumaeq_test.so only): option "hip_fail_after" = n makes the n-th checked HIP call from now on REPORT a failure
// after it has been issued -- the copy or launch is really in flight, which is exactly the state the error paths must clean up.
#ifdef MI_TEST_HOOKS
inline hipError_t test_hook_result(mi_ctx* c, hipError_t e)
{
    if (c && c->hip_fail_after > 0 && --c->hip_fail_after == 0 && e == hipSuccess) return hipErrorUnknown;
    return e;
}
#define MI_HOOKED(c, e) test_hook_result((c), (e))
#else
#define MI_HOOKED(c, e) (e)
#endif

#define HIPCHK(c, expr)                                         \
    do {                                                        \
        hipError_t e__ = MI_HOOKED((c), (expr));                \
        if (e__ != hipSuccess) return fail_hip((c), e__, #expr); \
    } while (0)

// "Never return while a DMA on caller memory is in flight" (drain_guard.hpp): error exits synchronise the watched streams.
struct HipStreamSync {
    void operator()(void* s) const { if (s) (void)hipStreamSynchronize((hipStream_t)s); else (void)hipDeviceSynchronize(); }
};
using StreamDrain = mi_host::DrainOnExit<HipStreamSync>;
inline unsigned long long* drain_counter(mi_ctx* c) { return &c->error_drains; }

template <class T>
mi_status grow_dev(mi_ctx* c, T** p, size_t* have, size_t need)
{
    if (need <= *have) return MI_OK;
    // allocations are not capturable, and kernel nodes of an earlier capture keep pointing at the scratch they were recorded with
    if (c->capturing) return fail(c, MI_ERR_UNSUPPORTED, "device scratch must grow inside a stream capture: size it with one eager call of this shape first");
    if (*p) {
        if (c->graph_captured) c->retired.push_back(*p);         // a captured graph may still replay into it
        else { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(*p)); }   // rare: scratch may be in use on a caller stream
        *p = nullptr; *have = 0;
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, need);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(c, MI_ERR_OOM, "device allocation failed"); }
    if (e != hipSuccess) return fail_hip(c, e, "hipMalloc");
    *p = (T*)q; *have = need;
    return MI_OK;
}

mi_status grow_pinned(mi_ctx* c, uint8_t** p, size_t* have, size_t need)
{
    if (need <= *have) return MI_OK;
    if (c->capturing) return fail(c, MI_ERR_UNSUPPORTED, "pinned staging must grow inside a stream capture");
    if (*p) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipHostFree(*p)); *p = nullptr; *have = 0; }
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, need, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(c, MI_ERR_OOM, "pinned allocation failed"); }
    if (e != hipSuccess) return fail_hip(c, e, "hipHostMalloc");
    *p = (uint8_t*)q; *have = need;
    return MI_OK;
}

// ---- kernel launch with optional timing ------------------------------------------------------------
// With profiling on, the kernel is dispatched through hipExtLaunchKernelGGL, which stamps the start and the stop event from the
// dispatch itself (the way CL_QUEUE_PROFILING_ENABLE stamps a kernel's own start / end, 1frameMeasure.cpp:81-85): no marker
// packets on the stream.  Two hipEventRecord calls around every launch -- round 1 -- cost the timed region of the bench 4 %.
struct Bracket {
    mi_ctx* c; int kernel; hipEvent_t a = nullptr, b = nullptr; bool on;
    Bracket(mi_ctx* c_, int k) : c(c_), kernel(k), on(!c_->capturing && (c_->profiling == 1 || (c_->profiling == 2 && k != MI_K_FUSED_FINISH))) {}
    Bracket(const Bracket&) = delete;
    Bracket& operator=(const Bracket&) = delete;
    hipError_t acquire()
    {
        if (!on) return hipSuccess;
        for (hipEvent_t* e : {&a, &b}) {
            if (!c->free_events.empty()) { *e = c->free_events.back(); c->free_events.pop_back(); }
            else { hipError_t r = hipEventCreate(e); if (r != hipSuccess) return r; }
        }
        return hipSuccess;
    }
    void submitted() { if (on) { c->pending.push_back({a, b, kernel}); a = b = nullptr; } }
    // a launch that failed after acquire(): the events go back to the pool instead of leaking
    ~Bracket() { for (hipEvent_t e : {a, b}) if (e) c->free_events.push_back(e); }
};

#define LAUNCH(c, s, kid, kern, grid, block, shmem, ...)                                              \
    do {                                                                                              \
        Bracket br__((c), (kid));                                                                     \
        HIPCHK((c), br__.acquire());                                                                  \
        if (br__.on) hipExtLaunchKernelGGL(kern, grid, block, shmem, (s), br__.a, br__.b, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, shmem, (s), __VA_ARGS__);                          \
        HIPCHK((c), hipGetLastError());                                                               \
        br__.submitted();                                                                             \
    } while (0)

// ---- geometry / grid heuristics -----------------------------------------------------------------
// Memory-bound kernels: aim for ~8 workgroups per CU in total, never less than 16 KiB per workgroup
// (a workgroup pays 64 LDS wave-ops to zero and fold its replicated histogram / LUT).
int blocks_per_frame(const mi_ctx* c, long long bytes_per_frame, int rows, int n_frames, int cap)
{
    const long long target = (long long)c->cu_count * 8;
    long long b = (target + n_frames - 1) / n_frames;
    const long long by_bytes = std::max<long long>(1, bytes_per_frame / 16384);
    b = std::min(b, by_bytes);
    if (rows > 1) b = std::min<long long>(b, rows);
    b = std::min<long long>(b, cap);
    return (int)std::max<long long>(1, b);
}

struct PlaneArgs {
    const uint8_t* src; size_t src_step, src_frame;
    uint8_t* dst; size_t dst_step, dst_frame;
    int width, height, n_frames;
};

mi_status check_plane(mi_ctx* c, const PlaneArgs& a, bool need_dst)
{
    if (!c) return MI_ERR_BAD_ARG;
    if (a.width < 0 || a.height < 0 || a.n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "negative size");
    if (a.width == 0 || a.height == 0 || a.n_frames == 0) return MI_OK;
    if (!a.src || (need_dst && !a.dst)) return fail(c, MI_ERR_BAD_ARG, "null plane pointer");
    if (a.src_step < (size_t)a.width || (need_dst && a.dst_step < (size_t)a.width)) return fail(c, MI_ERR_BAD_ARG, "step < width");
    if ((long long)a.width * a.height > 0x7fffffffLL) return fail(c, MI_ERR_UNSUPPORTED, "width*height must be < 2^31 (OpenCV: int total)");
    if (a.width > (1 << 24) || a.height > (1 << 24)) return fail(c, MI_ERR_UNSUPPORTED, "width/height must be <= 2^24");
    return MI_OK;
}

PlaneBatch make_plane(const PlaneArgs& a)
{
    PlaneBatch p;
    p.src = a.src; p.dst = a.dst;
    p.src_frame = (long long)a.src_frame; p.dst_frame = (long long)a.dst_frame;
    const bool contiguous = a.src_step == (size_t)a.width && (!a.dst || a.dst_step == (size_t)a.width);
    if (contiguous || a.height == 1) {
        p.rows = 1; p.row_bytes = (long long)a.width * a.height;
        p.src_step = p.row_bytes; p.dst_step = p.row_bytes;
    } else {
        p.rows = a.height; p.row_bytes = a.width;
        p.src_step = (long long)a.src_step; p.dst_step = (long long)a.dst_step;
    }
    return p;
}

constexpr int kMaxGridY = 65535;

// ---- stage launchers (all assume ctx lock held, device set) -------------------------------------
mi_status launch_hist_partials(mi_ctx* c, hipStream_t s, const PlaneArgs& a, int f0, int nf, int* nparts_out)
{
    PlaneArgs b = a;
    b.src = a.src + (size_t)f0 * a.src_frame; b.dst = nullptr; b.n_frames = nf;
    PlaneBatch p = make_plane(b);
    const int B = blocks_per_frame(c, (long long)a.width * a.height, p.rows, nf, 256);
    mi_status st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * B * 256 * sizeof(uint32_t));
    if (st) return st;
    LAUNCH(c, s, MI_K_HIST, hist_partial_kernel, dim3(B, nf), dim3(kHistThreads), 0, p, c->d_partial);
    *nparts_out = B;
    return MI_OK;
}

mi_status launch_apply(mi_ctx* c, hipStream_t s, const PlaneArgs& a, int f0, int nf, const uint8_t* d_luts, const UVJob* uv_all)
{
    PlaneArgs b = a;
    b.src = a.src + (size_t)f0 * a.src_frame; b.dst = a.dst + (size_t)f0 * a.dst_frame; b.n_frames = nf;
    PlaneBatch p = make_plane(b);
    UVJob uv{};
    long long bytes = (long long)a.width * a.height * 2;
    if (uv_all && uv_all->bytes > 0) {
        uv = *uv_all;
        uv.src = uv_all->src ? uv_all->src + (long long)f0 * uv_all->src_frame : nullptr;
        uv.dst = uv_all->dst + (long long)f0 * uv_all->dst_frame;
        bytes += uv.bytes * (uv.mode ? 2 : 1);
    }
    const int B = blocks_per_frame(c, bytes / 2, p.rows, nf, 2048);
    LAUNCH(c, s, MI_K_LUT_APPLY, lut_apply_kernel, dim3(B, nf), dim3(kThreads), 0, p, d_luts, uv);
    return MI_OK;
}
