// mi_lumaeq.hip -- C ABI (include/mi_lumaeq.h) over the gfx950 kernels in lumaeq_kernels.hip.h.
//
// Boundary being replaced (reference file:line):
//   cv::equalizeHist call site            OpenCVequalHist.cpp:145, nextimprovement.cpp:168
//   cv::CLAHE::apply call site            clahevideo.cpp:195, clahe1frame.cpp:93
//   FPGA backend host sequence            OpenCLequalHist.cpp:346-365 (setArg x5, write x2, task, read)
//   per-worker device objects + buffers   OpenCLequalHist.cpp:142-152, :175-186
// There is NO CPU fallback in this file: without a HIP device every entry point fails loudly.
#include "../../include/mi_lumaeq.h"
#include "lumaeq_kernels.hip.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace mi;

namespace {

struct PendingEvent { hipEvent_t a, b; int kernel; };

}  // namespace

// Concurrency guard for the fused kernel.  Its workgroups wait for each other, so every slice of a frame (T
// workgroups) must be co-resident.  Several contexts may run fused launches on one GPU at the same time (the
// worker pool does); each launch is then only guaranteed a share of the chip.  At most kMaxFusedCtxPerDevice live
// contexts per device get the fused path (later ones use the three-kernel path), and a frame is only fused when
// T <= (CUs * WGs/CU) / (2 * kMaxFusedCtxPerDevice), i.e. a launch that receives half of its fair share still
// has all of a frame's slices resident.  (Other processes on the GPU are covered by the bounded waits.)
// process-wide registry of caller-pinned host ranges (mi_host_register)
struct PinnedRange { uintptr_t lo, hi; };
static std::mutex g_pin_mu;
static std::vector<PinnedRange> g_pinned;
static bool host_range_pinned(const void* p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const auto& r : g_pinned) if (lo >= r.lo && hi <= r.hi) return true;
    return false;
}

constexpr int kMaxDevices = 64;
constexpr int kMaxFusedCtxPerDevice = 4;
static std::atomic<int> g_fused_ctx_live[kMaxDevices];

struct mi_ctx {
    int device = -1;
    bool fused_slot = false;                                     // this context holds one of the per-device fused slots
    hipStream_t stream = nullptr;
    std::mutex mu;
    int last_hip = 0;
    std::string last_msg = "ok";
    int cu_count = 256;

    // device scratch, grown lazily ("allocate once per size", OpenCLequalHist.cpp:175-186)
    uint32_t* d_partial = nullptr; size_t partial_bytes = 0;     // histogram partials
    uint8_t*  d_luts = nullptr;    size_t luts_bytes = 0;        // per-frame / per-tile LUTs
    uint32_t* d_fused = nullptr;   size_t fused_bytes = 0;       // hand-off block of the fused kernel (self-cleaning)
    size_t fused_cap = 0;                                        // frames the block is laid out for
    unsigned long long fused_work_base = 0;                      // value of the device ticket counter at the next launch
    uint32_t fused_epoch = 0;
    bool fused_dirty = true;                                     // block must be zeroed before the next launch
    bool fused_capture_safe = false;                             // set once a call was seen inside a stream capture (hipGraph):
                                                                 // from then on every launch zeroes the block itself and uses
                                                                 // constant epoch / ticket base, so a captured graph can be replayed
    uint32_t* h_status = nullptr;                                // pinned mirror of the device status word
    int fused_mode = 1;                                          // MI_LUMAEQ_FUSED=0 forces the 3-kernel path
    int fused_wgs_per_cu = 4;                                    // MI_LUMAEQ_FUSED_WGS_PER_CU
    int fused_vpt = kVPT;                                        // MI_LUMAEQ_FUSED_VPT (8, 16, 20, 24)
    int fused_acquire = 1;                                       // MI_LUMAEQ_FUSED_ACQUIRE
    int fused_fault_inject = 0;                                  // test hook (option "fused_fault_inject")
    int fused_timeout_ms = 2000;                                 // option "fused_timeout_ms"
    int clahe_float_tables = 1;                                  // option "clahe_float_tables": f32 pair tables in LDS (tiles_x <= 14)
    uint8_t*  d_stage_in = nullptr;  size_t stage_in_bytes = 0;  // device frame for the host-pointer forms
    uint8_t*  d_stage_out = nullptr; size_t stage_out_bytes = 0;
    uint8_t*  d_c16 = nullptr;     size_t c16_bytes = 0;         // 16-bit CLAHE: tile histograms + ushort LUTs (N4)
    uint8_t*  d_planes = nullptr;  size_t planes_bytes = 0;      // Y,U,V,Y' planes of the BGR luma pipeline (N3)
    uint8_t*  h_pin_in = nullptr;  size_t pin_in_bytes = 0;      // pinned staging
    uint8_t*  h_pin_out = nullptr; size_t pin_out_bytes = 0;

    // profiling
    bool profiling = false;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> free_events;
    std::vector<hipEvent_t> chunk_events;                        // D2H chunk completion (host-pointer forms)
    mi_profile prof{};
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

#define HIPCHK(c, expr)                                         \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return fail_hip((c), e__, #expr); \
    } while (0)

template <class T>
mi_status grow_dev(mi_ctx* c, T** p, size_t* have, size_t need)
{
    if (need <= *have) return MI_OK;
    if (*p) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(*p)); *p = nullptr; *have = 0; }   // rare: scratch may be in use on a caller stream
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
    if (*p) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipHostFree(*p)); *p = nullptr; *have = 0; }
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, need, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(c, MI_ERR_OOM, "pinned allocation failed"); }
    if (e != hipSuccess) return fail_hip(c, e, "hipHostMalloc");
    *p = (uint8_t*)q; *have = need;
    return MI_OK;
}

// ---- kernel launch with optional event bracketing ---------------------------------------------
struct Bracket {
    mi_ctx* c; hipStream_t s; int kernel; hipEvent_t a = nullptr, b = nullptr; bool on;
    Bracket(mi_ctx* c_, hipStream_t s_, int k) : c(c_), s(s_), kernel(k), on(c_->profiling) {}
    hipError_t begin()
    {
        if (!on) return hipSuccess;
        for (hipEvent_t* e : {&a, &b}) {
            if (!c->free_events.empty()) { *e = c->free_events.back(); c->free_events.pop_back(); }
            else { hipError_t r = hipEventCreate(e); if (r != hipSuccess) return r; }
        }
        return hipEventRecord(a, s);
    }
    hipError_t end()
    {
        if (!on) return hipSuccess;
        hipError_t r = hipEventRecord(b, s);
        c->pending.push_back({a, b, kernel});
        return r;
    }
};

#define LAUNCH(c, s, kid, kern, grid, block, shmem, ...)                          \
    do {                                                                          \
        Bracket br__((c), (s), (kid));                                            \
        HIPCHK((c), br__.begin());                                                \
        hipLaunchKernelGGL(kern, grid, block, shmem, (s), __VA_ARGS__);           \
        HIPCHK((c), hipGetLastError());                                           \
        HIPCHK((c), br__.end());                                                  \
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
    LAUNCH(c, s, MI_K_HIST, hist_partial_kernel, dim3(B, nf), dim3(kThreads), 0, p, c->d_partial);
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

// ---- fused single-read path -----------------------------------------------------------------------------
// Layout of the per-call hand-off block (uint32 words), zeroed by ONE memset node before the launch:
//   [0..31] work counter (u64) | [32..63] status | cnt[nf][32] | ready[nf][32] | ghist[nf][256] | lutpub[nf][128]
bool fused_applicable(const mi_ctx* c, const PlaneArgs& a, const UVJob* uv)
{
    if (!c->fused_mode || !c->fused_slot) return false;
    if (a.src_step != (size_t)a.width || a.dst_step != (size_t)a.width) return false;      // contiguous planes only
    const long long ysz = (long long)a.width * a.height;
    if (ysz % 16 != 0) return false;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | a.src_frame | a.dst_frame) & 15) return false;
    const long long slice = (long long)kThreads * c->fused_vpt;
    const long long T = (ysz / 16 + slice - 1) / slice;
    if (T > (long long)c->cu_count * c->fused_wgs_per_cu / (2 * kMaxFusedCtxPerDevice)) return false;   // co-residency guard (see g_fused_ctx_live)
    if (a.n_frames > (1 << 20)) return false;
    (void)uv;
    return true;
}

mi_status equalize_fused_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const UVJob* uv)
{
    const long long ysz = (long long)a.width * a.height;
    FusedJob j{};
    j.src = a.src; j.dst = a.dst; j.src_frame = (long long)a.src_frame; j.dst_frame = (long long)a.dst_frame;
    j.nvec = ysz / 16; j.total = (int)ysz; j.n_frames = a.n_frames;
    const long long slice = (long long)kThreads * c->fused_vpt;
    j.T = (int)((j.nvec + slice - 1) / slice);
    j.acquire = c->fused_acquire;
    j.fault_inject = c->fused_fault_inject;
    j.timeout_ticks = (unsigned long long)std::max(1, c->fused_timeout_ms) * 100000ull;
    j.U = 0;
    // UV as stand-alone 64 KiB tickets behind each frame's Y tickets: pure streaming work that fills the gaps while
    // other workgroups sit in their hand-off (measured 5 % faster than giving every Y ticket a share of the UV plane)
    if (uv && uv->bytes > 0) { j.uv = *uv; j.U = (int)((uv->bytes + 65535) / 65536); }
    // capacity-based layout so the regions never move between calls with different frame counts
    if ((size_t)a.n_frames > c->fused_cap) {
        size_t cap = std::max<size_t>(64, c->fused_cap);
        while (cap < (size_t)a.n_frames) cap *= 2;
        const size_t words = 64 + cap * (kFlagStride + kFlagStride + 256 + kLutPubWords);
        mi_status st = grow_dev(c, &c->d_fused, &c->fused_bytes, words * sizeof(uint32_t));
        if (st) return st;
        c->fused_cap = cap;
        c->fused_dirty = true;
    }
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) c->fused_capture_safe = true;
        else (void)hipGetLastError();
    }
    if (c->fused_dirty || c->fused_capture_safe) {           // first use, re-layout, reported failure, or graph-replayable mode
        const size_t nwords = c->fused_bytes / sizeof(uint32_t);
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (nwords + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           c->d_fused, nwords);
        HIPCHK(c, hipGetLastError());
        c->fused_work_base = 0;
        c->fused_dirty = false;
        if (c->fused_capture_safe) c->fused_epoch = 0;           // -> epoch 1 below, the same for every (re)play
    }
    const size_t cap = c->fused_cap;
    uint32_t* w = c->d_fused;
    j.work = reinterpret_cast<unsigned long long*>(w);
    j.status = w + 32;
    j.cnt = w + 64;
    j.ready = j.cnt + cap * kFlagStride;
    j.ghist = j.ready + cap * kFlagStride;
    j.lutpub = j.ghist + cap * 256;
    if (++c->fused_epoch == 0) c->fused_epoch = 1;
    j.epoch = c->fused_epoch;
    j.work_base = c->fused_work_base;
    const long long tickets = (long long)(j.T + j.U) * a.n_frames;
    const long long grid = std::min<long long>(tickets, (long long)c->cu_count * c->fused_wgs_per_cu);
    c->fused_work_base += (unsigned long long)tickets + (unsigned long long)grid;   // every workgroup draws one ticket past the end
    c->fused_dirty = true;                                   // cleared below once the launch has been enqueued
    switch (c->fused_vpt) {
        case 8:  LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<8>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        case 20: LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<20>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        case 24: LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<24>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        case 16: LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<16>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        default: return fail(c, MI_ERR_BAD_ARG, "bad fused_vpt");
    }
    c->fused_dirty = false;
    return MI_OK;
}

mi_status equalize_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const UVJob* uv)
{
    if (fused_applicable(c, a, uv)) return equalize_fused_dev(c, s, a, uv);
    for (int f0 = 0; f0 < a.n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, a.n_frames - f0);
        int nparts = 0;
        mi_status st = launch_hist_partials(c, s, a, f0, nf, &nparts);
        if (st) return st;
        st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)nf * 256);
        if (st) return st;
        LAUNCH(c, s, MI_K_EQ_LUT, equalize_lut_kernel, dim3(nf), dim3(kThreads), 0,
               (const uint32_t*)c->d_partial, nparts, (int)((long long)a.width * a.height), c->d_luts, (int32_t*)nullptr);
        st = launch_apply(c, s, a, f0, nf, c->d_luts, uv);
        if (st) return st;
    }
    return MI_OK;
}

// ---- CLAHE ----------------------------------------------------------------------------------------
mi_status clahe_geometry(mi_ctx* c, int width, int height, double clip_limit, int tiles_x, int tiles_y, ClaheGeom* g)
{
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if ((long long)tiles_x * tiles_y > (1 << 20)) return fail(c, MI_ERR_UNSUPPORTED, "tile grid too large");
    g->width = width; g->height = height; g->tiles_x = tiles_x; g->tiles_y = tiles_y;
    long long ew = width, eh = height;
    if (width % tiles_x != 0 || height % tiles_y != 0) {          // clahe.cpp: BOTH pads whenever EITHER is indivisible
        ew = (long long)width + (tiles_x - width % tiles_x);
        eh = (long long)height + (tiles_y - height % tiles_y);
    }
    g->tile_w = (int)(ew / tiles_x); g->tile_h = (int)(eh / tiles_y);
    const long long area = (long long)g->tile_w * g->tile_h;
    if (area > 0x7fffffffLL) return fail(c, MI_ERR_UNSUPPORTED, "tile area must be < 2^31");
    g->lut_scale = 255.0f / (float)(int)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * (int)area / 256);                // double math, truncation (clahe.cpp)
        clip = std::max(clip, 1);
    }
    g->clip = clip;
    g->inv_tw = 1.0f / (float)g->tile_w;
    g->inv_th = 1.0f / (float)g->tile_h;
    return MI_OK;
}

mi_status launch_tile_luts(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const ClaheGeom& g, int f0, int nf, uint8_t* d_luts_out)
{
    const int tiles = g.tiles_x * g.tiles_y;
    // splits per tile: enough workgroups to fill the chip, at least ~8 rows of work each
    long long want = ((long long)c->cu_count * 8 + (long long)tiles * nf - 1) / ((long long)tiles * nf);
    int S = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, g.tile_h / 8), 64LL}));
    mi_status st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * tiles * S * 256 * sizeof(uint32_t));
    if (st) return st;
    const uint8_t* src = a.src + (size_t)f0 * a.src_frame;
    // grid.y = tiles, grid.z = frames
    if (tiles > kMaxGridY) return fail(c, MI_ERR_UNSUPPORTED, "more than 65535 tiles per frame");
    LAUNCH(c, s, MI_K_TILE_HIST, tile_hist_kernel, dim3(S, tiles, nf), dim3(kThreads), 0,
           src, (long long)a.src_step, (long long)a.src_frame, g, c->d_partial);
    LAUNCH(c, s, MI_K_TILE_LUT, tile_lut_kernel, dim3(tiles, nf), dim3(kThreads), 0,
           (const uint32_t*)c->d_partial, S, g, d_luts_out);
    return MI_OK;
}

mi_status launch_interp(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const ClaheGeom& g, int f0, int nf,
                        const uint8_t* d_luts, const UVJob* uv_all)
{
    PlaneBatch p;
    p.src = a.src + (size_t)f0 * a.src_frame; p.dst = a.dst + (size_t)f0 * a.dst_frame;
    p.src_step = (long long)a.src_step; p.dst_step = (long long)a.dst_step;
    p.src_frame = (long long)a.src_frame; p.dst_frame = (long long)a.dst_frame;
    p.row_bytes = a.width; p.rows = a.height;
    UVJob uv{};
    if (uv_all && uv_all->bytes > 0) {
        uv = *uv_all;
        uv.src = uv_all->src ? uv_all->src + (long long)f0 * uv_all->src_frame : nullptr;
        uv.dst = uv_all->dst + (long long)f0 * uv_all->dst_frame;
    }
    const int npairs = g.tiles_x + 1;
    if (npairs <= kMaxPairsLds) {
        const int ngroups = (a.width + kInterpPx - 1) / kInterpPx;
        const int groups = std::min(ngroups, kThreads);
        const int segs = (ngroups + groups - 1) / groups;
        const int bands = g.tiles_y + 1;
        long long want = ((long long)c->cu_count * 8 + (long long)bands * nf * segs - 1) / ((long long)bands * nf * segs);
        const int rows_per_band = g.tile_h + 2 * kBandMargin;
        int subs = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, rows_per_band / 8), 64LL}));
        if ((long long)bands * subs > 0x7fffffffLL || segs > kMaxGridY) return fail(c, MI_ERR_UNSUPPORTED, "image too wide");
        if (npairs <= kMaxPairsLdsF32 && c->clahe_float_tables)
            LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp_kernel<true>, dim3(bands * subs, nf, segs), dim3(kThreads),
                   (size_t)npairs * 256 * 4 * sizeof(float), p, g, d_luts, subs, groups, uv);
        else
            LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp_kernel<false>, dim3(bands * subs, nf, segs), dim3(kThreads),
                   (size_t)npairs * 256 * sizeof(uint32_t), p, g, d_luts, subs, groups, uv);
    } else {
        if (a.height > kMaxGridY) return fail(c, MI_ERR_UNSUPPORTED, "height > 65535 with tiles_x > 62");
        LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp_global_kernel,
               dim3((a.width + kThreads - 1) / kThreads, a.height, nf), dim3(kThreads), 0, p, g, d_luts);
        if (uv.bytes > 0) {
            const int B = blocks_per_frame(c, uv.bytes, 1, nf, 2048);
            LAUNCH(c, s, MI_K_LUT_APPLY, uv_kernel, dim3(B, nf), dim3(kThreads), 0, uv);
        }
    }
    return MI_OK;
}

mi_status clahe_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, double clip_limit, int tiles_x, int tiles_y, const UVJob* uv)
{
    ClaheGeom g;
    mi_status st = clahe_geometry(c, a.width, a.height, clip_limit, tiles_x, tiles_y, &g);
    if (st) return st;
    const int tiles = tiles_x * tiles_y;
    const int chunk = std::min(kMaxGridY, 65535);
    for (int f0 = 0; f0 < a.n_frames; f0 += chunk) {
        const int nf = std::min(chunk, a.n_frames - f0);
        st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)nf * tiles * 256);
        if (st) return st;
        st = launch_tile_luts(c, s, a, g, f0, nf, c->d_luts);
        if (st) return st;
        st = launch_interp(c, s, a, g, f0, nf, c->d_luts, uv);
        if (st) return st;
    }
    return MI_OK;
}

UVJob nv12_uv(const uint8_t* in, uint8_t* out, int width, int height, mi_uv_mode mode)
{
    const long long y = (long long)width * height, uvb = y / 2;       // OpenCVequalHist.cpp:129-130
    UVJob uv;
    uv.src = in ? in + y : nullptr; uv.dst = out + y;
    uv.src_frame = y + uvb; uv.dst_frame = y + uvb;
    uv.bytes = uvb; uv.mode = mode == MI_UV_COPY ? 1 : 0;
    if (uv.mode == 1 && in == out) uv.bytes = 0;                      // in-place passthrough: nothing to move
    return uv;
}

struct Guard {
    mi_ctx* c; std::unique_lock<std::mutex> lk; hipError_t err;
    explicit Guard(mi_ctx* c_) : c(c_), lk(c_->mu) { err = hipSetDevice(c->device); }
};

#define ENTER(ctx)                                                   \
    if (!(ctx)) return MI_ERR_BAD_ARG;                               \
    Guard guard__(ctx);                                              \
    if (guard__.err != hipSuccess) return fail_hip((ctx), guard__.err, "hipSetDevice")

hipStream_t pick_stream(mi_ctx* c, void* stream) { return stream == MI_STREAM_CTX ? c->stream : (hipStream_t)stream; }

// ---- host-pointer plumbing ---------------------------------------------------------------------------
void copy_rows(uint8_t* dst, size_t dst_step, const uint8_t* src, size_t src_step, int width, int height)
{
    if (dst_step == (size_t)width && src_step == (size_t)width) { memcpy(dst, src, (size_t)width * height); return; }
    for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * dst_step, src + (size_t)y * src_step, (size_t)width);
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* mi_version(void) { return "mi_lumaeq 0.1 (gfx950)"; }

const char* mi_status_str(mi_status s)
{
    switch (s) {
        case MI_OK: return "MI_OK";
        case MI_ERR_BAD_ARG: return "MI_ERR_BAD_ARG";
        case MI_ERR_UNSUPPORTED: return "MI_ERR_UNSUPPORTED";
        case MI_ERR_HIP: return "MI_ERR_HIP";
        case MI_ERR_OOM: return "MI_ERR_OOM";
        case MI_ERR_NO_DEVICE: return "MI_ERR_NO_DEVICE";
    }
    return "MI_ERR_?";
}

const char* mi_kernel_name(int k)
{
    static const char* names[MI_K_COUNT] = {"hist_partial_kernel", "equalize_lut_kernel", "lut_apply_kernel",
                                            "tile_hist_kernel", "tile_lut_kernel", "clahe_interp_kernel", "equalize_fused_kernel", "color_kernel"};
    return (k >= 0 && k < MI_K_COUNT) ? names[k] : "?";
}

int mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

mi_status mi_ctx_create(int device, mi_ctx** out)
{
    if (!out) return MI_ERR_BAD_ARG;
    *out = nullptr;
    const int n = mi_device_count();
    if (n <= 0 || device < 0 || device >= n) return MI_ERR_NO_DEVICE;
    mi_ctx* c = new (std::nothrow) mi_ctx();
    if (!c) return MI_ERR_OOM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return MI_ERR_HIP;
    }
    // the 16-bit tile histogram uses 128 KiB of dynamic LDS (above the 64 KiB default limit)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_hist16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kHalf16 * (int)sizeof(uint32_t));
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
    *out = c;
    return MI_OK;
}

void mi_ctx_destroy(mi_ctx* c)
{
    if (!c) return;
    if (c->fused_slot && c->device >= 0 && c->device < kMaxDevices) g_fused_ctx_live[c->device].fetch_sub(1);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    for (auto e : c->chunk_events) (void)hipEventDestroy(e);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_luts) (void)hipFree(c->d_luts);
    if (c->d_fused) (void)hipFree(c->d_fused);
    if (c->d_planes) (void)hipFree(c->d_planes);
    if (c->d_c16) (void)hipFree(c->d_c16);
    if (c->h_status) (void)hipHostFree(c->h_status);
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
    c->profiling = enabled != 0;
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
        c->free_events.push_back(p.a);
        c->free_events.push_back(p.b);
    }
    c->pending.clear();
    if (out) *out = c->prof;
    if (reset) c->prof = mi_profile{};
    return MI_OK;
}

mi_status mi_host_register(void* ptr, size_t bytes)
{
    if (!ptr || bytes == 0) return MI_ERR_BAD_ARG;
    if (mi_device_count() <= 0) return MI_ERR_NO_DEVICE;
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.push_back({(uintptr_t)ptr, (uintptr_t)ptr + bytes});
    return MI_OK;
}

mi_status mi_host_unregister(void* ptr)
{
    if (!ptr) return MI_ERR_BAD_ARG;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = std::find_if(g_pinned.begin(), g_pinned.end(), [&](const PinnedRange& r) { return r.lo == (uintptr_t)ptr; });
        if (it == g_pinned.end()) return MI_ERR_BAD_ARG;
        g_pinned.erase(it);
    }
    if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return MI_ERR_HIP; }
    return MI_OK;
}

mi_status mi_ctx_set_option(mi_ctx* c, const char* name, int value)
{
    ENTER(c);
    if (!name) return fail(c, MI_ERR_BAD_ARG, "null option name");
    if (!strcmp(name, "fused")) { c->fused_mode = value; return MI_OK; }
    if (!strcmp(name, "fused_wgs_per_cu")) { c->fused_wgs_per_cu = std::max(1, std::min(8, value)); return MI_OK; }
    if (!strcmp(name, "fused_vpt")) { if (value != 8 && value != 16 && value != 20 && value != 24) return fail(c, MI_ERR_BAD_ARG, "fused_vpt must be 8, 16, 20 or 24"); c->fused_vpt = value; return MI_OK; }
    if (!strcmp(name, "fused_acquire")) { c->fused_acquire = value != 0; return MI_OK; }
    if (!strcmp(name, "fused_fault_inject")) { c->fused_fault_inject = value != 0; return MI_OK; }
    if (!strcmp(name, "fused_timeout_ms")) { c->fused_timeout_ms = std::max(1, value); return MI_OK; }
    if (!strcmp(name, "clahe_float_tables")) { c->clahe_float_tables = value != 0; return MI_OK; }
    return fail(c, MI_ERR_BAD_ARG, "unknown option");
}

// Waits for `stream` and reports a device-side failure of the fused kernel's bounded waits.
mi_status mi_ctx_synchronize(mi_ctx* c, void* stream)
{
    ENTER(c);
    hipStream_t s = pick_stream(c, stream);
    HIPCHK(c, hipStreamSynchronize(s));
    if (!c->d_fused) return MI_OK;
    if (!c->h_status) { void* q = nullptr; HIPCHK(c, hipHostMalloc(&q, 64, hipHostMallocDefault)); c->h_status = (uint32_t*)q; }
    HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_fused + 32, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (*c->h_status != 0) {
        c->fused_dirty = true;                                   // hand-off block is in an unknown state: zero it before the next launch
        c->last_hip = 0;
        return fail(c, MI_ERR_HIP, "fused equalize kernel: a bounded inter-workgroup wait expired; output invalid");
    }
    return MI_OK;
}

// ---- device-resident batched forms ------------------------------------------------------------------
mi_status mi_equalize_hist_u8_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, void* stream)
{
    ENTER(c);
    PlaneArgs a{(const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride, width, height, n_frames};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    return equalize_dev(c, pick_stream(c, stream), a, nullptr);
}

mi_status mi_equalize_hist_nv12_batch_dev(mi_ctx* c, const void* d_in, void* d_out, int width, int height, int n_frames,
                                          mi_uv_mode uv_mode, void* stream)
{
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
// Host rows -> pinned staging -> H2D -> kernels -> D2H -> pinned -> host rows, all on the context's
// stream, synchronous on return.  `nv12_mode` < 0: plain Y plane; otherwise whole NV12 frame.
static mi_status host_op(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                         int width, int height, int nv12_mode, bool is_clahe, double clip_limit, int tiles_x, int tiles_y)
{
    const size_t ybytes = (size_t)width * height;
    const size_t uvbytes = nv12_mode >= 0 ? ybytes / 2 : 0;
    const bool copy_uv_in = nv12_mode == MI_UV_COPY;
    const size_t in_bytes = ybytes + (copy_uv_in ? uvbytes : 0);
    const size_t frame_bytes = ybytes + uvbytes;
    mi_status st;
    if ((st = grow_pinned(c, &c->h_pin_in, &c->pin_in_bytes, in_bytes))) return st;
    if ((st = grow_pinned(c, &c->h_pin_out, &c->pin_out_bytes, frame_bytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_in, &c->stage_in_bytes, frame_bytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, frame_bytes))) return st;
    hipStream_t s = c->stream;
    // Caller-pinned, contiguous buffers (mi_host_register) are DMA'd directly; everything else is staged.
    const bool in_direct = src_step == (size_t)width && host_range_pinned(src, in_bytes);
    const bool out_direct = dst_step == (size_t)width && host_range_pinned(dst, frame_bytes);
    // Chunked staging: the host copy of chunk i+1 into pinned memory overlaps the DMA of chunk i (and the other
    // way round on the way back), so a frame costs ~max(memcpy, PCIe) per direction instead of their sum.
    const int rows_per_chunk = std::max(1, (int)((size_t)(2u << 20) / (size_t)width));
    if (in_direct) {
        HIPCHK(c, hipMemcpyAsync(c->d_stage_in, src, in_bytes, hipMemcpyHostToDevice, s));
    } else {
        for (int y0 = 0; y0 < height; y0 += rows_per_chunk) {
            const int nr = std::min(rows_per_chunk, height - y0);
            const size_t off = (size_t)y0 * width;
            copy_rows(c->h_pin_in + off, (size_t)width, src + (size_t)y0 * src_step, src_step, width, nr);
            HIPCHK(c, hipMemcpyAsync(c->d_stage_in + off, c->h_pin_in + off, (size_t)nr * width, hipMemcpyHostToDevice, s));
        }
        if (copy_uv_in) {                                           // tightly packed NV12 (src_step == width)
            memcpy(c->h_pin_in + ybytes, src + ybytes, uvbytes);
            HIPCHK(c, hipMemcpyAsync(c->d_stage_in + ybytes, c->h_pin_in + ybytes, uvbytes, hipMemcpyHostToDevice, s));
        }
    }
    PlaneArgs a{c->d_stage_in, (size_t)width, frame_bytes, c->d_stage_out, (size_t)width, frame_bytes, width, height, 1};
    UVJob uv{};
    if (nv12_mode >= 0) uv = nv12_uv(c->d_stage_in, c->d_stage_out, width, height, (mi_uv_mode)nv12_mode);
    st = is_clahe ? clahe_dev(c, s, a, clip_limit, tiles_x, tiles_y, nv12_mode >= 0 ? &uv : nullptr)
                  : equalize_dev(c, s, a, nv12_mode >= 0 ? &uv : nullptr);
    if (st) return st;
    const bool check_status = !is_clahe && c->d_fused;
    if (check_status) {
        if (!c->h_status) { void* q = nullptr; HIPCHK(c, hipHostMalloc(&q, 64, hipHostMallocDefault)); c->h_status = (uint32_t*)q; }
        HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_fused + 32, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    if (out_direct) {
        HIPCHK(c, hipMemcpyAsync(dst, c->d_stage_out, frame_bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (check_status && *c->h_status != 0) {
            c->fused_dirty = true;
            return fail(c, MI_ERR_HIP, "fused equalize kernel: a bounded inter-workgroup wait expired; output invalid");
        }
        return MI_OK;
    }
    // device -> pinned in chunks, each followed by an event; then drain chunk by chunk into the caller's rows
    struct Chunk { size_t off, bytes; int y0, nr; };
    std::vector<Chunk> chunks;
    for (int y0 = 0; y0 < height; y0 += rows_per_chunk) {
        const int nr = std::min(rows_per_chunk, height - y0);
        chunks.push_back({(size_t)y0 * width, (size_t)nr * width, y0, nr});
    }
    if (uvbytes) chunks.push_back({ybytes, uvbytes, -1, 0});
    while (c->chunk_events.size() < chunks.size()) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->chunk_events.push_back(e);
    }
    for (size_t i = 0; i < chunks.size(); ++i) {
        HIPCHK(c, hipMemcpyAsync(c->h_pin_out + chunks[i].off, c->d_stage_out + chunks[i].off, chunks[i].bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->chunk_events[i], s));
    }
    for (size_t i = 0; i < chunks.size(); ++i) {
        HIPCHK(c, hipEventSynchronize(c->chunk_events[i]));
        if (i == 0 && check_status && *c->h_status != 0) {
            c->fused_dirty = true;
            (void)hipStreamSynchronize(s);
            return fail(c, MI_ERR_HIP, "fused equalize kernel: a bounded inter-workgroup wait expired; output invalid");
        }
        if (chunks[i].y0 >= 0)
            copy_rows(dst + (size_t)chunks[i].y0 * dst_step, dst_step, c->h_pin_out + chunks[i].off, (size_t)width, width, chunks[i].nr);
        else
            memcpy(dst + ybytes, c->h_pin_out + ybytes, uvbytes);
    }
    return MI_OK;
}

mi_status mi_equalize_hist_u8(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height)
{
    ENTER(c);
    PlaneArgs a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0) return st;
    return host_op(c, src, src_step, dst, dst_step, width, height, -1, false, 0.0, 0, 0);
}

mi_status mi_clahe_u8(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height,
                      double clip_limit, int tiles_x, int tiles_y)
{
    ENTER(c);
    PlaneArgs a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0) return MI_OK;
    return host_op(c, src, src_step, dst, dst_step, width, height, -1, true, clip_limit, tiles_x, tiles_y);
}

mi_status mi_equalize_hist_nv12(mi_ctx* c, const uint8_t* in, uint8_t* out, int width, int height, mi_uv_mode uv_mode)
{
    ENTER(c);
    if (uv_mode != MI_UV_FILL128 && uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    PlaneArgs a{in, (size_t)std::max(width, 0), 0, out, (size_t)std::max(width, 0), 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st || width == 0 || height == 0) return st;
    return host_op(c, in, (size_t)width, out, (size_t)width, width, height, (int)uv_mode, false, 0.0, 0, 0);
}

mi_status mi_clahe_nv12(mi_ctx* c, const uint8_t* in, uint8_t* out, int width, int height, mi_uv_mode uv_mode,
                        double clip_limit, int tiles_x, int tiles_y)
{
    ENTER(c);
    if (uv_mode != MI_UV_FILL128 && uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    PlaneArgs a{in, (size_t)std::max(width, 0), 0, out, (size_t)std::max(width, 0), 0, width, height, 1};
    mi_status st = check_plane(c, a, true);
    if (st) return st;
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0) return MI_OK;
    return host_op(c, in, (size_t)width, out, (size_t)width, width, height, (int)uv_mode, true, clip_limit, tiles_x, tiles_y);
}

}  // extern "C"

// ---- colour-domain neighbours ---------------------------------------------------------------
namespace {

struct Color3Args {
    const uint8_t* src; size_t src_step, src_frame;
    uint8_t* dst; size_t dst_step, dst_frame;
    int width, height, n_frames;
};

mi_status check_color3(mi_ctx* c, const Color3Args& a)
{
    if (a.width < 0 || a.height < 0 || a.n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "negative size");
    if (a.width == 0 || a.height == 0 || a.n_frames == 0) return MI_OK;
    if (!a.src || !a.dst) return fail(c, MI_ERR_BAD_ARG, "null image pointer");
    if (a.src_step < (size_t)a.width * 3 || a.dst_step < (size_t)a.width * 3) return fail(c, MI_ERR_BAD_ARG, "step < 3*width");
    if ((long long)a.width * a.height > 0x7fffffffLL / 3) return fail(c, MI_ERR_UNSUPPORTED, "image too large");
    return MI_OK;
}

template <int MODE>
mi_status launch_color(mi_ctx* c, hipStream_t s, ColorJob j, int n_frames)
{
    const long long px = j.row_px * (long long)j.rows;
    const int gy = std::min(j.rows, 65535);
    long long bx = ((long long)c->cu_count * 8 + (long long)gy * n_frames - 1) / ((long long)gy * n_frames);
    bx = std::max<long long>(1, std::min<long long>(bx, (j.row_px / 16 + kThreads - 1) / kThreads + 1));
    (void)px;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        ColorJob q = j;
        if (q.src) q.src += (long long)f0 * j.src_frame;
        if (q.dst) q.dst += (long long)f0 * j.dst_frame;
        if (q.p0) { q.p0 += (long long)f0 * j.plane_frame; q.p1 += (long long)f0 * j.plane_frame; q.p2 += (long long)f0 * j.plane_frame; }
        LAUNCH(c, s, MI_K_COLOR, color_kernel<MODE>, dim3((unsigned)bx, gy, nf), dim3(kThreads), 0, q);
    }
    return MI_OK;
}

ColorJob color_job(const Color3Args& a)
{
    ColorJob j{};
    j.src = a.src; j.dst = a.dst;
    j.src_frame = (long long)a.src_frame; j.dst_frame = (long long)a.dst_frame;
    const bool contiguous = (!a.src || a.src_step == (size_t)a.width * 3) && (!a.dst || a.dst_step == (size_t)a.width * 3);
    if (contiguous || a.height == 1) { j.rows = 1; j.row_px = (long long)a.width * a.height; j.src_step = j.dst_step = j.row_px * 3; }
    else { j.rows = a.height; j.row_px = a.width; j.src_step = (long long)a.src_step; j.dst_step = (long long)a.dst_step; }
    return j;
}

mi_status cvt_color_dev(mi_ctx* c, hipStream_t s, const Color3Args& a, int code)
{
    ColorJob j = color_job(a);
    if (code == MI_COLOR_BGR2YUV) return launch_color<0>(c, s, j, a.n_frames);
    if (code == MI_COLOR_YUV2BGR) return launch_color<1>(c, s, j, a.n_frames);
    return fail(c, MI_ERR_UNSUPPORTED, "colour code must be MI_COLOR_BGR2YUV (82) or MI_COLOR_YUV2BGR (84)");
}

mi_status bgr_luma_dev(mi_ctx* c, hipStream_t s, const Color3Args& a, int op, double clip, int tx, int ty)
{
    if (op != MI_OP_EQUALIZE && op != MI_OP_CLAHE) return fail(c, MI_ERR_BAD_ARG, "op must be MI_OP_EQUALIZE or MI_OP_CLAHE");
    if (op == MI_OP_CLAHE && (tx <= 0 || ty <= 0)) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    const size_t plane = ((size_t)a.width * a.height + 15) & ~(size_t)15;           // keep every plane 16-B aligned
    const size_t per_frame = plane * 4;                                             // Y, U, V, Y'
    mi_status st = grow_dev(c, &c->d_planes, &c->planes_bytes, per_frame * (size_t)a.n_frames);
    if (st) return st;
    uint8_t* Y = c->d_planes; uint8_t* U = Y + plane; uint8_t* V = U + plane; uint8_t* Y2 = V + plane;
    // cvtColor(BGR2YUV) + split
    Color3Args in = a; in.dst = nullptr;
    ColorJob j = color_job(in);
    j.dst = nullptr; j.p0 = Y; j.p1 = U; j.p2 = V; j.plane_frame = (long long)per_frame;
    // planes are written tightly (row pitch = width), so plane offsets use row*width even for strided sources
    if ((st = launch_color<2>(c, s, j, a.n_frames))) return st;
    // the luma op on the Y planes
    PlaneArgs pa{Y, (size_t)a.width, per_frame, Y2, (size_t)a.width, per_frame, a.width, a.height, a.n_frames};
    st = op == MI_OP_EQUALIZE ? equalize_dev(c, s, pa, nullptr) : clahe_dev(c, s, pa, clip, tx, ty, nullptr);
    if (st) return st;
    // merge + cvtColor(YUV2BGR)
    Color3Args out = a; out.src = nullptr;
    ColorJob k = color_job(out);
    k.src = nullptr; k.p0 = Y2; k.p1 = U; k.p2 = V; k.plane_frame = (long long)per_frame;
    return launch_color<3>(c, s, k, a.n_frames);
}

// host images staged like host_op(): rows -> pinned -> device (tight) -> op -> pinned -> rows
mi_status color_host_op(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height,
                        bool luma, int code_or_op, double clip, int tx, int ty)
{
    const size_t row = (size_t)width * 3, bytes = row * height;
    mi_status st;
    if ((st = grow_pinned(c, &c->h_pin_in, &c->pin_in_bytes, bytes))) return st;
    if ((st = grow_pinned(c, &c->h_pin_out, &c->pin_out_bytes, bytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_in, &c->stage_in_bytes, bytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, bytes))) return st;
    hipStream_t s = c->stream;
    for (int y = 0; y < height; ++y) memcpy(c->h_pin_in + (size_t)y * row, src + (size_t)y * src_step, row);
    HIPCHK(c, hipMemcpyAsync(c->d_stage_in, c->h_pin_in, bytes, hipMemcpyHostToDevice, s));
    Color3Args a{c->d_stage_in, row, bytes, c->d_stage_out, row, bytes, width, height, 1};
    st = luma ? bgr_luma_dev(c, s, a, code_or_op, clip, tx, ty) : cvt_color_dev(c, s, a, code_or_op);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->h_pin_out, c->d_stage_out, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * dst_step, c->h_pin_out + (size_t)y * row, row);
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_cvt_color_u8c3_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                      void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                      int width, int height, int n_frames, int code, void* stream)
{
    ENTER(c);
    Color3Args a{(const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride, width, height, n_frames};
    mi_status st = check_color3(c, a);
    if (st) return st;
    if (code != MI_COLOR_BGR2YUV && code != MI_COLOR_YUV2BGR) return fail(c, MI_ERR_UNSUPPORTED, "colour code must be 82 (BGR2YUV) or 84 (YUV2BGR)");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    return cvt_color_dev(c, pick_stream(c, stream), a, code);
}

mi_status mi_bgr_luma_op_u8c3_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, int op, double clip_limit, int tiles_x, int tiles_y, void* stream)
{
    ENTER(c);
    Color3Args a{(const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride, width, height, n_frames};
    mi_status st = check_color3(c, a);
    if (st) return st;
    if (op != MI_OP_EQUALIZE && op != MI_OP_CLAHE) return fail(c, MI_ERR_BAD_ARG, "op must be MI_OP_EQUALIZE or MI_OP_CLAHE");
    if (op == MI_OP_CLAHE && (tiles_x <= 0 || tiles_y <= 0)) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    return bgr_luma_dev(c, pick_stream(c, stream), a, op, clip_limit, tiles_x, tiles_y);
}

mi_status mi_cvt_color_u8c3(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height, int code)
{
    ENTER(c);
    Color3Args a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_color3(c, a);
    if (st) return st;
    if (code != MI_COLOR_BGR2YUV && code != MI_COLOR_YUV2BGR) return fail(c, MI_ERR_UNSUPPORTED, "colour code must be 82 (BGR2YUV) or 84 (YUV2BGR)");
    if (width == 0 || height == 0) return MI_OK;
    return color_host_op(c, src, src_step, dst, dst_step, width, height, false, code, 0.0, 0, 0);
}

mi_status mi_bgr_luma_op_u8c3(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height,
                              int op, double clip_limit, int tiles_x, int tiles_y)
{
    ENTER(c);
    Color3Args a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_color3(c, a);
    if (st) return st;
    if (op != MI_OP_EQUALIZE && op != MI_OP_CLAHE) return fail(c, MI_ERR_BAD_ARG, "op must be MI_OP_EQUALIZE or MI_OP_CLAHE");
    if (op == MI_OP_CLAHE && (tiles_x <= 0 || tiles_y <= 0)) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0) return MI_OK;
    return color_host_op(c, src, src_step, dst, dst_step, width, height, true, op, clip_limit, tiles_x, tiles_y);
}

}  // extern "C"

// ---- CLAHE on CV_16UC1 ----------------------------------------------------------------------
namespace {

mi_status clahe16_dev(mi_ctx* c, hipStream_t s, const uint8_t* src, size_t src_step, size_t src_frame, uint8_t* dst, size_t dst_step,
                      size_t dst_frame, int width, int height, int n_frames, double clip_limit, int tiles_x, int tiles_y)
{
    ClaheGeom g;
    mi_status st = clahe_geometry(c, width, height, clip_limit, tiles_x, tiles_y, &g);
    if (st) return st;
    const int tiles = tiles_x * tiles_y;
    if (tiles > kMaxGridY || height > kMaxGridY) return fail(c, MI_ERR_UNSUPPORTED, "16-bit CLAHE: more than 65535 tiles or rows");
    const long long area = (long long)g.tile_w * g.tile_h;
    const float lut_scale16 = 65535.0f / (float)(int)area;
    int clip16 = 0;
    if (clip_limit > 0.0) { clip16 = (int)(clip_limit * (int)area / 65536); clip16 = std::max(clip16, 1); }
    // scratch per frame: tile histograms (u32) + ushort LUTs; frames are processed in chunks that keep it <= ~256 MiB
    const size_t per_frame = (size_t)tiles * kHist16 * (sizeof(uint32_t) + sizeof(uint16_t));
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_frames, ((size_t)256 << 20) / per_frame));
    st = grow_dev(c, &c->d_c16, &c->c16_bytes, per_frame * (size_t)chunk);
    if (st) return st;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = std::min(chunk, n_frames - f0);
        uint32_t* hist = reinterpret_cast<uint32_t*>(c->d_c16);
        uint16_t* luts = reinterpret_cast<uint16_t*>(c->d_c16 + (size_t)nf * tiles * kHist16 * sizeof(uint32_t));
        LAUNCH(c, s, MI_K_TILE_HIST, tile_hist16_kernel, dim3(tiles, nf), dim3(1024), kHalf16 * sizeof(uint32_t),
               src + (size_t)f0 * src_frame, (long long)src_step, (long long)src_frame, g, hist);
        LAUNCH(c, s, MI_K_TILE_LUT, tile_lut16_kernel, dim3(tiles, nf), dim3(1024), 0, (const uint32_t*)hist, g, lut_scale16, clip16, luts);
        LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16_kernel, dim3((width + kThreads - 1) / kThreads, height, nf), dim3(kThreads), 0,
               src + (size_t)f0 * src_frame, (long long)src_step, (long long)src_frame,
               dst + (size_t)f0 * dst_frame, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)luts);
    }
    return MI_OK;
}

mi_status check_u16(mi_ctx* c, const void* src, size_t src_step, const void* dst, size_t dst_step, int width, int height, int n_frames,
                    int tiles_x, int tiles_y)
{
    if (width < 0 || height < 0 || n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "negative size");
    if (tiles_x <= 0 || tiles_y <= 0) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    if (!src || !dst) return fail(c, MI_ERR_BAD_ARG, "null plane pointer");
    if (src_step < (size_t)width * 2 || dst_step < (size_t)width * 2) return fail(c, MI_ERR_BAD_ARG, "step < 2*width");
    if ((src_step | dst_step | (uintptr_t)src | (uintptr_t)dst) & 1) return fail(c, MI_ERR_BAD_ARG, "16-bit planes must be 2-byte aligned");
    if ((long long)width * height > 0x3fffffffLL) return fail(c, MI_ERR_UNSUPPORTED, "image too large");
    if (width > (1 << 24) || height > (1 << 24)) return fail(c, MI_ERR_UNSUPPORTED, "width/height must be <= 2^24");
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_clahe_u16_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                 void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                 int width, int height, int n_frames, double clip_limit, int tiles_x, int tiles_y, void* stream)
{
    ENTER(c);
    mi_status st = check_u16(c, d_src, src_step, d_dst, dst_step, width, height, n_frames, tiles_x, tiles_y);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    return clahe16_dev(c, pick_stream(c, stream), (const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride,
                       width, height, n_frames, clip_limit, tiles_x, tiles_y);
}

mi_status mi_clahe_u16(mi_ctx* c, const uint16_t* src, size_t src_step, uint16_t* dst, size_t dst_step, int width, int height,
                       double clip_limit, int tiles_x, int tiles_y)
{
    ENTER(c);
    mi_status st = check_u16(c, src, src_step, dst, dst_step, width, height, 1, tiles_x, tiles_y);
    if (st || width == 0 || height == 0) return st;
    const size_t row = (size_t)width * 2, bytes = row * height;
    if ((st = grow_pinned(c, &c->h_pin_in, &c->pin_in_bytes, bytes))) return st;
    if ((st = grow_pinned(c, &c->h_pin_out, &c->pin_out_bytes, bytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_in, &c->stage_in_bytes, bytes))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, bytes))) return st;
    hipStream_t s = c->stream;
    for (int y = 0; y < height; ++y) memcpy(c->h_pin_in + (size_t)y * row, (const uint8_t*)src + (size_t)y * src_step, row);
    HIPCHK(c, hipMemcpyAsync(c->d_stage_in, c->h_pin_in, bytes, hipMemcpyHostToDevice, s));
    st = clahe16_dev(c, s, c->d_stage_in, row, bytes, c->d_stage_out, row, bytes, width, height, 1, clip_limit, tiles_x, tiles_y);
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->h_pin_out, c->d_stage_out, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int y = 0; y < height; ++y) memcpy((uint8_t*)dst + (size_t)y * dst_step, c->h_pin_out + (size_t)y * row, row);
    return MI_OK;
}

}  // extern "C"
