// pipe.inc.hpp -- mi_pipe: asynchronous, in-order NV12 frame pipeline on ONE context (host frame in -> host frame out)
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).
//
// What it replaces: the reference's worker does map -> op -> rebuild -> push synchronously per frame
// (OpenCVequalHist.cpp:102-196), and its accelerator variant blocks on every transfer (OpenCLequalHist.cpp:356-365:
// write, write, task, finish, read, finish).  On a discrete GPU that serialisation leaves the link idle two thirds of the
// time, so one worker per GPU cannot keep a GPU fed.  A pipe keeps `depth` frames in flight on three HIP streams:
//
//     s_h2d :  H2D(k+2)            | copy engine, host -> device
//     s_k   :  op(k+1) [+ finish]  | compute; waits for H2D(k+1) by event
//     s_d2h :  D2H(k)              | copy engine, device -> host; waits for op(k) by event
//
// so PCIe runs in both directions at once while the kernels (tens of microseconds per 4K frame) hide behind it.
// Frames complete in submission order (mi_pipe_wait).  Host memory that was registered with mi_host_register (a recycled
// frame pool) is DMA'd asynchronously as it is; unpinned (pageable) memory is copied through pinned staging buffers of the
// slot by the calling thread and the context's helper thread (into them at submit, out of them at wait) -- the library never
// hands the runtime memory it did not pin itself (host_op() in capi.inc.hpp says why).
// Error paths: a failure inside submit drains the three streams before it returns (nothing stays in flight on the caller's
// `in` / `out`) and does not occupy a slot; a failure inside wait drains them and STILL retires the slot, so the tags a caller
// keeps (FramePool's inflight queue, Pipe._held in the Python binding) stay in step with the pipe.  One pipe per context; while
// frames are pending the context's other compute entry points answer MI_ERR_BUSY.
// All pipes of a process on one device share the SAME streams: a second worker on a GPU then interleaves its frames
// into the same queues instead of adding queues (8 streams on one device were measured 30 % SLOWER than 3: HIP multiplexes
// streams onto 4 hardware queues, and unrelated copies end up ordered behind each other); as a side effect the fused
// kernels of different contexts never overlap on a device when they come from pipes.
// Each copy direction has TWO streams and consecutive frames alternate between them (option "pipe_copy_streams", default 2):
// the copy engine idles ~10-13 us between two dependent copies of one stream (profiles/r03_a_host_form_timeline.txt), 6-8 % of
// a 4K plane's transfer time; with the next frame's copy already running on the other stream the link never waits for it.
// UV handling: "host" (default for Y-only ops) moves only the Y plane over the bus and fills / copies the UV half on the
// host inside mi_pipe_wait while the engines are busy; "device" ships whole NV12 frames and lets the kernels do it.

struct PipeSlot {
    int lane = 0;                                                 // which of the two copy streams per direction carries this frame
    uint8_t* d_in = nullptr; uint8_t* d_out = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_k = nullptr, ev_done = nullptr;
    const uint8_t* in = nullptr; uint8_t* out = nullptr;
    uint64_t tag = 0;
    bool out_staged = false;                                      // the D2H queued at submit lands in h_out: mi_pipe_wait copies it to the caller's frame
    uint8_t* h_in = nullptr; uint8_t* h_out = nullptr;            // pinned staging for frames the caller did not pin (allocated on first use)
    mi_status st = MI_OK;
};

// process-wide stream triple per device, created by the first pipe, destroyed with the last
struct PipeStreams {
    hipStream_t h2d[2] = {nullptr, nullptr}, k = nullptr, d2h[2] = {nullptr, nullptr};
    int users = 0;
    hipStream_t* all(int i) { return i == 0 ? &h2d[0] : i == 1 ? &h2d[1] : i == 2 ? &k : i == 3 ? &d2h[0] : &d2h[1]; }
};
static std::mutex g_pipe_streams_mu;
static PipeStreams g_pipe_streams[kMaxDevices];
static std::atomic<uint32_t> g_pipe_lane[kMaxDevices];          // frames submitted on the device by ALL pipes: picks the copy lane
static std::atomic<int> g_pipe_users[kMaxDevices];               // pipes sharing the device's streams (PipeStreams::users, readable without its lock)
// One frame's enqueue sequence (H2D, event, wait, kernels, event, wait, D2H, event) is issued under this per-device lock.  The
// runtime dispatches directly from the calling thread and some of these calls block for tens of microseconds while holding the
// stream's lock; two workers interleaving their calls on the shared streams call by call made EACH sequence slower (2 workers
// 4650 frames/s against 5400 for one; with AMD_DIRECT_DISPATCH=0 -- submission from the runtime's own thread -- 5290).  Frame by
// frame the streams see exactly the one-worker call pattern, while the workers' own work (UV half, staging copies, delivery,
// waiting) still runs in parallel.
static std::mutex g_pipe_submit_mu[kMaxDevices];

struct mi_pipe {
    mi_ctx* c = nullptr;
    mi_pipe_config cfg{};
    hipStream_t s_h2d[2] = {nullptr, nullptr}, s_k = nullptr, s_d2h[2] = {nullptr, nullptr};
    int n_copy = 2;                                               // copy streams per direction in use (1 or 2)
    bool private_streams = false;                                 // MI_LUMAEQ_PIPE_PRIVATE_STREAMS=1: this pipe owns its five streams
    std::vector<PipeSlot> slots;
    size_t head = 0, count = 0;
    size_t ybytes = 0, uvbytes = 0, xfer_in = 0, xfer_out = 0;
    bool uv_dev = false;
    uint32_t* h_hard = nullptr;                                   // pinned: [slot] = device "unrecoverable frames" counter after that frame
    int wait_mode = 0;                                            // how mi_pipe_wait waits (MI_LUMAEQ_PIPE_WAIT, read once at mi_pipe_create):
                                                                  // 0 poll with back-off (default), 1 hipEventSynchronize, 2 poll without sleeping
    int wait_spin_us = 0;                                         // mode 0: how long it polls before the first sleep when frames are queued behind
                                                                  // (MI_LUMAEQ_PIPE_WAIT_SPIN_US; 4K: 0 / 5 / 20 / 50 us = 1.51 / 1.58 / 1.63 / 1.80 host cores
                                                                  // per unpaced worker at the same 5.44 k frames/s, profiles/r04_d_*; set by frame size in
                                                                  // mi_pipe_create: frames below 8 MiB are only polled)
    uint64_t submitted = 0, completed = 0;
};

namespace {

mi_status pipe_run_op(mi_pipe* p, PipeSlot& sl)
{
    mi_ctx* c = p->c;
    const mi_pipe_config& g = p->cfg;
    const size_t fstride = p->ybytes + p->uvbytes;
    if (g.op == MI_OP_CHANNELS) return nv12_bgr_equalize_dev(c, p->s_k, sl.d_in, fstride, sl.d_out, fstride, g.width, g.height, 1);
    PlaneArgs a{sl.d_in, (size_t)g.width, fstride, sl.d_out, (size_t)g.width, fstride, g.width, g.height, 1};
    UVJob uv{};
    const UVJob* puv = nullptr;
    if (p->uv_dev) { uv = nv12_uv(sl.d_in, sl.d_out, g.width, g.height, g.uv_mode); puv = &uv; }
    if (g.op == MI_OP_CLAHE) return clahe_dev(c, p->s_k, a, g.clip_limit, g.tiles_x, g.tiles_y, puv);
    // "frames the repair refused" of THIS launch goes straight into the slot's word of pinned host memory (written by the finish
    // kernel, read by mi_pipe_wait after the frame's download): exact per frame, and no 4-byte copy behind every download
    uint32_t* word = p->h_hard + (&sl - p->slots.data());
    *word = 0;
    c->fused_hard_word = word;
    const mi_status st = equalize_dev(c, p->s_k, a, puv);
    c->fused_hard_word = nullptr;
    return st;
}

// The runtime multiplexes HIP streams onto a few hardware queues (4 per priority level), and a hardware queue executes its
// packets in order: the marker behind a 170 us device->host copy, sharing a hardware queue with the compute stream, holds the
// next frame's kernels back until that copy is done (measured: a second worker's streams shifted the mapping and cost 14 %;
// GPU_MAX_HW_QUEUES=16 restored it).  The runtime keeps a separate pool of hardware queues per stream priority, so the three
// roles get three priorities: kernels high, uploads normal, downloads low -- a kernel packet can then never queue up behind a
// copy's marker, whatever else the process has created.  (Priority only orders compute queues; copies go to the DMA engines.)
hipError_t pipe_stream_create(hipStream_t* s, int role /* 0,1 upload lanes; 2 kernels; 3,4 download lanes */)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    int prio = role == 2 ? greatest : (role >= 3 ? least : (least + greatest) / 2);
    if (const char* v = getenv("MI_LUMAEQ_PIPE_UPLOAD_PRIORITY")) { if (role < 2) prio = atoi(v) ? greatest : least; }
    if (getenv("MI_LUMAEQ_PIPE_FLAT_PRIORITY")) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio);
}

void pipe_free(mi_pipe* p)
{
    if (!p) return;
    (void)hipSetDevice(p->c->device);
    for (hipStream_t s : {p->s_h2d[0], p->s_h2d[1], p->s_k, p->s_d2h[0], p->s_d2h[1]}) if (s) (void)hipStreamSynchronize(s);
    g_pending_dma.retire_all(p);                                  // frames never waited for: their transfers ended with the streams above
    for (auto& sl : p->slots) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        for (hipEvent_t e : {sl.ev_h2d, sl.ev_k, sl.ev_done}) if (e) (void)hipEventDestroy(e);
    }
    if (p->h_hard) (void)hipHostFree(p->h_hard);
    if (p->s_k && p->private_streams) {
        for (hipStream_t s : {p->s_h2d[0], p->s_h2d[1], p->s_k, p->s_d2h[0], p->s_d2h[1]}) if (s) (void)hipStreamDestroy(s);
    } else if (p->s_k) {
        std::lock_guard<std::mutex> lk(g_pipe_streams_mu);
        PipeStreams& ps = g_pipe_streams[p->c->device];
        g_pipe_users[p->c->device].fetch_sub(1, std::memory_order_relaxed);
        if (--ps.users == 0) {
            for (int i = 0; i < 5; ++i) if (*ps.all(i)) (void)hipStreamDestroy(*ps.all(i));
            ps = PipeStreams{};
        }
    }
    delete p;
}

}  // namespace

extern "C" {

mi_status mi_pipe_create(mi_ctx* c, const mi_pipe_config* cfg, mi_pipe** out)
{
    ENTER(c);
    if (!cfg || !out) return fail(c, MI_ERR_BAD_ARG, "null config / out");
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0) return fail(c, MI_ERR_BAD_ARG, "pipe needs a positive frame size");
    if ((cfg->width & 1) || (cfg->height & 1)) return fail(c, MI_ERR_BAD_ARG, "NV12 frames have even width and height");
    if ((long long)cfg->width * cfg->height > 0x7fffffffLL / 3) return fail(c, MI_ERR_UNSUPPORTED, "frame too large");
    if (cfg->op != MI_OP_EQUALIZE && cfg->op != MI_OP_CLAHE && cfg->op != MI_OP_CHANNELS) return fail(c, MI_ERR_BAD_ARG, "bad op");
    if (cfg->uv_mode != MI_UV_FILL128 && cfg->uv_mode != MI_UV_COPY) return fail(c, MI_ERR_BAD_ARG, "bad uv_mode");
    if (cfg->op == MI_OP_CLAHE && (cfg->tiles_x <= 0 || cfg->tiles_y <= 0)) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (cfg->uv_policy < MI_PIPE_UV_AUTO || cfg->uv_policy > MI_PIPE_UV_DEVICE) return fail(c, MI_ERR_BAD_ARG, "bad uv_policy");
    if (c->pipes_open > 0) return fail(c, MI_ERR_BUSY, "this context already has a pipe: one pipe per context (the pipe uses the context's scratch)");
    mi_pipe* p = new (std::nothrow) mi_pipe();
    if (!p) return fail(c, MI_ERR_OOM, "pipe allocation failed");
    p->c = c; p->cfg = *cfg;
    p->ybytes = (size_t)cfg->width * cfg->height; p->uvbytes = p->ybytes / 2;
    // Default depth by frame size (profiles/r04_t_*, r04_u_*): one thread that submits and waits on 4K frames is fastest with THREE in
    // flight (5.44-5.59 k frames/s; four: 4.74-4.81 k, two: 3.8-4.3 k -- a fourth frame only deepens the copy lanes' queues), while
    // 1080p frames, bound by their per-frame launch sequence, want more (six: 17.2 k, four: 15.8 k, three: 13.0 k through the pool)
    p->cfg.depth = cfg->depth > 0 ? std::max(2, std::min(16, cfg->depth)) : (p->ybytes + p->uvbytes >= ((size_t)8 << 20) ? 3 : 6);
    // the channel op needs chroma on the device; otherwise the UV half stays on the host unless asked for
    p->uv_dev = cfg->op == MI_OP_CHANNELS || cfg->uv_policy == MI_PIPE_UV_DEVICE;
    p->xfer_in = p->ybytes + ((p->uv_dev && (cfg->op == MI_OP_CHANNELS || cfg->uv_mode == MI_UV_COPY)) ? p->uvbytes : 0);
    p->xfer_out = p->ybytes + (p->uv_dev ? p->uvbytes : 0);
    auto bail = [&](mi_status st) { pipe_free(p); return st; };
    if (c->device >= kMaxDevices) { fail(c, MI_ERR_UNSUPPORTED, "pipe: device index too large"); return bail(MI_ERR_UNSUPPORTED); }
    p->private_streams = c->pipe_private_streams != 0;
    const char* wait_env = getenv("MI_LUMAEQ_PIPE_WAIT");
    if (wait_env) p->wait_mode = !strcmp(wait_env, "sync") ? 1 : (!strcmp(wait_env, "spin") ? 2 : 0);
    if (getenv("MI_LUMAEQ_PIPE_WAIT_SYNC")) p->wait_mode = 1;     // (round-3 spelling)
    // frames below 8 MiB (1080p: ~58 us per frame, 720p: ~43 us) are polled for a few frame times before the first sleep: one sleep is
    // ~55 us with the kernel's default timer slack -- a whole frame -- and cost 4-5 % of the throughput for 0.1-0.3 of a core
    // (profiles/r04_z_*).  The poll is BOUNDED (kSmallFrameSpinUs, about five 1080p frames): a wait that lasts longer is a stalled or
    // shared GPU, and from then on the thread sleeps between polls like a 4K worker instead of keeping a core busy for as long as the
    // stall lasts (round 4 polled for up to a second: ADVICE r4).  MI_LUMAEQ_PIPE_WAIT=backoff sleeps from the first poll at any size.
    constexpr int kSmallFrameSpinUs = 300;
    p->wait_spin_us = p->ybytes + p->uvbytes >= ((size_t)8 << 20) ? 0 : kSmallFrameSpinUs;
    if (wait_env && !strcmp(wait_env, "backoff")) p->wait_spin_us = 0;
    if (const char* e = getenv("MI_LUMAEQ_PIPE_WAIT_SPIN_US")) p->wait_spin_us = std::max(0, std::min(1000000, atoi(e)));
    if (p->private_streams) {
        PipeStreams own;
        for (int i = 0; i < 5; ++i) {
            hipError_t e = pipe_stream_create(own.all(i), i);
            if (e != hipSuccess) {
                for (int q = 0; q < 5; ++q) if (*own.all(q)) (void)hipStreamDestroy(*own.all(q));
                fail_hip(c, e, "hipStreamCreateWithFlags");
                return bail(MI_ERR_HIP);
            }
        }
        p->s_h2d[0] = own.h2d[0]; p->s_h2d[1] = own.h2d[1]; p->s_k = own.k; p->s_d2h[0] = own.d2h[0]; p->s_d2h[1] = own.d2h[1];
        p->n_copy = c->pipe_copy_streams > 1 ? 2 : 1;
    } else {
        std::lock_guard<std::mutex> lk(g_pipe_streams_mu);
        PipeStreams& ps = g_pipe_streams[c->device];
        if (ps.users == 0) {
            for (int i = 0; i < 5; ++i) {
                hipError_t e = pipe_stream_create(ps.all(i), i);
                if (e != hipSuccess) {
                    for (int q = 0; q < 5; ++q) if (*ps.all(q)) (void)hipStreamDestroy(*ps.all(q));
                    ps = PipeStreams{};
                    fail_hip(c, e, "hipStreamCreateWithPriority");
                    return bail(MI_ERR_HIP);
                }
            }
        }
        ++ps.users;
        g_pipe_users[c->device].fetch_add(1, std::memory_order_relaxed);
        p->s_h2d[0] = ps.h2d[0]; p->s_h2d[1] = ps.h2d[1]; p->s_k = ps.k; p->s_d2h[0] = ps.d2h[0]; p->s_d2h[1] = ps.d2h[1];
        p->n_copy = c->pipe_copy_streams > 1 ? 2 : 1;
    }
    p->slots.resize((size_t)p->cfg.depth);
    const size_t fbytes = p->ybytes + p->uvbytes;
    for (auto& sl : p->slots) {
        for (uint8_t** d : {&sl.d_in, &sl.d_out}) {
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, fbytes);
            if (e != hipSuccess) { (void)hipGetLastError(); fail(c, MI_ERR_OOM, "pipe: device frame allocation failed"); return bail(MI_ERR_OOM); }
            *d = (uint8_t*)q;
        }
        for (hipEvent_t* e : {&sl.ev_h2d, &sl.ev_k, &sl.ev_done}) {
            hipError_t r = hipEventCreateWithFlags(e, hipEventDisableTiming);
            if (r != hipSuccess) { fail_hip(c, r, "hipEventCreateWithFlags"); return bail(MI_ERR_HIP); }
        }
    }
    {
        void* q = nullptr;
        hipError_t e = hipHostMalloc(&q, 64 * sizeof(uint32_t), hipHostMallocDefault);
        if (e != hipSuccess) { fail_hip(c, e, "hipHostMalloc"); return bail(MI_ERR_HIP); }
        p->h_hard = (uint32_t*)q;
        memset(p->h_hard, 0, 64 * sizeof(uint32_t));
    }
    // warm-up: sizes the context's scratch for this shape and makes the runtime create the hardware queues / copy-engine
    // state behind all three streams now, not under the first real frame (26 ms were measured on a first frame otherwise)
    {
        PipeSlot& sl = p->slots[0];
        void* pin = nullptr;
        const size_t wb = std::min<size_t>(fbytes, (size_t)4 << 20);
        hipError_t e = hipHostMalloc(&pin, wb, hipHostMallocDefault);
        if (e != hipSuccess) { fail_hip(c, e, "hipHostMalloc"); return bail(MI_ERR_HIP); }
        memset(pin, 128, wb);
        auto step = [&](hipError_t r, const char* what) { if (r != hipSuccess && e == hipSuccess) { e = r; fail_hip(c, r, what); } };
        mi_status st = MI_OK;
        for (int lane = 0; lane < p->n_copy && st == MI_OK && e == hipSuccess; ++lane) {       // every stream a frame will travel on
            step(hipMemsetAsync(sl.d_in, 128, fbytes, p->s_h2d[lane]), "hipMemsetAsync");
            step(hipMemcpyAsync(sl.d_in, pin, wb, hipMemcpyHostToDevice, p->s_h2d[lane]), "hipMemcpyAsync");
            step(hipEventRecord(sl.ev_h2d, p->s_h2d[lane]), "hipEventRecord");
            step(hipStreamWaitEvent(p->s_k, sl.ev_h2d, 0), "hipStreamWaitEvent");
            st = e == hipSuccess ? pipe_run_op(p, sl) : MI_ERR_HIP;
            if (st == MI_OK) {
                step(hipEventRecord(sl.ev_k, p->s_k), "hipEventRecord");
                step(hipStreamWaitEvent(p->s_d2h[lane], sl.ev_k, 0), "hipStreamWaitEvent");
                step(hipMemcpyAsync(pin, sl.d_out, wb, hipMemcpyDeviceToHost, p->s_d2h[lane]), "hipMemcpyAsync");
                step(hipStreamSynchronize(p->s_d2h[lane]), "hipStreamSynchronize");
            }
        }
        (void)hipStreamSynchronize(p->s_k);
        (void)hipHostFree(pin);
        if (st) return bail(st);
        if (e != hipSuccess) return bail(MI_ERR_HIP);
    }
    ++c->pipes_open;
    *out = p;
    return MI_OK;
}

void mi_pipe_destroy(mi_pipe* p)
{
    if (!p) return;
    mi_ctx* c = p->c;
    std::unique_lock<std::mutex> lk(c->mu);
    c->pipe_pending -= (int)p->count;                             // frames never waited for: pipe_free() drains the streams
    if (c->pipes_open > 0) --c->pipes_open;
    pipe_free(p);
}

int mi_pipe_pending(const mi_pipe* p) { return p ? (int)p->count : 0; }
int mi_pipe_depth(const mi_pipe* p) { return p ? p->cfg.depth : 0; }

mi_status mi_pipe_submit(mi_pipe* p, const uint8_t* in, uint8_t* out, uint64_t tag)
{
    if (!p) return MI_ERR_BAD_ARG;
    mi_ctx* c = p->c;
    ENTER(c);
    if (!in || !out) return fail(c, MI_ERR_BAD_ARG, "null frame pointer");
    if (p->count == p->slots.size()) return fail(c, MI_ERR_BUSY, "pipe is full: call mi_pipe_wait first");
    PipeSlot& sl = p->slots[(p->head + p->count) % p->slots.size()];
    sl.in = in; sl.out = out; sl.tag = tag; sl.st = MI_OK;
    const size_t fbytes = p->ybytes + p->uvbytes;
    auto staging = [&](uint8_t** h, size_t bytes) -> mi_status {
        if (*h) return MI_OK;
        void* q = nullptr;
        hipError_t e = hipHostMalloc(&q, bytes, hipHostMallocDefault);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(c, MI_ERR_OOM, "pipe: pinned staging allocation failed"); }
        if (e != hipSuccess) return fail_hip(c, e, "hipHostMalloc");
        *h = (uint8_t*)q;
        return MI_OK;
    };
    // pinned memory is DMA'd as it is; anything else goes through the slot's pinned staging buffers.  Both are settled before
    // anything is enqueued: an allocation failure leaves nothing to undo.  The caller's ranges are entered into the pending-DMA table
    // BEFORE they are judged (pending_ranges.hpp: mi_host_unregister then answers MI_ERR_BUSY until mi_pipe_wait has retired the frame)
    // and leave it again on every exit that queued nothing on them.
    const uint64_t slot_id = (uint64_t)(&sl - p->slots.data());
    const size_t out_bytes = p->uv_dev ? fbytes : p->ybytes;
    g_pending_dma.add(p, slot_id, in, p->xfer_in);
    g_pending_dma.add(p, slot_id, out, out_bytes);
    struct PendingGuard {                                         // error exits: the drain guard below has waited for the streams by then
        mi_pipe* p; uint64_t id; bool keep = false;               // (declared first, so destroyed last)
        ~PendingGuard() { if (!keep) g_pending_dma.retire(p, id); }
    } pending{p, slot_id};
    const bool in_pinned = host_range_pinned(in, p->xfer_in, &c->pin_neg);
    const bool out_pinned = host_range_pinned(out, out_bytes, &c->pin_neg);
    mi_status st;
    if (!in_pinned && (st = staging(&sl.h_in, fbytes))) return st;
    if (!out_pinned && (st = staging(&sl.h_out, fbytes))) return st;
    for (bool direct : {in_pinned, out_pinned}) ++(direct ? c->planes_direct : c->planes_staged);
    sl.out_staged = !out_pinned;
    const uint8_t* h2d_src = in;
    if (!in_pinned) {
        CrewCall crew(c, p->xfer_in >= 4 * mi_host::CopyCrew::kMinBytes);
        crew.copy(sl.h_in, p->xfer_in, in, p->xfer_in, (int)std::min<size_t>(p->xfer_in, 0x7fffffff), 1);
        h2d_src = sl.h_in;
    }
    // from here on copies on caller memory are (about to be) in flight: every error exit waits for all three streams first
    std::unique_lock<std::mutex> submit_lk(g_pipe_submit_mu[c->device], std::defer_lock);
    if (!p->private_streams) submit_lk.lock();
    // consecutive frames ON THE DEVICE (whichever pipe they come from) travel on alternate copy streams
    // (with more than two pipes feeding one device both lanes of a direction end up with deep queues and single transfers were
    // measured 4x slower -- 2.4 k frames/s for four workers against 4.7 k on one lane: such a crowd shares lane 0)
    const bool crowd = !p->private_streams && g_pipe_users[c->device].load(std::memory_order_relaxed) > 2;
    sl.lane = (p->n_copy > 1 && !crowd) ? (int)(g_pipe_lane[c->device].fetch_add(1, std::memory_order_relaxed) & 1) : 0;
    hipStream_t s_h2d = p->s_h2d[sl.lane], s_d2h = p->s_d2h[sl.lane];
    StreamDrain drain(HipStreamSync{}, drain_counter(c));
    drain.watch(s_h2d); drain.watch(p->s_k); drain.watch(s_d2h);
    // a device-form call on a caller's stream since the last frame: the kernels share this context's scratch with it
    if (c->scratch_foreign) { HIPCHK(c, hipStreamWaitEvent(p->s_k, c->ev_scratch, 0)); c->scratch_foreign = false; }
    HIPCHK(c, hipMemcpyAsync(sl.d_in, h2d_src, p->xfer_in, hipMemcpyHostToDevice, s_h2d));
    HIPCHK(c, hipEventRecord(sl.ev_h2d, s_h2d));
    HIPCHK(c, hipStreamWaitEvent(p->s_k, sl.ev_h2d, 0));
    if ((st = pipe_run_op(p, sl))) return st;
    HIPCHK(c, hipEventRecord(sl.ev_k, p->s_k));
    HIPCHK(c, hipStreamWaitEvent(s_d2h, sl.ev_k, 0));
    HIPCHK(c, hipMemcpyAsync(sl.out_staged ? sl.h_out : out, sl.d_out, p->xfer_out, hipMemcpyDeviceToHost, s_d2h));
    HIPCHK(c, hipEventRecord(sl.ev_done, s_d2h));
    drain.done();                                                 // success: the frame stays in flight, that is the point of a pipe
    pending.keep = in_pinned || out_pinned;                       // a staged frame's transfers run on the slot's own buffers, not the caller's
    ++p->count; ++p->submitted; ++c->pipe_pending;
    return MI_OK;
}

mi_status mi_pipe_wait(mi_pipe* p, uint64_t* tag, uint8_t** out_frame)
{
    if (!p) return MI_ERR_BAD_ARG;
    mi_ctx* c = p->c;
    ENTER(c);
    if (p->count == 0) return fail(c, MI_ERR_BAD_ARG, "mi_pipe_wait: nothing pending");
    PipeSlot& sl = p->slots[p->head];
    const size_t slot = p->head;
    if (tag) *tag = sl.tag;
    if (out_frame) *out_frame = sl.out;
    // the host's share of the frame first: it overlaps whatever the copy engines are still doing
    if (!p->uv_dev) {
        if (p->cfg.uv_mode == MI_UV_FILL128) memset(sl.out + p->ybytes, 128, p->uvbytes);
        else if (sl.out != sl.in) memmove(sl.out + p->ybytes, sl.in + p->ybytes, p->uvbytes);
    }
    mi_status st = MI_OK;
    // Wait by polling hipEventQuery: hipEventSynchronize holds runtime locks while it blocks, and a second worker's enqueue calls on
    // the same device then queue up behind it (2 workers 4600 frames/s against 5400 for one).  Polling without a pause, as round 3
    // did, keeps one host core per worker 100 % busy for nothing (a streaming host also captures and encodes:
    // OpenCVequalHist.cpp:102-196), so after `wait_spin_us` the thread SLEEPS between polls, each time for a quarter of the time it
    // has waited so far (at most 200 us): the completion is noticed at most 25 % (+ the kernel's timer slack) late, whatever the
    // transfer takes, and the frames queued behind this one keep the engines busy meanwhile.  A frame with NOTHING queued behind it
    // is a caller waiting for its result (a paced stream: one 0.35 ms wait per 16.7 ms frame period): it is polled for up to 2 ms
    // before the first sleep, so its latency is what the transfer takes (sleeping cost it 0.09 ms of 0.38: profiles/r04_c_*).
    hipError_t e = hipSuccess;
    if (p->wait_mode == 1) e = hipEventSynchronize(sl.ev_done);
    else {
        const auto t_wait = std::chrono::steady_clock::now();
        for (;;) {
            e = hipEventQuery(sl.ev_done);
            if (e != hipErrorNotReady) break;
            const long long waited_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wait).count();
            if (p->wait_mode == 2 || waited_us < (p->count == 1 ? std::max(2000, p->wait_spin_us) : p->wait_spin_us)) { for (int k = 0; k < 16; ++k) __builtin_ia32_pause(); }
            else std::this_thread::sleep_for(std::chrono::microseconds(std::min<long long>(200, std::max<long long>(1, waited_us / 4))));
        }
    }
    e = MI_HOOKED(c, e);
    if (e != hipSuccess) {
        // the frame is lost, but nothing may stay in flight on its buffers and the slot must not be handed out while a DMA is pending
        st = fail_hip(c, e, "hipEventSynchronize(frame done)");
        for (hipStream_t s : {p->s_h2d[sl.lane], p->s_k, p->s_d2h[sl.lane]}) (void)hipStreamSynchronize(s);
        ++c->error_drains;
    } else if (sl.out_staged) {
        CrewCall crew(c, p->xfer_out >= 4 * mi_host::CopyCrew::kMinBytes);
        crew.copy(sl.out, p->xfer_out, sl.h_out, p->xfer_out, (int)std::min<size_t>(p->xfer_out, 0x7fffffff), 1);
    }
    // the slot is retired WHATEVER happened: one mi_pipe_wait = one tag gone, so the caller's own queue stays in step
    // (its transfers are over on both paths above -- the event was reached, or the streams were drained)
    g_pending_dma.retire(p, (uint64_t)slot);
    p->head = (p->head + 1) % p->slots.size();
    --p->count; ++p->completed; --c->pipe_pending;
    if (st) return st;
    const uint64_t hard = c->fused_stat_base[2] + p->h_hard[slot];
    if (p->cfg.op == MI_OP_EQUALIZE && hard > c->fused_seen_hard) {
        c->fused_seen_hard = hard;
        return fail(c, MI_ERR_HIP, "fused equalize kernel: a frame could not be repaired after an expired inter-workgroup wait; output invalid");
    }
    return MI_OK;
}

}  // extern "C"
