// equalize_fused.inc.hpp -- host side of the fused single-read equalizeHist kernel + equalize_dev dispatch
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

// ---- fused single-read path -----------------------------------------------------------------------------
// Layout of the hand-off block (uint32 words; zeroed once when (re)allocated -- every launch pair leaves it ready for the next):
//   [0..127] control words (kFused*: ticket counter, status, launch sequence number, finish arrivals, sticky statistics)
//   cnt[cap][32] | ready[cap][32] | ghist[cap][256] | lutpub[cap][128] | sflag[ticket_cap]
// (the ticket stamps are part of the block: one word per ticket of the largest launch the block is laid out for).
// Returns the slice size (16-byte vectors per thread) the fused kernel should run with, or 0 when the launch must take
// the three-kernel path.  The co-residency allowance (see g_fused_ctx_live) is the conservative 1/8 of the chip when
// other fused contexts exist on the device and half of the chip while this one is alone (an 8K frame is 405 tickets).
// Smaller slices for launches of one or two frames were measured and dropped: a single 4K frame takes 23.5-24.4 us with
// 80 / 64 / 32 KiB slices alike (fixed hand-off latency, not bandwidth), and two frames are slower in 32 KiB slices.
int fused_pick_vpt(const mi_ctx* c, const PlaneArgs& a, const UVJob* uv)
{
    if (!c->fused_mode || !c->fused_slot) return 0;
    if (a.src_step != (size_t)a.width || a.dst_step != (size_t)a.width) return 0;      // contiguous planes only
    const long long ysz = (long long)a.width * a.height;
    if (ysz % 16 != 0) return 0;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | a.src_frame | a.dst_frame) & 15) return 0;
    if (a.n_frames > (1 << 20)) return 0;
    (void)uv;
    const long long nvec = ysz / 16;
    auto tickets = [&](int vpt) { const long long s = (long long)kThreads * vpt; return (nvec + s - 1) / s; };
    const bool alone = c->device >= 0 && c->device < kMaxDevices && g_fused_ctx_live[c->device].load() <= 1;
    const long long limit = (long long)c->cu_count * c->fused_wgs_per_cu / (2 * (alone ? 1 : kMaxFusedCtxPerDevice));
    return tickets(c->fused_vpt) <= limit ? c->fused_vpt : 0;
}

bool fused_applicable(const mi_ctx* c, const PlaneArgs& a, const UVJob* uv) { return fused_pick_vpt(c, a, uv) != 0; }

// Sticky statistics of the hand-off block (device words) + what earlier blocks of this context had accumulated.
mi_status fused_read_stats(mi_ctx* c, hipStream_t s, uint64_t out[4])
{
    for (int k = 0; k < 4; ++k) out[k] = c->fused_stat_base[k];
    if (!c->d_fused) return MI_OK;
    if (!c->h_status) { void* q = nullptr; HIPCHK(c, hipHostMalloc(&q, 64, hipHostMallocDefault)); c->h_status = (uint32_t*)q; }
    HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_fused + kFusedStats, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int k = 0; k < 3; ++k) out[k] += c->h_status[k];
    if (c->h_status[3]) out[3] = c->h_status[3];
    return MI_OK;
}

// ---- demotion -----------------------------------------------------------------------------------------------
// A fused launch that loses the co-residency race (another process holds the compute units; the per-process allowance above
// cannot see it) stalls for its bound -- 50 ms -- and is repaired: correct bytes, three frames late at 60 fps.  A context that
// keeps meeting that fate gives the fused path up for a while: `fused_demote_after` repaired launches within a window of
// kFusedWindow fused launches route the following launches through the three-kernel path (no inter-workgroup dependency,
// nothing to stall) for `fused_reprobe_ms`; then ONE fused launch probes the GPU again, and a repair during the probe window
// demotes at once for twice as long (up to 64x).  The host learns about repairs from a word of pinned memory the finish
// kernel writes (FusedJob::host_repaired): no copy, no synchronisation, valid for every form including the stream-ordered ones.
// (The reference's accelerator path has no notion of a slow or failed device call at all: OpenCLequalHist.cpp:346-367.)
constexpr uint32_t kFusedWindow = 32;
constexpr int kMirrorWords = 16;                                 // one word per hand-off block generation (mod 16)

uint64_t fused_repaired_seen(const mi_ctx* c)
{
    uint64_t n = c->fused_repaired_base;
    for (int k = 0; k < kMirrorWords; ++k) n += __atomic_load_n(c->h_mirror + k, __ATOMIC_RELAXED);
    return n;
}

// unrecoverable frames, as mirrored by the finish kernel (exact once the stream the launch ran on has been waited for)
uint64_t fused_hard_seen(const mi_ctx* c)
{
    uint64_t n = c->fused_hard_base;
    for (int k = 0; k < kMirrorWords; ++k) n += __atomic_load_n(c->h_mirror + kMirrorWords + k, __ATOMIC_RELAXED);
    return n;
}

bool fused_admit(mi_ctx* c)
{
    if (c->fused_demote_after <= 0) return true;
    if (c->capturing) return !c->fused_demoted;                   // a capture records whichever path is current; no bookkeeping
    const uint64_t repaired = fused_repaired_seen(c);
    const auto now = std::chrono::steady_clock::now();
    if (c->fused_demoted) {
        if (now < c->fused_reprobe_at) return false;
        c->fused_demoted = false; c->fused_probing = true;       // the period is over: this launch probes
        c->fused_window_start_repaired = repaired; c->fused_window_launches = 0;
    }
    const uint64_t limit = c->fused_probing ? 1u : (uint64_t)c->fused_demote_after;
    if (repaired - c->fused_window_start_repaired >= limit) {
        c->fused_reprobe_ms_now = c->fused_probing ? std::min<long long>(2LL * c->fused_reprobe_ms_now, 64LL * c->fused_reprobe_ms) : c->fused_reprobe_ms;
        c->fused_demoted = true; c->fused_probing = false;
        ++c->fused_demotions;
        c->fused_reprobe_at = now + std::chrono::milliseconds(c->fused_reprobe_ms_now);
        return false;
    }
    if (++c->fused_window_launches >= kFusedWindow) {            // a window without enough repairs: start the next one
        c->fused_window_start_repaired = repaired; c->fused_window_launches = 0;
        if (c->fused_probing) { c->fused_probing = false; c->fused_reprobe_ms_now = c->fused_reprobe_ms; }
    }
    return true;
}

mi_status equalize_fused_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const UVJob* uv)
{
    const long long ysz = (long long)a.width * a.height;
    const int vpt = fused_pick_vpt(c, a, uv);
    if (!vpt) return fail(c, MI_ERR_UNSUPPORTED, "fused path not applicable");
    FusedJob j{};
    j.src = a.src; j.dst = a.dst; j.src_frame = (long long)a.src_frame; j.dst_frame = (long long)a.dst_frame;
    j.nvec = ysz / 16; j.total = (int)ysz; j.n_frames = a.n_frames;
    const long long slice = (long long)kThreads * vpt;
    j.slice_vecs = (int)slice;
    j.T = (int)((j.nvec + slice - 1) / slice);
    j.acquire = c->fused_acquire;
    j.timeout_ticks = (unsigned long long)std::max(1, c->fused_timeout_ms) * 100000ull;       // s_memrealtime ticks at 100 MHz
#ifdef MI_TEST_HOOKS
    j.fault_inject = c->fused_fault_inject;
    if (c->fused_timeout_us > 0) j.timeout_ticks = (unsigned long long)c->fused_timeout_us * 100ull;
#endif
    j.U = 0;
    // UV as stand-alone 64 KiB tickets behind each frame's Y tickets: pure streaming work that fills the gaps while
    // other workgroups sit in their hand-off (measured 5 % faster than giving every Y ticket a share of the UV plane)
    if (uv && uv->bytes > 0) { j.uv = *uv; j.U = (int)((uv->bytes + 65535) / 65536); }
    const long long tickets = (long long)(j.T + j.U) * a.n_frames;
    // A fused kernel whose finish kernel never followed (the launch in between failed) left the ticket counter and the sequence
    // number as they were: start from a clean block instead of trusting them.
    if (c->fused_pair_open) {
        if (c->capturing) return fail(c, MI_ERR_UNSUPPORTED, "the fused path must be reset by an eager call after a failed launch");
        HIPCHK(c, hipDeviceSynchronize());
        if (c->d_fused) {
            uint64_t st4[4];                                     // the sticky statistics live in the block: carry them over
            mi_status st = fused_read_stats(c, s, st4);
            if (st) return st;
            for (int k = 0; k < 4; ++k) c->fused_stat_base[k] = st4[k];
            const size_t words = c->fused_bytes / sizeof(uint32_t);
            hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (words + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                               c->d_fused, words);
            HIPCHK(c, hipGetLastError());
            for (int half = 0; half < 2; ++half) {              // the block restarts from zero: what its mirror words held moves into the bases
                uint32_t* mw = c->h_mirror + half * kMirrorWords + (c->fused_generation % kMirrorWords);
                (half ? c->fused_hard_base : c->fused_repaired_base) += __atomic_load_n(mw, __ATOMIC_RELAXED);
                __atomic_store_n(mw, 0u, __ATOMIC_RELAXED);
            }
        }
        c->fused_pair_open = false;
    }
    // One allocation holds the control words, the per-frame hand-off regions AND the ticket stamps, laid out by capacity so that
    // the regions never move between calls with different frame counts.  When either capacity is exceeded the whole block is
    // replaced: stamps, epochs and checksums always share a lifetime (a block that a captured graph still replays into keeps
    // its own stamps -- a new block's epochs can never meet an old block's stamps).
    if ((size_t)a.n_frames > c->fused_cap || (size_t)tickets > c->fused_ticket_cap) {
        if (c->capturing)
            return fail(c, MI_ERR_UNSUPPORTED, "device scratch must grow inside a stream capture: size it with one eager call of this shape first");
        size_t cap = std::max<size_t>(64, c->fused_cap);
        while (cap < (size_t)a.n_frames) cap *= 2;
        size_t tcap = std::max<size_t>((size_t)1 << 14, c->fused_ticket_cap);
        while (tcap < (size_t)tickets) tcap *= 2;
        const size_t words = kFusedCtlWords + cap * (kFlagStride + kFlagStride + 256 + kLutPubWords) + tcap;
        if (c->d_fused) {                                        // keep what the old block had counted
            uint64_t st4[4];
            mi_status st = fused_read_stats(c, s, st4);
            if (st) return st;
            for (int k = 0; k < 4; ++k) c->fused_stat_base[k] = st4[k];
        }
        c->fused_cap = 0; c->fused_ticket_cap = 0;
        mi_status st = grow_dev(c, &c->d_fused, &c->fused_bytes, std::max(words * sizeof(uint32_t), c->fused_bytes + 4));
        if (st) return st;
        c->fused_cap = cap; c->fused_ticket_cap = tcap;
        // hipMalloc memory is not guaranteed to be zero: epoch arithmetic starts from a clean block
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (words + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           c->d_fused, words);
        HIPCHK(c, hipGetLastError());
        // this block's word of the host mirror (repaired launches, written by the finish kernel)
        c->fused_generation += 1;
        uint32_t* mw = c->h_mirror + (c->fused_generation % kMirrorWords);
        c->fused_repaired_base += __atomic_load_n(mw, __ATOMIC_RELAXED);
        __atomic_store_n(mw, 0u, __ATOMIC_RELAXED);
        c->fused_hard_base += __atomic_load_n(mw + kMirrorWords, __ATOMIC_RELAXED);
        __atomic_store_n(mw + kMirrorWords, 0u, __ATOMIC_RELAXED);
    }
    const size_t cap = c->fused_cap;
    uint32_t* w = c->d_fused;
    j.ctl = w;
    j.cnt = w + kFusedCtlWords;
    j.ready = j.cnt + cap * kFlagStride;
    j.ghist = j.ready + cap * kFlagStride;
    j.lutpub = j.ghist + cap * 256;
    j.sflag = j.lutpub + cap * kLutPubWords;
    j.host_repaired = c->h_mirror + (c->fused_generation % kMirrorWords);
    j.host_hard = c->fused_hard_word ? c->fused_hard_word : j.host_repaired + kMirrorWords;
    const long long grid = std::min<long long>(tickets, (long long)c->cu_count * c->fused_wgs_per_cu);
    if (!c->capturing) c->fused_pair_open = true;
    switch (vpt) {
        case 8:  LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<8>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        case 20: LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<20>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        case 24: LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<24>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        case 16: LAUNCH(c, s, MI_K_FUSED, equalize_fused_kernel<16>, dim3((unsigned)grid), dim3(kThreads), 0, j); break;
        default: return fail(c, MI_ERR_BAD_ARG, "bad fused_vpt");
    }
    // always: housekeeping in the normal case, stamp-driven repair when a bounded wait expired (kernels/equalize_fused.hip.h)
    const int fin_grid = (int)std::min<long long>(a.n_frames, (long long)c->cu_count * 4);
    LAUNCH(c, s, MI_K_FUSED_FINISH, fused_finish_kernel, dim3((unsigned)fin_grid), dim3(kThreads), 0, j);
    c->fused_pair_open = false;
    return MI_OK;
}

// Up to eight frames (a stream's frame, a cv::Mat call; up to sixteen of 1080p or less): histogram + LUT in one launch whose last
// workgroup writes the LUT, then the apply kernel.  No inter-workgroup waits, so no finish kernel and nothing to repair: 17 us per
// 4K frame against 22.5 us for the fused pair, whose single read cannot pay for its hand-off latency on so little data.
mi_status equalize_two_kernel_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const UVJob* uv)
{
    const size_t need = (size_t)a.n_frames * (256 + 1) * sizeof(uint32_t);
    if (need > c->ghist_bytes) {
        mi_status st = grow_dev(c, &c->d_ghist, &c->ghist_bytes, std::max<size_t>(need, 64 * 257 * sizeof(uint32_t)));
        if (st) return st;
        HIPCHK(c, hipMemsetAsync(c->d_ghist, 0, c->ghist_bytes, s));      // once: every launch leaves the scratch zeroed
    }
    mi_status st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)a.n_frames * 256);
    if (st) return st;
    PlaneArgs b = a;
    b.dst = nullptr;
    const PlaneBatch p = make_plane(b);
    // workgroups per frame: every one of them ends in up to 256 global atomics on the frame's histogram, so FEWER than the
    // streaming kernels' ~8 per CU: about two per CU in all, at least 32 KiB each, at most 128 per frame (probe of round 3: four 4K
    // frames 34.5 us with 256 per frame, 28.6 with 128; four 1080p frames 17.3 us with 126, 14.5 with 64)
    const long long px = (long long)a.width * a.height;
    long long Bq = std::min<long long>({(long long)c->cu_count * 2 / a.n_frames, 128LL, std::max<long long>(1, px / 32768)});
    if (p.rows > 1) Bq = std::min<long long>(Bq, p.rows);
    const int B = (int)std::max<long long>(1, Bq);
    uint32_t* cnt = c->d_ghist + (size_t)a.n_frames * 256;
    LAUNCH(c, s, MI_K_HIST, hist_lut_kernel, dim3(B, a.n_frames), dim3(kHistThreads), 0, p, c->d_ghist, cnt,
           (int)((long long)a.width * a.height), c->d_luts);
    return launch_apply(c, s, a, 0, a.n_frames, c->d_luts, uv);
}

mi_status equalize_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const UVJob* uv)
{
    // measured (profiles/r03_l_single_frame.txt, us per call, two-kernel / fused pair / three-kernel): 4K 1 frame 17.0 / 22.5 / 24.5,
    // 2 frames 20.6 / 28.2 / 29.5, 4 frames 29.3 / 38.9 / 37.1, 8 frames 45.5 / 52.2 / 53.2, 16 frames 87.2 / 87.0 / 90.5;
    // 1080p 4 frames 15.3 / 22.5 / 19.7, 8 frames 21.6 / 28.2 / 25.1, 16 frames 29.5 / 39.3 / 35.7
    const long long px = (long long)a.width * a.height;
    const int k2 = c->two_kernel_max_frames;
    if (k2 > 0 && (a.n_frames <= k2 || (a.n_frames <= 2 * k2 && px <= 1920LL * 1088))) return equalize_two_kernel_dev(c, s, a, uv);
    if (fused_applicable(c, a, uv) && fused_admit(c)) return equalize_fused_dev(c, s, a, uv);
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
