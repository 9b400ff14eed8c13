// equalize_fused.inc.hpp -- host side of the fused single-read equalizeHist kernel + equalize_dev dispatch
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

// ---- fused single-read path -----------------------------------------------------------------------------
// Layout of the hand-off block (uint32 words; zeroed by zero_words_kernel when first used, re-laid-out, after a reported failure or
// in graph-replayable mode -- otherwise every launch leaves it clean for the next one):
//   [0..31] work counter (u64) | [32..63] status | cnt[nf][32] | ready[nf][32] | ghist[nf][256] | lutpub[nf][128]
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

mi_status equalize_fused_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const UVJob* uv)
{
    const long long ysz = (long long)a.width * a.height;
    const int vpt = fused_pick_vpt(c, a, uv);
    if (!vpt) return fail(c, MI_ERR_UNSUPPORTED, "fused path not applicable");
    FusedJob j{};
    j.src = a.src; j.dst = a.dst; j.src_frame = (long long)a.src_frame; j.dst_frame = (long long)a.dst_frame;
    j.nvec = ysz / 16; j.total = (int)ysz; j.n_frames = a.n_frames;
    const long long slice = (long long)kThreads * vpt;
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
    switch (vpt) {
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
