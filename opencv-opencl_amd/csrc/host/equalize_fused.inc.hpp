// equalize_fused.inc.hpp -- host side of the fused single-read equalizeHist kernel + equalize_dev dispatch
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

// ---- fused single-read path -----------------------------------------------------------------------------
// Layout of the hand-off block (uint32 words; zeroed once when (re)allocated -- every launch pair leaves it ready for the next):
//   [0..127] control words (kFused*: ticket counter, status, launch sequence number, finish arrivals, sticky statistics)
//   cnt[cap][32] | ready[cap][32] | ghist[cap][256] | lutpub[cap][128]
// plus a separate array of ticket stamps (d_fused_flags), one word per ticket of the largest launch seen.
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
    j.fault_inject = c->fused_fault_inject;
    j.timeout_ticks = c->fused_timeout_us > 0 ? (unsigned long long)c->fused_timeout_us * 100ull     // s_memrealtime ticks at 100 MHz
                                              : (unsigned long long)std::max(1, c->fused_timeout_ms) * 100000ull;
    j.U = 0;
    // UV as stand-alone 64 KiB tickets behind each frame's Y tickets: pure streaming work that fills the gaps while
    // other workgroups sit in their hand-off (measured 5 % faster than giving every Y ticket a share of the UV plane)
    if (uv && uv->bytes > 0) { j.uv = *uv; j.U = (int)((uv->bytes + 65535) / 65536); }
    const long long tickets = (long long)(j.T + j.U) * a.n_frames;
    // capacity-based layout so the regions never move between calls with different frame counts
    if (c->capturing && ((size_t)a.n_frames > c->fused_cap || (size_t)tickets * sizeof(uint32_t) > c->fused_flags_bytes))
        return fail(c, MI_ERR_UNSUPPORTED, "device scratch must grow inside a stream capture: size it with one eager call of this shape first");
    if ((size_t)a.n_frames > c->fused_cap) {
        size_t cap = std::max<size_t>(64, c->fused_cap);
        while (cap < (size_t)a.n_frames) cap *= 2;
        const size_t words = kFusedCtlWords + cap * (kFlagStride + kFlagStride + 256 + kLutPubWords);
        if (c->d_fused) {                                        // keep what the old block had counted
            uint64_t st4[4];
            mi_status st = fused_read_stats(c, s, st4);
            if (st) return st;
            for (int k = 0; k < 4; ++k) c->fused_stat_base[k] = st4[k];
        }
        mi_status st = grow_dev(c, &c->d_fused, &c->fused_bytes, words * sizeof(uint32_t));
        if (st) return st;
        c->fused_cap = cap;
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (words + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           c->d_fused, words);
        HIPCHK(c, hipGetLastError());
        if (c->d_fused_flags) {                                  // the new block restarts its epochs: old stamps must not match them
            const size_t nwords = c->fused_flags_bytes / sizeof(uint32_t);
            hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (nwords + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                               c->d_fused_flags, nwords);
            HIPCHK(c, hipGetLastError());
        }
    }
    {
        // ticket stamps: stale words are harmless (they carry other launches' epochs), so growth needs no zeroing --
        // except once, for epoch-0 safety of a fresh allocation (hipMalloc memory is not guaranteed to be zero)
        const size_t need = (size_t)tickets * sizeof(uint32_t);
        if (need > c->fused_flags_bytes) {
            mi_status st = grow_dev(c, &c->d_fused_flags, &c->fused_flags_bytes, std::max(need, (size_t)1 << 16));
            if (st) return st;
            const size_t nwords = c->fused_flags_bytes / sizeof(uint32_t);
            hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (nwords + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                               c->d_fused_flags, nwords);
            HIPCHK(c, hipGetLastError());
        }
    }
    const size_t cap = c->fused_cap;
    uint32_t* w = c->d_fused;
    j.ctl = w;
    j.cnt = w + kFusedCtlWords;
    j.ready = j.cnt + cap * kFlagStride;
    j.ghist = j.ready + cap * kFlagStride;
    j.lutpub = j.ghist + cap * 256;
    j.sflag = c->d_fused_flags;
    const long long grid = std::min<long long>(tickets, (long long)c->cu_count * c->fused_wgs_per_cu);
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
