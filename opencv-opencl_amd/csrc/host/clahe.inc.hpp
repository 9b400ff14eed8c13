// clahe.inc.hpp -- CLAHE geometry and launch sequence (8-bit)
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

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
    g->contract = c->clahe_fp_contract;
    return MI_OK;
}

// tiles per workgroup of the batched tile-histogram pass, by tile size.  Measured (profiles/r02_p_clahe_ab_tiles_per_wg.txt, tile
// histograms per launch with 1 / 2 / 4 tiles per workgroup): 640x360 8x8 261 / 202 / 201 us, 1280x720 8x8 185 / 169 / 173,
// 1920x1080 8x8 135 / 138 / 144, 3840x2160 16x16 132 / 135 / 142, 3840x2160 8x8 125 / 134 / 134 -- two tiles per workgroup pay
// below ~3 vectors per lane of a 512-thread workgroup (tiles of less than ~24 K pixels), never more than two, and not above.
int clahe_auto_tiles_per_wg(const ClaheGeom& g)
{
    return (long long)g.tile_w * g.tile_h < 3LL * 16 * 512 ? 2 : 1;
}

mi_status launch_tile_luts(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const ClaheGeom& g, int f0, int nf, uint8_t* d_luts_out)
{
    const int tiles = g.tiles_x * g.tiles_y;
    // Splits per tile.  Splitting costs a third launch (partials -> tile_lut_kernel) and pays only for LARGE tiles on few frames, and then
    // only up to about one workgroup per CU: one 4K frame 8x8 takes 29.8 us with 32 splits (the round-2 rule: ~8 workgroups per CU),
    // 21.9 us with none (64 workgroups on 256 CUs!) and 20.5 us with 4; 1080p 8x8 (tiles of 32 K pixels) is fastest unsplit at every
    // batch size (four frames: 18.7 us against 27.1) -- profiles/r03_m_clahe_splits.txt.
    const long long tile_px = (long long)g.tile_w * g.tile_h;
    const long long want = tile_px >= 65536 ? (long long)c->cu_count / ((long long)tiles * nf) : 1;
    int S = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, g.tile_h / 8), 64LL}));
    mi_status st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * tiles * S * 256 * sizeof(uint32_t));
    if (st) return st;
    const uint8_t* src = a.src + (size_t)f0 * a.src_frame;
    // grid.y = tiles, grid.z = frames
    if (tiles > kMaxGridY) return fail(c, MI_ERR_UNSUPPORTED, "more than 65535 tiles per frame");
    // one workgroup per tile: the LUT is computed where the histogram was built, no partials, no second launch
    uint8_t* direct = S == 1 ? d_luts_out : nullptr;
    // XCD-aware tile order only where the round-robin placement is predictable: one workgroup per tile, tile count a multiple of 8
    const int xcd_map = (c->clahe_xcd_map && S == 1 && tiles % 8 == 0) ? 1 : 0;
    // 512 threads per workgroup (32 waves per CU sharing four 32 KiB histograms) measured 6-8 % faster than 256 (20 waves) and
    // than 1024 (32 waves, two histograms) at 4K and 1080p, 64-frame batches: profiles/r02_c_clahe_ab.txt
    // small tiles in large batches are bound by the rate at which workgroups are dispatched (~4 ns apiece): K tiles per workgroup
    int K = 1;
    if (direct && c->clahe_hist_threads == 512) {
        const int want_k = c->clahe_tiles_per_wg > 0 ? c->clahe_tiles_per_wg : clahe_auto_tiles_per_wg(g);
        const int run = xcd_map ? tiles / 8 : tiles;             // tiles a workgroup may take consecutively
        for (K = std::max(1, std::min(want_k, 8)); K > 1 && run % K != 0; --K) {}
        // the batch must still fill the chip with workgroups
        while (K > 1 && (long long)tiles / K * nf < (long long)c->cu_count * 8) K >>= 1;
        if (K > 1 && run % K != 0) K = 1;
    }
    if (K > 1) {
        LAUNCH(c, s, MI_K_TILE_HIST, tile_hist_multi_kernel, dim3(1, tiles / K, nf), dim3(kTileMultiThreads), 0,
               src, (long long)a.src_step, (long long)a.src_frame, g, direct, tiles, K, xcd_map);
        return MI_OK;
    }
    if (c->clahe_hist_threads == 512)
        LAUNCH(c, s, MI_K_TILE_HIST, tile_hist_kernel<512>, dim3(S, tiles, nf), dim3(512), 0,
               src, (long long)a.src_step, (long long)a.src_frame, g, c->d_partial, direct, xcd_map);
    else
        LAUNCH(c, s, MI_K_TILE_HIST, tile_hist_kernel<kThreads>, dim3(S, tiles, nf), dim3(kThreads), 0,
               src, (long long)a.src_step, (long long)a.src_frame, g, c->d_partial, direct, xcd_map);
    if (!direct)
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
        int groups = std::min(ngroups, kThreads);
        // float tables hold kMaxPairsLdsF32 pairs.  A wider grid still gets them when the frame is cut into column segments narrow
        // enough to touch few pairs: a span of G groups touches at most floor(16 G / tile_w) + 2 pairs (+ 1 of slack for the float
        // expression that decides them).  Segment tables of 9 pairs (36 KiB, four workgroups per CU like the 8 x 8 case) measured best
        // (profiles/r02_s_clahe_ab_seg_pairs.txt: interpolation of 4K 16x16 327 -> 284 us, 1080p 16x16 317 -> 294, 4K 32x32 338 -> 299
        // against the uchar-quad tables; 15 pairs = 60 KiB = two workgroups per CU is SLOWER than the quads); below ~40 groups per
        // segment (tiles narrower than ~110 pixels: 720p 16x16 365 -> 386) the quads stay.
        bool seg_tables = false;
        const int seg_cap = std::max(4, std::min(c->clahe_seg_pairs, kMaxPairsLdsF32));      // pairs per segment table (4 KiB of LDS each)
        if (npairs > kMaxPairsLdsF32 && c->clahe_float_tables) {
            const int gmax = (int)(((long long)(seg_cap - 3) * g.tile_w) / kInterpPx);
            if (gmax >= 40) {                                 // equal segments: a short last one would leave most of its workgroups idle
                const int nseg = (ngroups + std::min(groups, gmax) - 1) / std::min(groups, gmax);
                groups = (ngroups + nseg - 1) / nseg;
                seg_tables = true;
            }
        }
        const int segs = (ngroups + groups - 1) / groups;
        const int bands = g.tiles_y + 1;
        long long want = ((long long)c->cu_count * 8 + (long long)bands * nf * segs - 1) / ((long long)bands * nf * segs);
        const int rows_per_band = g.tile_h + 2 * kBandMargin;
        int subs = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, rows_per_band / 8), 64LL}));
        if ((long long)bands * subs > 0x7fffffffLL || segs > kMaxGridY) return fail(c, MI_ERR_UNSUPPORTED, "image too wide");
        const dim3 grid(bands * subs, nf, segs);
        if ((npairs <= kMaxPairsLdsF32 || seg_tables) && c->clahe_float_tables) {
            const int cap = seg_tables ? seg_cap : kMaxPairsLdsF32;
            const size_t lds = (size_t)std::min(npairs, cap) * 256 * 4 * sizeof(float);
            if (g.contract) LAUNCH(c, s, MI_K_CLAHE_INTERP, (clahe_interp_kernel<true, true>), grid, dim3(kThreads), lds, p, g, d_luts, subs, groups, uv, cap);
            else            LAUNCH(c, s, MI_K_CLAHE_INTERP, (clahe_interp_kernel<true, false>), grid, dim3(kThreads), lds, p, g, d_luts, subs, groups, uv, cap);
        } else {
            const size_t lds = (size_t)npairs * 256 * sizeof(uint32_t);
            if (g.contract) LAUNCH(c, s, MI_K_CLAHE_INTERP, (clahe_interp_kernel<false, true>), grid, dim3(kThreads), lds, p, g, d_luts, subs, groups, uv, kMaxPairsLds + 1);
            else            LAUNCH(c, s, MI_K_CLAHE_INTERP, (clahe_interp_kernel<false, false>), grid, dim3(kThreads), lds, p, g, d_luts, subs, groups, uv, kMaxPairsLds + 1);
        }
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

// ---- single-read CLAHE by cells (kernels/clahe_cell.hip.h, docs/experiments.md R5.4) ------------------------------------------
// The cell kernel only runs on REGULAR geometries: no padding, tile_w a multiple of 32, planes and pitches on 16-byte boundaries, and
// the band / pair boundaries -- decided by the reference's own float expressions, evaluated here exactly as the kernels evaluate them
// (this file is built with -ffp-contract=off) -- at the same offset inside every tile: columns at tile_w / 2, rows at `ysplit`.
static inline int host_tile_index(int p, float inv, int contract)
{
    const float v = contract ? fmaf((float)p, inv, -0.5f) : (float)p * inv - 0.5f;
    const int i = (int)v;
    return i - ((float)i > v);                                   // cvFloor
}

bool clahe_cell_geometry(const mi_ctx* c, const PlaneArgs& a, const ClaheGeom& g, CellGeom* cg)
{
    if (a.width % g.tiles_x != 0 || a.height % g.tiles_y != 0) return false;               // padded: reflection inside the histogram pass
    if (g.tile_w % 32 != 0 || g.tiles_x > 64 || g.tiles_y > 64) return false;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | a.src_step | a.dst_step | a.src_frame | a.dst_frame) & 15) return false;
    const int cw = g.tile_w / 2;
    cg->cells_x = 2 * g.tiles_x; cg->cells_y = 2 * g.tiles_y;
    cg->groups = cw / 16;
    if (cg->groups < 1 || cg->groups > kThreads) return false;
    cg->phases = kThreads / cg->groups;
    for (int tx = 0; tx < g.tiles_x; ++tx) {                     // columns [tx*tw, tx*tw + cw) -> pair tx, the rest -> pair tx + 1
        const int x = tx * g.tile_w;
        if (host_tile_index(x, g.inv_tw, g.contract) != tx - 1 || host_tile_index(x + cw - 1, g.inv_tw, g.contract) != tx - 1 ||
            host_tile_index(x + cw, g.inv_tw, g.contract) != tx || host_tile_index(x + g.tile_w - 1, g.inv_tw, g.contract) != tx) return false;
    }
    int ys = 0;
    while (ys < g.tile_h && host_tile_index(ys, g.inv_th, g.contract) < 0) ++ys;
    if (ys <= 0 || ys >= g.tile_h) return false;
    for (int ty = 0; ty < g.tiles_y; ++ty) {                     // rows [ty*th, ty*th + ys) -> band ty, the rest -> band ty + 1
        const int y = ty * g.tile_h;
        if (host_tile_index(y, g.inv_th, g.contract) != ty - 1 || host_tile_index(y + ys - 1, g.inv_th, g.contract) != ty - 1 ||
            host_tile_index(y + ys, g.inv_th, g.contract) != ty || host_tile_index(y + g.tile_h - 1, g.inv_th, g.contract) != ty) return false;
    }
    cg->ysplit = ys;
    cg->variant = c->clahe_cell_variant;
    if (std::max(ys, g.tile_h - ys) > cg->phases * kCellVPT) return false;                  // a lane keeps kCellVPT rows
    if ((long long)(std::max(ys, g.tile_h - ys) - 1) * (long long)std::max(a.src_step, a.dst_step) + cw > 0x3fffffffLL) return false;
    if ((long long)cg->cells_x * cg->cells_y > 0x7fffffffLL) return false;
    (void)c;
    return true;
}

mi_status launch_cells(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const ClaheGeom& g, const CellGeom& cg, int f0, int nf,
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
    const int cells = cg.cells_x * cg.cells_y;
    mi_status st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * cells * 256 * sizeof(uint32_t));
    if (st) return st;
    if (g.contract) LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_cell_kernel<true>, dim3(cells, nf), dim3(kThreads), 0, p, g, cg, d_luts, c->d_partial, uv);
    else            LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_cell_kernel<false>, dim3(cells, nf), dim3(kThreads), 0, p, g, cg, d_luts, c->d_partial, uv);
    return MI_OK;
}

// ---- the fused cell kernel: hand-off block, launch pair, statistics, admission ------------------------------------------------
constexpr int kCellWgsPerCu = 2;                                // 512-thread workgroups: two per CU (launch bounds), each with two cells in flight
constexpr int kCellMirrorBase = 32;                              // words of mi_ctx::h_mirror that belong to the cell path (one per block generation mod 16)

uint64_t cell_repaired_seen(const mi_ctx* c)
{
    uint64_t n = c->cell_repaired_base;
    for (int k = 0; k < 16; ++k) n += __atomic_load_n(c->h_mirror + kCellMirrorBase + k, __ATOMIC_RELAXED);
    return n;
}

// sticky statistics of the hand-off block (device words) + what earlier blocks of this context had accumulated
mi_status cell_read_stats(mi_ctx* c, hipStream_t s, uint64_t out[3])
{
    for (int k = 0; k < 3; ++k) out[k] = c->cell_stat_base[k];
    if (!c->d_cell) return MI_OK;
    if (!c->h_status) { void* q = nullptr; HIPCHK(c, hipHostMalloc(&q, 64, hipHostMallocDefault)); c->h_status = (uint32_t*)q; }
    HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_cell + kCellStats, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int k = 0; k < 2; ++k) out[k] += c->h_status[k];
    if (c->h_status[2]) out[2] = c->h_status[2];
    return MI_OK;
}

// Co-residency: the kernel needs 8 * cells_x workgroups running at once (kernels/clahe_cell.hip.h).  The allowance is the fused
// equalizeHist kernel's (g_fused_ctx_live): half of the chip's guaranteed workgroup slots while this context is alone on the device,
// 1 / (2 * kMaxFusedCtxPerDevice) of them otherwise.
bool clahe_cell_applicable(const mi_ctx* c, const PlaneArgs& a, const ClaheGeom& g, CellGeom* cg)
{
    if (c->clahe_single_read != 1 || !c->fused_mode || !c->fused_slot) return false;
    if (a.n_frames < c->clahe_single_read_min_frames || a.n_frames > (1 << 20)) return false;
    if (!clahe_cell_geometry(c, a, g, cg)) return false;
    if (std::max(cg->ysplit, g.tile_h - cg->ysplit) > (kCell2Threads / cg->groups) * kCell2VPT) return false;     // a lane of the fused kernel keeps kCell2VPT rows per cell
    const bool alone = c->device >= 0 && c->device < kMaxDevices && g_fused_ctx_live[c->device].load() <= 1;
    const long long limit = (long long)c->cu_count * kCellWgsPerCu / (2 * (alone ? 1 : kMaxFusedCtxPerDevice));
    return (long long)kCellQueues * cg->cells_x <= limit;
}

// One repaired launch (a 50 ms stall) gives the path up for `fused_reprobe_ms`; the launch after that probes, a repaired probe doubles
// the period (up to 64 x), kFusedWindow clean launches end the probation.  Same rule as the fused equalizeHist path, one strike.
bool clahe_cell_admit(mi_ctx* c)
{
    if (c->fused_demote_after <= 0) return true;
    if (c->capturing) return !c->cell_demoted;
    const uint64_t repaired = cell_repaired_seen(c);
    const auto now = std::chrono::steady_clock::now();
    if (c->cell_demoted) {
        if (now < c->cell_reprobe_at) return false;
        c->cell_demoted = false; c->cell_probing = true; c->cell_clean_launches = 0;
    }
    if (repaired > c->cell_repaired_last) {
        c->cell_repaired_last = repaired;
        c->cell_reprobe_ms_now = c->cell_probing ? (int)std::min<long long>(2LL * c->cell_reprobe_ms_now, 64LL * c->fused_reprobe_ms) : c->fused_reprobe_ms;
        c->cell_demoted = true; c->cell_probing = false;
        ++c->cell_demotions;
        c->cell_reprobe_at = now + std::chrono::milliseconds(c->cell_reprobe_ms_now);
        return false;
    }
    if (c->cell_probing && ++c->cell_clean_launches >= kFusedWindow) { c->cell_probing = false; c->cell_reprobe_ms_now = c->fused_reprobe_ms; }
    return true;
}

mi_status clahe_cell_fused_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, const ClaheGeom& g, const CellGeom& cg, const UVJob* uv)
{
    const int tiles = g.tiles_x * g.tiles_y, cells = cg.cells_x * cg.cells_y;
    const size_t need_tiles = (size_t)a.n_frames * tiles, need_cells = (size_t)a.n_frames * cells;
    mi_status st = grow_dev(c, &c->d_luts, &c->luts_bytes, need_tiles * 256);       // plain LUT bytes: scratch of the repair pass
    if (st) return st;
    auto zero_block = [&]() -> mi_status {
        const size_t words = c->cell_bytes / sizeof(uint32_t);
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(1024, (words + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, c->d_cell, words);
        HIPCHK(c, hipGetLastError());
        return MI_OK;
    };
    auto retire_mirror = [&]() {                                  // the block restarts from zero: what its mirror word held moves into the base
        uint32_t* mw = c->h_mirror + kCellMirrorBase + (c->cell_generation % 16);
        c->cell_repaired_base += __atomic_load_n(mw, __ATOMIC_RELAXED);
        __atomic_store_n(mw, 0u, __ATOMIC_RELAXED);
    };
    if (c->cell_pair_open) {                                      // a fused kernel whose finish kernel never followed: counters in an unknown state
        if (c->capturing) return fail(c, MI_ERR_UNSUPPORTED, "the fused CLAHE path must be reset by an eager call after a failed launch");
        HIPCHK(c, hipDeviceSynchronize());
        if (c->d_cell) {
            uint64_t st3[3];
            if ((st = cell_read_stats(c, s, st3))) return st;
            for (int k = 0; k < 3; ++k) c->cell_stat_base[k] = st3[k];
            if ((st = zero_block())) return st;
            retire_mirror();
        }
        c->cell_pair_open = false;
    }
    if (need_tiles > c->cell_cap_tiles || need_cells > c->cell_cap_cells) {
        if (c->capturing)
            return fail(c, MI_ERR_UNSUPPORTED, "device scratch must grow inside a stream capture: size it with one eager call of this shape first");
        size_t capt = std::max<size_t>(1024, c->cell_cap_tiles), capc = std::max<size_t>(4096, c->cell_cap_cells);
        while (capt < need_tiles) capt *= 2;
        while (capc < need_cells) capc *= 2;
        const size_t words = kCellCtlWords + capt * (2 + kCellLutWords + 256) + capc;
        if (c->d_cell) {
            uint64_t st3[3];
            if ((st = cell_read_stats(c, s, st3))) return st;
            for (int k = 0; k < 3; ++k) c->cell_stat_base[k] = st3[k];
        }
        c->cell_cap_tiles = 0; c->cell_cap_cells = 0;
        st = grow_dev(c, &c->d_cell, &c->cell_bytes, std::max(words * sizeof(uint32_t), c->cell_bytes + 4));
        if (st) return st;
        c->cell_cap_tiles = capt; c->cell_cap_cells = capc;
        if ((st = zero_block())) return st;                       // hipMalloc memory is not guaranteed to be zero: epoch arithmetic starts from a clean block
        c->cell_generation += 1;
        retire_mirror();
    }
    CellJob j{};
    j.p.src = a.src; j.p.dst = a.dst;
    j.p.src_step = (long long)a.src_step; j.p.dst_step = (long long)a.dst_step;
    j.p.src_frame = (long long)a.src_frame; j.p.dst_frame = (long long)a.dst_frame;
    j.p.row_bytes = a.width; j.p.rows = a.height;
    j.g = g; j.cg = cg; j.cg.variant = c->clahe_cell_variant & (64 | 128);       // measurement switches (tools/clahe_cell_ablate.py), 0 in production
    if (j.cg.variant & 64) {                                      // ... which take the tile LUTs from a tile-histogram pass
        if ((st = launch_tile_luts(c, s, a, g, 0, a.n_frames, c->d_luts))) return st;
    }
    if (uv && uv->bytes > 0) j.uv = *uv;
    j.n_frames = a.n_frames;
    j.rows_per_queue = (cg.cells_y + kCellQueues - 1) / kCellQueues;
    j.acquire = 0;                                                // no fence anywhere: see kernels/clahe_cell.hip.h, step 4
    j.timeout_ticks = (unsigned long long)std::max(1, c->fused_timeout_ms) * 100000ull;       // s_memrealtime ticks at 100 MHz
#ifdef MI_TEST_HOOKS
    j.fault_inject = c->fused_fault_inject;
    if (c->fused_timeout_us > 0) j.timeout_ticks = (unsigned long long)c->fused_timeout_us * 100ull;
#endif
    uint32_t* w = c->d_cell;
    j.ctl = w;
    j.tcnt = w + kCellCtlWords;
    j.tready = j.tcnt + c->cell_cap_tiles;
    j.lutpub = j.tready + c->cell_cap_tiles;
    j.ghist = j.lutpub + c->cell_cap_tiles * kCellLutWords;
    j.sflag = j.ghist + c->cell_cap_tiles * 256;
    j.luts_fix = c->d_luts;
    j.host_repaired = c->h_mirror + kCellMirrorBase + (c->cell_generation % 16);
    // persistent grid: every workgroup resident from the start (two 512-thread workgroups per CU: the kernel's launch bounds),
    // a multiple of eight so that every dispenser has the same number of workgroups, never more than there are tickets per dispenser
    const long long per_queue = (long long)j.rows_per_queue * cg.cells_x * a.n_frames;
    const long long n = std::max<long long>(cg.cells_x, std::min<long long>(per_queue, (long long)c->cu_count * kCellWgsPerCu / kCellQueues));
    const unsigned grid = (unsigned)(n * kCellQueues);
    if (!c->capturing) c->cell_pair_open = true;
    if (g.contract) LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_cell_fused_kernel<true>, dim3(grid), dim3(kCell2Threads), 0, j);
    else            LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_cell_fused_kernel<false>, dim3(grid), dim3(kCell2Threads), 0, j);
    // always: housekeeping in the normal case, stamp-driven repair when a bounded wait expired (kernels/clahe_cell.hip.h)
    const int fin_grid = (int)std::min<long long>(a.n_frames, (long long)c->cu_count * 4);
    if (g.contract) LAUNCH(c, s, MI_K_FUSED_FINISH, clahe_cell_finish_kernel<true>, dim3((unsigned)fin_grid), dim3(kThreads), 0, j);
    else            LAUNCH(c, s, MI_K_FUSED_FINISH, clahe_cell_finish_kernel<false>, dim3((unsigned)fin_grid), dim3(kThreads), 0, j);
    c->cell_pair_open = false;
    return MI_OK;
}

mi_status clahe_dev(mi_ctx* c, hipStream_t s, const PlaneArgs& a, double clip_limit, int tiles_x, int tiles_y, const UVJob* uv)
{
    ClaheGeom g;
    mi_status st = clahe_geometry(c, a.width, a.height, clip_limit, tiles_x, tiles_y, &g);
    if (st) return st;
    const int tiles = tiles_x * tiles_y;
    CellGeom cg{};
    if (clahe_cell_applicable(c, a, g, &cg) && clahe_cell_admit(c)) return clahe_cell_fused_dev(c, s, a, g, cg, uv);
    if (c->clahe_single_read == 2 && a.n_frames <= kMaxGridY && clahe_cell_geometry(c, a, g, &cg)) {
        // stage-1 measurement form (R5.5): the tile LUTs come from the tile-histogram pass, the cell kernel does a fused kernel's per-workgroup work
        st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)a.n_frames * cg.cells_x * cg.cells_y * 256 * sizeof(uint32_t));   // before the LUT pass: no reallocation between the two launches
        if (st) return st;
        st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)a.n_frames * tiles * 256);
        if (st) return st;
        st = launch_tile_luts(c, s, a, g, 0, a.n_frames, c->d_luts);
        if (st) return st;
        return launch_cells(c, s, a, g, cg, 0, a.n_frames, c->d_luts, uv);
    }
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
