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
