// clahe16.inc.hpp -- CLAHE on CV_16UC1: launcher + extern "C" entry points
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

// ---- CLAHE on CV_16UC1 ----------------------------------------------------------------------
namespace {

constexpr int kWideHintWord = 32;          // h_mirror[32]: sequence number of the last call that met a 14-bit rectangle; [33]: of the last call executed
constexpr int32_t kWideHintCalls = 8;      // ... and for how many executed calls after it clahe_interp16_mid_kernel is still launched

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
    // scratch per frame: tile histograms (u32) + ushort LUTs + the tiles' populated ranges + the frame's range; frames are
    // processed in chunks that keep it <= ~2 GiB (address space, not traffic: 288 GB of HBM) (the value-major copy of the LUTs reuses the histogram area, which is
    // dead once the LUTs exist).  Only the bins a frame populates are ever written or read (kernels/clahe16.hip.h).
    const size_t per_frame = (size_t)tiles * kHist16 * (sizeof(uint32_t) + sizeof(uint16_t)) + ((size_t)tiles + 1) * sizeof(Range16) + sizeof(uint32_t);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_frames, ((size_t)2 << 30) / per_frame));
    st = grow_dev(c, &c->d_c16, &c->c16_bytes, per_frame * (size_t)chunk);
    if (st) return st;
    // per-frame arrival words of tile_hist12_kernel: zero between launches (its last workgroup per frame leaves them so)
    const size_t sync_need = ((size_t)chunk * 4 + 4) * sizeof(uint32_t);      // + the context's shift hint (last word group)
    if (sync_need > c->sync16_bytes) {
        st = grow_dev(c, &c->d_sync16, &c->sync16_bytes, std::max<size_t>(sync_need, (64 * 4 + 4) * sizeof(uint32_t)));
        if (st) return st;
        HIPCHK(c, hipMemsetAsync(c->d_sync16, 0, c->sync16_bytes, s));
    }
    // vector path of the tile histogram: no REFLECT_101 padding, 8-pixel groups inside one tile, 16-B aligned rows
    const int vec = width % tiles_x == 0 && height % tiles_y == 0 && g.tile_w % 8 == 0 &&
                    (((uintptr_t)src | src_step | src_frame) & 15) == 0;
    const int npairs = tiles_x + 1, bands = tiles_y + 1;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = std::min(chunk, n_frames - f0);
        uint32_t* hist = reinterpret_cast<uint32_t*>(c->d_c16);
        uint16_t* luts = reinterpret_cast<uint16_t*>(c->d_c16 + (size_t)nf * tiles * kHist16 * sizeof(uint32_t));
        Range16* ranges = reinterpret_cast<Range16*>(c->d_c16 + (size_t)nf * tiles * kHist16 * (sizeof(uint32_t) + sizeof(uint16_t)));
        Range16* franges = ranges + (size_t)nf * tiles;
        uint32_t* fdone = reinterpret_cast<uint32_t*>(franges + nf);
        const bool bet12 = vec && c->clahe16_fast12;
        // Rectangles whose range needs 8193..16384 table entries (14-bit content) have a kernel of their own since round 6
        // (clahe_interp16_mid_kernel: one window of a 128-KiB table).  Its launch costs ~8 us whether or not it finds work, so with the
        // option at its default (1) it is launched only while such a rectangle was seen in one of the context's last executed calls
        // (kernels/clahe16.hip.h WideHint); 2 = always (tests), 0 = never.  In place a frame is shared out by RECTANGLES: one window of the
        // small table, one window of the mid kernel's, or -- several windows -- the kernel that gathers from the LUTs in L2.
        const uint32_t seq = ++c->c16_seq;
        const WideHint wh{c->d_sync16 + c->sync16_bytes / sizeof(uint32_t) - 2, c->h_mirror + kWideHintWord, seq};
        const bool mid_recent = mi_host::mid_kernel_wanted(c->clahe16_wide, __atomic_load_n(c->h_mirror + kWideHintWord + 1, __ATOMIC_RELAXED),
                                                           __atomic_load_n(c->h_mirror + kWideHintWord, __ATOMIC_RELAXED), kWideHintCalls);      // host/wide_hint.hpp
        const bool mid_runs = mid_recent && nf <= 1024 && !(tiles <= 64 && c->clahe16_transposed);
        // the context's shift hint: two words at the end of the arrival scratch (read / collect, rolled over by the interpolation kernel)
        uint32_t* hint = bet12 ? c->d_sync16 + c->sync16_bytes / sizeof(uint32_t) - 4 : nullptr;
        // 12-bit bet (kernels/clahe16.hip.h): vector geometry only; a tile that loses it is redone the careful way in the same workgroup
        if (bet12)
            LAUNCH(c, s, MI_K_TILE_HIST, (tile_hist12_kernel<kHist12Threads, kCopies12>), dim3(tiles, nf), dim3(kHist12Threads), kHist12Words * sizeof(uint32_t),
                   src + (size_t)f0 * src_frame, (long long)src_step, (long long)src_frame, g, hist, ranges, lut_scale16, clip16, luts,
                   c->d_sync16, franges, fdone, hint);
        else
            LAUNCH(c, s, MI_K_TILE_HIST, tile_hist16_kernel, dim3(tiles, nf), dim3(1024), kHalf16 * sizeof(uint32_t),
                   src + (size_t)f0 * src_frame, (long long)src_step, (long long)src_frame, g, hist, ranges, vec);
        LAUNCH(c, s, MI_K_TILE_LUT, tile_lut16_kernel, dim3(tiles, nf), dim3(1024), 0, (const uint32_t*)hist, (const Range16*)ranges, g,
               lut_scale16, clip16, luts, franges, (const uint32_t*)(bet12 ? fdone : nullptr), hint, wh);
        if (tiles <= 64 && c->clahe16_transposed) {
            // value-major LUTs (one cache line per pixel value): transposed into the histogram area, which is dead by now
            uint16_t* lutT = reinterpret_cast<uint16_t*>(hist);
            LAUNCH(c, s, MI_K_TILE_LUT, transpose_lut16_kernel, dim3(kHist16 / 256, nf), dim3(kThreads), (size_t)tiles * 256 * sizeof(uint16_t),
                   (const uint16_t*)luts, lutT, tiles);
            LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16T_kernel, dim3((width + kThreads - 1) / kThreads, height, nf), dim3(kThreads), 0,
                   src + (size_t)f0 * src_frame, (long long)src_step, (long long)src_frame,
                   dst + (size_t)f0 * dst_frame, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)lutT, (const Range16*)franges, hint);
        } else {
            // one workgroup per (tile pair, band, sub-band).  Few frames: enough sub-bands to fill the chip (four workgroups per CU), never
            // less than ~16 rows each.  Many frames: still TWO sub-bands per band while they keep 64 rows -- 2 workgroups are resident per
            // CU and a band-high workgroup lives ~50 us, so the last round of a launch is long and half empty (16 4K frames: 1296 unequal
            // workgroups on 512 slots); three or more cost more in table staging (64 KiB per workgroup) than they gain
            // (profiles/r03_n_clahe16_interp_subs.txt).
            const long long per_sub = (long long)npairs * bands * nf;
            const long long want = ((long long)c->cu_count * 4 + per_sub - 1) / per_sub;
            int subs = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, g.tile_h / 16), 16LL}));
            if (subs < 2 && g.tile_h >= 128) subs = 2;
            const long long rows = (long long)bands * subs * nf;
            const long long grid = (rows + 7) / 8 * 8 * npairs;           // rows are dealt to the 8 XCDs whole (kernels/clahe16.hip.h)
            if (grid > 0x7fffffffLL) return fail(c, MI_ERR_UNSUPPORTED, "16-bit CLAHE: tile grid too large");
            const uint8_t* sp = src + (size_t)f0 * src_frame;
            uint8_t* dp = dst + (size_t)f0 * dst_frame;
            if (g.contract)
                LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16_kernel<true>, dim3((unsigned)grid), dim3(kInterp16Threads),
                       (size_t)kInterp16Entries * sizeof(uint2), sp, (long long)src_step, (long long)src_frame,
                       dp, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)luts, (const Range16*)franges, subs, nf, (const Range16*)ranges, hint,
                       mid_runs ? 1 : 0, wh);
            else
                LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16_kernel<false>, dim3((unsigned)grid), dim3(kInterp16Threads),
                       (size_t)kInterp16Entries * sizeof(uint2), sp, (long long)src_step, (long long)src_frame,
                       dp, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)luts, (const Range16*)franges, subs, nf, (const Range16*)ranges, hint,
                       mid_runs ? 1 : 0, wh);
            if (mid_runs) {
                ++c->c16_mid_launches;
                // the same work items, walked by one persistent workgroup per CU (a multiple of 8 workgroups: each stays on its XCD)
                const long long grid_p = std::min<long long>(grid, (long long)(std::max(c->cu_count, 8) + 7) / 8 * 8);
                if (g.contract)
                    LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16_mid_kernel<true>, dim3((unsigned)grid_p), dim3(kInterp16MidThreads),
                           (size_t)kInterp16MidEntries * sizeof(uint2), sp, (long long)src_step, (long long)src_frame,
                           dp, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)luts, (const Range16*)franges, subs, nf, (const Range16*)ranges);
                else
                    LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16_mid_kernel<false>, dim3((unsigned)grid_p), dim3(kInterp16MidThreads),
                           (size_t)kInterp16MidEntries * sizeof(uint2), sp, (long long)src_step, (long long)src_frame,
                           dp, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)luts, (const Range16*)franges, subs, nf, (const Range16*)ranges);
            }
            // IN-PLACE rectangles whose range does not fit one window of a table (their workgroups above returned at once); the launch is a
            // no-op for frames without any, and is left out altogether when the call is not in place (it cost 8 us per call)
            if (sp == dp) {
                const long long wide_items = (long long)((width + kThreads - 1) / kThreads) * height;
                LAUNCH(c, s, MI_K_CLAHE_INTERP, clahe_interp16_wide_kernel, dim3((unsigned)std::min<long long>(wide_items, std::max(512, 2048 / nf)), 1, nf), dim3(kThreads), 0,
                       sp, (long long)src_step, (long long)src_frame, dp, (long long)dst_step, (long long)dst_frame, g, (const uint16_t*)luts, (const Range16*)franges, mid_runs ? 1 : 0, (const Range16*)ranges);
            }
        }
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
    ENTER_COMPUTE(c);
    mi_status st = check_u16(c, d_src, src_step, d_dst, dst_step, width, height, n_frames, tiles_x, tiles_y);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    return clahe16_dev(c, pick_stream(c, stream), (const uint8_t*)d_src, src_step, src_frame_stride, (uint8_t*)d_dst, dst_step, dst_frame_stride,
                       width, height, n_frames, clip_limit, tiles_x, tiles_y);
}

mi_status mi_clahe_u16(mi_ctx* c, const uint16_t* src, size_t src_step, uint16_t* dst, size_t dst_step, int width, int height,
                       double clip_limit, int tiles_x, int tiles_y)
{
    ENTER_COMPUTE(c);
    mi_status st = check_u16(c, src, src_step, dst, dst_step, width, height, 1, tiles_x, tiles_y);
    if (st || width == 0 || height == 0) return st;
    const size_t row = (size_t)width * 2, bytes = row * height;
    hipStream_t s = c->stream;
    StreamDrain drain(HipStreamSync{}, drain_counter(c));
    if ((st = stage_in(c, s, (const uint8_t*)src, src_step, row, (size_t)height, drain))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, bytes))) return st;
    st = clahe16_dev(c, s, c->d_stage_in, row, bytes, c->d_stage_out, row, bytes, width, height, 1, clip_limit, tiles_x, tiles_y);
    if (st) return st;
    return stage_out(c, s, (uint8_t*)dst, dst_step, row, (size_t)height, drain);
}

}  // extern "C"
