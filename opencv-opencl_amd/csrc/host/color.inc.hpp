// color.inc.hpp -- colour-domain neighbours: launchers + extern "C" entry points
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

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
    const int gy = std::min(j.rows, 65535);
    long long bx = ((long long)c->cu_count * 8 + (long long)gy * n_frames - 1) / ((long long)gy * n_frames);
    bx = std::max<long long>(1, std::min<long long>(bx, (j.row_px / 16 + kThreads - 1) / kThreads + 1));
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
    if (op == MI_OP_EQUALIZE && c->bgr_fused) {
        // two passes over the interleaved image (9 B/px): Y histogram on the fly, then convert + LUT + convert back
        ColorJob j = color_job(a);
        for (int f0 = 0; f0 < a.n_frames; f0 += kMaxGridY) {
            const int nf = std::min(kMaxGridY, a.n_frames - f0);
            ColorJob q = j;
            q.src += (long long)f0 * j.src_frame; q.dst += (long long)f0 * j.dst_frame;
            const int B = blocks_per_frame(c, (long long)a.width * a.height * 3, 1, nf, 256);
            mi_status st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * B * 256 * sizeof(uint32_t));
            if (st) return st;
            if ((st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)nf * 256))) return st;
            LAUNCH(c, s, MI_K_COLOR, bgr_luma_hist_kernel, dim3(B, 1, nf), dim3(kBgrThreads), 0, q, c->d_partial);
            LAUNCH(c, s, MI_K_EQ_LUT, equalize_lut_kernel, dim3(nf), dim3(kThreads), 0,
                   (const uint32_t*)c->d_partial, B, (int)((long long)a.width * a.height), c->d_luts, (int32_t*)nullptr);
            const int B2 = blocks_per_frame(c, (long long)a.width * a.height * 3, 1, nf, 2048);
            LAUNCH(c, s, MI_K_COLOR, bgr_luma_apply_kernel, dim3(B2, 1, nf), dim3(kBgrThreads), 0, q, (const uint8_t*)c->d_luts);
        }
        return MI_OK;
    }
    if (op == MI_OP_CLAHE && c->bgr_fused) {
        // two passes over the interleaved image (9 B/px) when the shape allows it: no REFLECT_101 padding, 16-pixel groups
        // that never straddle a tile, 16-B aligned rows, f32 pair tables (tiles_x <= 14)
        ClaheGeom g;
        mi_status st = clahe_geometry(c, a.width, a.height, clip, tx, ty, &g);
        if (st) return st;
        const int tiles = tx * ty;
        const bool shape_ok = !g.contract && a.width % tx == 0 && a.height % ty == 0 && g.tile_w % 16 == 0 && tx + 1 <= kMaxPairsLdsF32 &&
                              tiles <= kMaxGridY && a.width / kInterpPx <= kThreads * kMaxGridY &&
                              (((uintptr_t)a.src | (uintptr_t)a.dst | a.src_step | a.dst_step | a.src_frame | a.dst_frame) & 15) == 0;
        if (shape_ok) {
            const ColorJob j = color_job(a);
            for (int f0 = 0; f0 < a.n_frames; f0 += kMaxGridY) {
                const int nf = std::min(kMaxGridY, a.n_frames - f0);
                ColorJob q = j;
                q.src += (long long)f0 * j.src_frame; q.dst += (long long)f0 * j.dst_frame;
                q.src_step = (long long)a.src_step; q.dst_step = (long long)a.dst_step;       // row-wise addressing (color_job() may flatten)
                long long want = ((long long)c->cu_count * 8 + (long long)tiles * nf - 1) / ((long long)tiles * nf);
                const int S = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, g.tile_h / 8), 64LL}));
                if ((st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * tiles * S * 256 * sizeof(uint32_t)))) return st;
                if ((st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)nf * tiles * 256))) return st;
                uint8_t* direct = S == 1 ? c->d_luts : nullptr;
                LAUNCH(c, s, MI_K_TILE_HIST, bgr_tile_hist_kernel<512>, dim3(S, tiles, nf), dim3(512), 0,
                       q.src, q.src_step, q.src_frame, g, c->d_partial, direct);
                if (!direct)
                    LAUNCH(c, s, MI_K_TILE_LUT, tile_lut_kernel, dim3(tiles, nf), dim3(kThreads), 0, (const uint32_t*)c->d_partial, S, g, c->d_luts);
                const int ngroups = a.width / kInterpPx;
                const int groups = std::min(ngroups, kThreads);
                const int segs = (ngroups + groups - 1) / groups;
                const int bands = ty + 1;
                want = ((long long)c->cu_count * 8 + (long long)bands * nf * segs - 1) / ((long long)bands * nf * segs);
                const int subs = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, (g.tile_h + 2 * kBandMargin) / 8), 64LL}));
                LAUNCH(c, s, MI_K_CLAHE_INTERP, bgr_clahe_interp_kernel, dim3(bands * subs, nf, segs), dim3(kThreads),
                       (size_t)(tx + 1) * 256 * 4 * sizeof(float), q, g, (const uint8_t*)c->d_luts, subs, groups);
            }
            return MI_OK;
        }
    }
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


// ---- BASELINE.json config 5 read literally: NV12 -> BGR -> equalizeHist on B, G, R -> NV12 (kernels: color.hip.h) ----
constexpr int kMaxFramesCh = kMaxGridY / 3;        // the LUT kernel runs one workgroup per (frame, channel)

mi_status check_nv12_420(mi_ctx* c, const void* in, const void* out, int width, int height, int n_frames)
{
    if (width < 0 || height < 0 || n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "negative size");
    if ((width & 1) || (height & 1)) return fail(c, MI_ERR_BAD_ARG, "4:2:0 conversion needs even width and height");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    if (!in || !out) return fail(c, MI_ERR_BAD_ARG, "null frame pointer");
    if ((long long)width * height > 0x7fffffffLL / 3) return fail(c, MI_ERR_UNSUPPORTED, "image too large");
    return MI_OK;
}

mi_status nv12_bgr_equalize_dev(mi_ctx* c, hipStream_t s, const uint8_t* in, size_t in_frame, uint8_t* out, size_t out_frame,
                                int width, int height, int n_frames)
{
    const long long ysz = (long long)width * height;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFramesCh) {
        const int nf = std::min(kMaxFramesCh, n_frames - f0);
        Nv12Job j{};
        j.in = in + (size_t)f0 * in_frame; j.out = out + (size_t)f0 * out_frame;
        j.in_frame = (long long)in_frame; j.out_frame = (long long)out_frame;
        j.width = width; j.height = height;
        j.vec = (width % 16 == 0) && ((((uintptr_t)j.in | (uintptr_t)j.out | in_frame | out_frame) & 15) == 0);
        const int B = blocks_per_frame(c, ysz * 3 / 2, 1, nf, 256);
        mi_status st = grow_dev(c, &c->d_partial, &c->partial_bytes, (size_t)nf * 3 * B * 256 * sizeof(uint32_t));
        if (st) return st;
        if ((st = grow_dev(c, &c->d_luts, &c->luts_bytes, (size_t)nf * 3 * 256))) return st;
        LAUNCH(c, s, MI_K_COLOR, nv12_bgr_hist_kernel, dim3(B, nf), dim3(kNv12Threads), 0, j, c->d_partial);
        LAUNCH(c, s, MI_K_EQ_LUT, equalize_lut_kernel, dim3(nf * 3), dim3(kThreads), 0,
               (const uint32_t*)c->d_partial, B, (int)ysz, c->d_luts, (int32_t*)nullptr);
        const int B2 = blocks_per_frame(c, ysz * 3 / 2, 1, nf, 2048);
        LAUNCH(c, s, MI_K_COLOR, nv12_bgr_apply_kernel, dim3(B2, nf), dim3(kNv12Threads), 0, j, (const uint8_t*)c->d_luts);
    }
    return MI_OK;
}


// cv::cvtColor 4:2:0 codes as stand-alone conversions.  `c3_*` describe the CV_8UC3 side, the planar side is tight.
mi_status cvt420_dev(mi_ctx* c, hipStream_t s, int code, const uint8_t* src, uint8_t* dst, size_t c3_step, size_t c3_frame,
                     size_t planar_frame, int width, int height, int n_frames)
{
    const long long blocks = (long long)(width / 2) * (height / 2);
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        Cvt420Job j{};
        const bool enc = code == MI_COLOR_BGR2YUV_I420;
        j.src = src + (size_t)f0 * (enc ? c3_frame : planar_frame);
        j.dst = dst + (size_t)f0 * (enc ? planar_frame : c3_frame);
        j.c3_step = (long long)c3_step; j.c3_frame = (long long)c3_frame; j.planar_frame = (long long)planar_frame;
        j.width = width; j.height = height;
        long long bx = ((long long)c->cu_count * 8 + nf - 1) / nf;
        bx = std::max<long long>(1, std::min<long long>(bx, (blocks + kThreads - 1) / kThreads));
        // 16 x 2 pixel groups per lane with 16-byte accesses when everything is 16-byte aligned, else 2 x 2 blocks with byte accesses
        const int vec = width % 16 == 0 && (((uintptr_t)j.src | (uintptr_t)j.dst | c3_step | c3_frame | planar_frame) & 15) == 0;
        if (vec) bx = std::max<long long>(1, std::min<long long>(bx, (blocks / 8 + kThreads - 1) / kThreads));
        if (enc) LAUNCH(c, s, MI_K_COLOR, cvt420_kernel<0>, dim3((unsigned)bx, nf), dim3(kThreads), 0, j, vec);
        else     LAUNCH(c, s, MI_K_COLOR, cvt420_kernel<1>, dim3((unsigned)bx, nf), dim3(kThreads), 0, j, vec);
    }
    return MI_OK;
}

mi_status check_cvt420(mi_ctx* c, const void* src, size_t src_step, const void* dst, size_t dst_step, int width, int height, int n_frames, int code)
{
    if (code != MI_COLOR_BGR2YUV_I420 && code != MI_COLOR_YUV2BGR_NV12)
        return fail(c, MI_ERR_UNSUPPORTED, "4:2:0 colour code must be 128 (BGR2YUV_I420) or 93 (YUV2BGR_NV12)");
    if (width < 0 || height < 0 || n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "negative size");
    if ((width & 1) || (height & 1)) return fail(c, MI_ERR_BAD_ARG, "4:2:0 conversion needs even width and height");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    if (!src || !dst) return fail(c, MI_ERR_BAD_ARG, "null image pointer");
    const size_t need_src = code == MI_COLOR_BGR2YUV_I420 ? (size_t)width * 3 : (size_t)width;
    const size_t need_dst = code == MI_COLOR_BGR2YUV_I420 ? (size_t)width : (size_t)width * 3;
    if (src_step < need_src || dst_step < need_dst) return fail(c, MI_ERR_BAD_ARG, "step too small for the image width");
    if ((long long)width * height > 0x7fffffffLL / 3) return fail(c, MI_ERR_UNSUPPORTED, "image too large");
    return MI_OK;
}

// host images staged like host_op(): rows -> pinned -> device (tight) -> op -> pinned -> rows
mi_status color_host_op(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height,
                        bool luma, int code_or_op, double clip, int tx, int ty)
{
    const size_t row = (size_t)width * 3, bytes = row * height;
    hipStream_t s = c->stream;
    StreamDrain drain(HipStreamSync{}, drain_counter(c));
    mi_status st = stage_in(c, s, src, src_step, row, (size_t)height, drain);
    if (st) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, bytes))) return st;
    Color3Args a{c->d_stage_in, row, bytes, c->d_stage_out, row, bytes, width, height, 1};
    st = luma ? bgr_luma_dev(c, s, a, code_or_op, clip, tx, ty) : cvt_color_dev(c, s, a, code_or_op);
    if (st) return st;
    return stage_out(c, s, dst, dst_step, row, (size_t)height, drain);
}

}  // namespace

extern "C" {

mi_status mi_cvt_color_u8c3_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                      void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                      int width, int height, int n_frames, int code, void* stream)
{
    ENTER_COMPUTE(c);
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
    ENTER_COMPUTE(c);
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
    ENTER_COMPUTE(c);
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
    ENTER_COMPUTE(c);
    Color3Args a{src, src_step, 0, dst, dst_step, 0, width, height, 1};
    mi_status st = check_color3(c, a);
    if (st) return st;
    if (op != MI_OP_EQUALIZE && op != MI_OP_CLAHE) return fail(c, MI_ERR_BAD_ARG, "op must be MI_OP_EQUALIZE or MI_OP_CLAHE");
    if (op == MI_OP_CLAHE && (tiles_x <= 0 || tiles_y <= 0)) return fail(c, MI_ERR_BAD_ARG, "tile grid must be >= 1x1");
    if (width == 0 || height == 0) return MI_OK;
    return color_host_op(c, src, src_step, dst, dst_step, width, height, true, op, clip_limit, tiles_x, tiles_y);
}

mi_status mi_nv12_bgr_equalize_batch_dev(mi_ctx* c, const void* d_in, size_t in_frame_stride, void* d_out, size_t out_frame_stride,
                                         int width, int height, int n_frames, void* stream)
{
    ENTER_COMPUTE(c);
    mi_status st = check_nv12_420(c, d_in, d_out, width, height, n_frames);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    const size_t frame = (size_t)width * height * 3 / 2;
    if (n_frames > 1 && (in_frame_stride < frame || out_frame_stride < frame)) return fail(c, MI_ERR_BAD_ARG, "frame stride < width*height*3/2");
    return nv12_bgr_equalize_dev(c, pick_stream(c, stream), (const uint8_t*)d_in, in_frame_stride, (uint8_t*)d_out, out_frame_stride,
                                 width, height, n_frames);
}

mi_status mi_nv12_bgr_equalize(mi_ctx* c, const uint8_t* nv12_in, uint8_t* nv12_out, int width, int height)
{
    ENTER_COMPUTE(c);
    mi_status st = check_nv12_420(c, nv12_in, nv12_out, width, height, 1);
    if (st || width == 0 || height == 0) return st;
    const size_t bytes = (size_t)width * height * 3 / 2;
    hipStream_t s = c->stream;
    StreamDrain drain(HipStreamSync{}, drain_counter(c));
    if ((st = stage_in(c, s, nv12_in, bytes, bytes, 1, drain))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, bytes))) return st;
    st = nv12_bgr_equalize_dev(c, s, c->d_stage_in, bytes, c->d_stage_out, bytes, width, height, 1);
    if (st) return st;
    return stage_out(c, s, nv12_out, bytes, bytes, 1, drain);
}

mi_status mi_cvt_color_420_u8_batch_dev(mi_ctx* c, const void* d_src, size_t src_step, size_t src_frame_stride,
                                        void* d_dst, size_t dst_step, size_t dst_frame_stride,
                                        int width, int height, int n_frames, int code, void* stream)
{
    ENTER_COMPUTE(c);
    mi_status st = check_cvt420(c, d_src, src_step, d_dst, dst_step, width, height, n_frames, code);
    if (st || width == 0 || height == 0 || n_frames == 0) return st;
    const bool enc = code == MI_COLOR_BGR2YUV_I420;
    if ((enc ? dst_step : src_step) != (size_t)width) return fail(c, MI_ERR_UNSUPPORTED, "device form: the planar image must be tightly packed (step == width)");
    return cvt420_dev(c, pick_stream(c, stream), code, (const uint8_t*)d_src, (uint8_t*)d_dst, enc ? src_step : dst_step,
                      enc ? src_frame_stride : dst_frame_stride, enc ? dst_frame_stride : src_frame_stride, width, height, n_frames);
}

mi_status mi_cvt_color_420_u8(mi_ctx* c, const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height, int code)
{
    ENTER_COMPUTE(c);
    mi_status st = check_cvt420(c, src, src_step, dst, dst_step, width, height, 1, code);
    if (st || width == 0 || height == 0) return st;
    const bool enc = code == MI_COLOR_BGR2YUV_I420;
    const size_t c3_row = (size_t)width * 3, c3_bytes = c3_row * height, pl_bytes = (size_t)width * height * 3 / 2;
    const size_t in_row = enc ? c3_row : (size_t)width, in_rows = enc ? (size_t)height : (size_t)height * 3 / 2;
    const size_t out_row = enc ? (size_t)width : c3_row, out_rows = enc ? (size_t)height * 3 / 2 : (size_t)height, out_bytes = enc ? pl_bytes : c3_bytes;
    hipStream_t s = c->stream;
    StreamDrain drain(HipStreamSync{}, drain_counter(c));
    if ((st = stage_in(c, s, src, src_step, in_row, in_rows, drain))) return st;
    if ((st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, out_bytes))) return st;
    st = cvt420_dev(c, s, code, c->d_stage_in, c->d_stage_out, c3_row, c3_bytes, pl_bytes, width, height, 1);
    if (st) return st;
    return stage_out(c, s, dst, dst_step, out_row, out_rows, drain);
}

}  // extern "C"
