// diff.inc.hpp -- absdiff + analyzeDiff, the reference's own parity statistic (1frameMeasure.cpp:91-100), as C ABI entry points
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

namespace {

mi_status analyze_diff_dev(mi_ctx* c, hipStream_t s, const uint8_t* a, size_t a_step, size_t a_frame, const uint8_t* b, size_t b_step,
                           size_t b_frame, uint8_t* diff, size_t d_step, size_t d_frame, int width, int height, int n_frames,
                           int threshold, uint32_t* d_stats)
{
    const long long total = (long long)width * height;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxGridY) {
        const int nf = std::min(kMaxGridY, n_frames - f0);
        uint32_t* st = d_stats + 4 * (size_t)f0;
        LAUNCH(c, s, MI_K_DIFF, diff_init_kernel, dim3((nf + kThreads - 1) / kThreads), dim3(kThreads), 0, st, nf, (uint32_t)total);
        if (total == 0) continue;
        DiffJob j{};
        j.a = a + (size_t)f0 * a_frame; j.b = b ? b + (size_t)f0 * b_frame : nullptr; j.diff = diff ? diff + (size_t)f0 * d_frame : nullptr;
        j.a_frame = (long long)a_frame; j.b_frame = (long long)b_frame; j.d_frame = (long long)d_frame;
        const bool contiguous = a_step == (size_t)width && (!b || b_step == (size_t)width) && (!diff || d_step == (size_t)width);
        if (contiguous || height == 1) { j.rows = 1; j.row_bytes = total; j.a_step = j.b_step = j.d_step = total; }
        else { j.rows = height; j.row_bytes = width; j.a_step = (long long)a_step; j.b_step = (long long)b_step; j.d_step = (long long)d_step; }
        j.threshold = threshold;
        const int B = blocks_per_frame(c, total * (b ? 2 : 1), j.rows, nf, 2048);
        LAUNCH(c, s, MI_K_DIFF, analyze_diff_kernel, dim3(B, nf), dim3(kThreads), 0, j, st);
    }
    return MI_OK;
}

mi_status check_diff_args(mi_ctx* c, const void* a, size_t a_step, const void* b, size_t b_step, const void* diff, size_t d_step,
                          int width, int height, int n_frames, int threshold)
{
    if (width < 0 || height < 0 || n_frames < 0) return fail(c, MI_ERR_BAD_ARG, "negative size");
    if (threshold < 0 || threshold > 255) return fail(c, MI_ERR_BAD_ARG, "threshold must be in [0, 255]");
    if ((long long)width * height > 0x7fffffffLL) return fail(c, MI_ERR_UNSUPPORTED, "width*height must be < 2^31");
    if (width == 0 || height == 0 || n_frames == 0) return MI_OK;
    if (!a) return fail(c, MI_ERR_BAD_ARG, "null plane pointer");
    if (a_step < (size_t)width || (b && b_step < (size_t)width) || (diff && d_step < (size_t)width)) return fail(c, MI_ERR_BAD_ARG, "step < width");
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_analyze_diff_u8_batch_dev(mi_ctx* c, const void* d_a, size_t a_step, size_t a_frame_stride,
                                       const void* d_b, size_t b_step, size_t b_frame_stride,
                                       void* d_diff, size_t diff_step, size_t diff_frame_stride,
                                       int width, int height, int n_frames, int threshold, mi_diff_stats* d_stats, void* stream)
{
    ENTER_COMPUTE(c);
    mi_status st = check_diff_args(c, d_a, a_step, d_b, b_step, d_diff, diff_step, width, height, n_frames, threshold);
    if (st) return st;
    if (n_frames == 0) return MI_OK;
    if (!d_stats) return fail(c, MI_ERR_BAD_ARG, "null d_stats");
    hipStream_t s = pick_stream(c, stream);
    return analyze_diff_dev(c, s, (const uint8_t*)d_a, a_step, a_frame_stride, (const uint8_t*)d_b, b_step, b_frame_stride,
                            (uint8_t*)d_diff, diff_step, diff_frame_stride, width, height, n_frames, threshold, (uint32_t*)d_stats);
}

// Host planes: a (and b) are uploaded into the context's staging frames as compact planes, the difference image comes back the same way.
mi_status mi_analyze_diff_u8(mi_ctx* c, const uint8_t* a, size_t a_step, const uint8_t* b, size_t b_step,
                             uint8_t* diff, size_t diff_step, int width, int height, int threshold, mi_diff_stats* out)
{
    ENTER_COMPUTE(c);
    mi_status st = check_diff_args(c, a, a_step, b, b_step, diff, diff_step, width, height, 1, threshold);
    if (st) return st;
    if (!out) return fail(c, MI_ERR_BAD_ARG, "null out");
    const size_t plane = (size_t)width * height;
    if (plane == 0) { *out = mi_diff_stats{0, 0, 0, 0}; return MI_OK; }
    hipStream_t s = c->stream;
    c->capturing = false;
    if ((st = grow_dev(c, &c->d_stage_in, &c->stage_in_bytes, 2 * plane + 64))) return st;
    if (diff && (st = grow_dev(c, &c->d_stage_out, &c->stage_out_bytes, plane))) return st;
    uint8_t* d_a = c->d_stage_in;
    uint8_t* d_b = b ? c->d_stage_in + plane : nullptr;
    uint32_t* d_st = reinterpret_cast<uint32_t*>(c->d_stage_in + ((2 * plane + 15) & ~(size_t)15));
    // host planes are packed into the context's pinned staging (the library never hands the runtime memory it did not pin itself)
    if ((st = grow_pinned(c, &c->h_pin_in, &c->pin_in_bytes, 2 * plane))) return st;
    if ((st = grow_pinned(c, &c->h_pin_out, &c->pin_out_bytes, plane + 64))) return st;
    copy_rows(c->h_pin_in, (size_t)width, a, a_step, width, height);
    if (b) copy_rows(c->h_pin_in + plane, (size_t)width, b, b_step, width, height);
    StreamDrain drain(HipStreamSync{}, drain_counter(c));        // only library-owned staging is in flight here; an error exit still leaves the stream idle
    drain.watch(s);
    HIPCHK(c, hipMemcpyAsync(d_a, c->h_pin_in, (b ? 2 : 1) * plane, hipMemcpyHostToDevice, s));
    st = analyze_diff_dev(c, s, d_a, (size_t)width, plane, d_b, (size_t)width, plane, diff ? c->d_stage_out : nullptr, (size_t)width, plane,
                          width, height, 1, threshold, d_st);
    if (st) return st;
    if (diff) HIPCHK(c, hipMemcpyAsync(c->h_pin_out, c->d_stage_out, plane, hipMemcpyDeviceToHost, s));
    mi_diff_stats* h_st = reinterpret_cast<mi_diff_stats*>(c->h_pin_out + ((plane + 15) & ~(size_t)15));
    HIPCHK(c, hipMemcpyAsync(h_st, d_st, sizeof(mi_diff_stats), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    drain.done();
    if (diff) copy_rows(diff, diff_step, c->h_pin_out, (size_t)width, width, height);
    *out = *h_st;
    return MI_OK;
}

}  // extern "C"
