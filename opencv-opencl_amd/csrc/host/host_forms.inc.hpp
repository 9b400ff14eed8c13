// host_forms.inc.hpp -- NV12 UV job, context guard, host-pointer plumbing helpers
// Included by ../mi_lumaeq.hip (one translation unit; not a stand-alone header).

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

// Every entry point holds the context's lock for its whole duration.  Leaving a device-form call that ran on a caller's stream
// while a pipe is open on the context records that point in an event: the pipe's compute stream waits for it before it touches
// the scratch the two share (d_fused, d_luts, d_partial); see mi_pipe_submit.
struct Guard {
    mi_ctx* c; std::unique_lock<std::mutex> lk; hipError_t err;
    explicit Guard(mi_ctx* c_) : c(c_), lk(c_->mu) { err = hipSetDevice(c->device); c->capturing = false; c->cur_stream_set = false; }
    ~Guard()
    {
        if (c->pipes_open > 0 && c->cur_stream_set && !c->capturing && c->ev_scratch
            && hipEventRecord(c->ev_scratch, c->cur_stream) == hipSuccess)
            c->scratch_foreign = true;
        c->cur_stream_set = false;
    }
};

#define ENTER(ctx)                                                   \
    if (!(ctx)) return MI_ERR_BAD_ARG;                               \
    Guard guard__(ctx);                                              \
    if (guard__.err != hipSuccess) return fail_hip((ctx), guard__.err, "hipSetDevice")

// The context's private stream (MI_STREAM_CTX, the host-pointer forms, statistics reads), created on first use.
mi_status ensure_stream(mi_ctx* c)
{
    if (c->stream) return MI_OK;
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    return MI_OK;
}

// Entry points that use the context's scratch: refused while frames are pending in the context's pipe (include/mi_lumaeq.h:
// "the context's other entry points may be used while no frame is pending") -- the pipe's streams would race them for it.
#define ENTER_COMPUTE(ctx)                                           \
    ENTER(ctx);                                                      \
    if ((ctx)->pipe_pending > 0) return fail((ctx), MI_ERR_BUSY, "frames are pending in this context's pipe: call mi_pipe_wait first"); \
    if (mi_status st_es__ = ensure_stream(ctx)) return st_es__

// Also notes whether the chosen stream is being captured into a hipGraph: scratch growth is refused then, and from the
// first capture on no scratch a graph node may reference is ever freed (grow_dev).
hipStream_t pick_stream(mi_ctx* c, void* stream)
{
    hipStream_t s = stream == MI_STREAM_CTX ? c->stream : (hipStream_t)stream;
    c->capturing = false;
    c->cur_stream = s; c->cur_stream_set = true;
    if (s != c->stream) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess) c->capturing = cap != hipStreamCaptureStatusNone;
        else (void)hipGetLastError();
    }
    if (c->capturing) c->graph_captured = true;
    return s;
}

// ---- host-pointer plumbing ---------------------------------------------------------------------------
void copy_rows(uint8_t* dst, size_t dst_step, const uint8_t* src, size_t src_step, int width, int height)
{
    if (dst_step == (size_t)width && src_step == (size_t)width) { memcpy(dst, src, (size_t)width * height); return; }
    for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * dst_step, src + (size_t)y * src_step, (size_t)width);
}

// Generic host image <-> device staging for the less travelled host forms (colour, 4:2:0, 16-bit): `rows` rows of `row`
// bytes at pitch `step`.  Contiguous images in PINNED memory go to the copy engine as they are; everything else is packed
// through the context's pinned buffers (the library never hands the runtime memory it did not pin itself, see host_op()).
// `drain` watches the stream from the first copy on caller memory; stage_out() synchronises the stream and releases it.
mi_status stage_in(mi_ctx* c, hipStream_t s, const uint8_t* src, size_t step, size_t row, size_t rows, StreamDrain& drain)
{
    const size_t bytes = row * rows;
    mi_status st = grow_dev(c, &c->d_stage_in, &c->stage_in_bytes, bytes);
    if (st) return st;
    const bool direct = (step == row || rows == 1) && host_range_pinned(src, bytes, &c->pin_neg);
    if (!direct) {
        if ((st = grow_pinned(c, &c->h_pin_in, &c->pin_in_bytes, bytes))) return st;
        copy_rows(c->h_pin_in, row, src, step, (int)row, (int)rows);
    }
    drain.watch(s);
    HIPCHK(c, hipMemcpyAsync(c->d_stage_in, direct ? src : c->h_pin_in, bytes, hipMemcpyHostToDevice, s));
    return MI_OK;
}

mi_status stage_out(mi_ctx* c, hipStream_t s, uint8_t* dst, size_t step, size_t row, size_t rows, StreamDrain& drain)
{
    const size_t bytes = row * rows;
    const bool direct = (step == row || rows == 1) && host_range_pinned(dst, bytes, &c->pin_neg);
    mi_status st;
    if (!direct && (st = grow_pinned(c, &c->h_pin_out, &c->pin_out_bytes, bytes))) return st;
    drain.watch(s);
    HIPCHK(c, hipMemcpyAsync(direct ? dst : c->h_pin_out, c->d_stage_out, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    drain.done();
    if (!direct) copy_rows(dst, step, c->h_pin_out, row, (int)row, (int)rows);
    return MI_OK;
}

}  // namespace
