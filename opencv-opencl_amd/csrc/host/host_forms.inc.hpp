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

struct Guard {
    mi_ctx* c; std::unique_lock<std::mutex> lk; hipError_t err;
    explicit Guard(mi_ctx* c_) : c(c_), lk(c_->mu) { err = hipSetDevice(c->device); }
};

#define ENTER(ctx)                                                   \
    if (!(ctx)) return MI_ERR_BAD_ARG;                               \
    Guard guard__(ctx);                                              \
    if (guard__.err != hipSuccess) return fail_hip((ctx), guard__.err, "hipSetDevice")

hipStream_t pick_stream(mi_ctx* c, void* stream) { return stream == MI_STREAM_CTX ? c->stream : (hipStream_t)stream; }

// ---- host-pointer plumbing ---------------------------------------------------------------------------
void copy_rows(uint8_t* dst, size_t dst_step, const uint8_t* src, size_t src_step, int width, int height)
{
    if (dst_step == (size_t)width && src_step == (size_t)width) { memcpy(dst, src, (size_t)width * height); return; }
    for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * dst_step, src + (size_t)y * src_step, (size_t)width);
}

}  // namespace
