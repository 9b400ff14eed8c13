// Per-call cost of the host-pointer NV12 form, single thread: pageable vs registered buffers, equalize vs CLAHE.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../include/mi_lumaeq.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int W = 3840, H = 2160; const size_t fb = (size_t)W * H * 3 / 2;
    mi_ctx* c; if (mi_ctx_create(0, &c) != MI_OK) { printf("no device\n"); return 1; }
    unsigned char* in = (unsigned char*)aligned_alloc(4096, fb); unsigned char* out = (unsigned char*)aligned_alloc(4096, fb);
    for (size_t i = 0; i < fb; ++i) in[i] = (unsigned char)(i * 2654435761u >> 24);
    memset(out, 0, fb);
    for (int pinned = 0; pinned < 2; ++pinned) {
        if (pinned) { mi_host_register(in, fb); mi_host_register(out, fb); }
        for (int op = 0; op < 2; ++op) for (int uv = 0; uv < 2; ++uv) {
            for (int w = 0; w < 5; ++w) op ? mi_clahe_nv12(c, in, out, W, H, (mi_uv_mode)uv, 2.0, 8, 8) : mi_equalize_hist_nv12(c, in, out, W, H, (mi_uv_mode)uv);
            const int reps = 100; const double t0 = now();
            for (int r = 0; r < reps; ++r) op ? mi_clahe_nv12(c, in, out, W, H, (mi_uv_mode)uv, 2.0, 8, 8) : mi_equalize_hist_nv12(c, in, out, W, H, (mi_uv_mode)uv);
            printf("%s %s uv=%s: %.3f ms/frame\n", pinned ? "registered" : "pageable  ", op ? "clahe   " : "equalize", uv ? "copy" : "fill", (now() - t0) / reps * 1e3);
        }
    }
    // Y plane only (the cv::Mat form)
    for (int w = 0; w < 5; ++w) mi_equalize_hist_u8(c, in, W, out, W, W, H);
    double t0 = now(); for (int r = 0; r < 100; ++r) mi_equalize_hist_u8(c, in, W, out, W, W, H);
    printf("registered Y plane only: %.3f ms\n", (now() - t0) / 100 * 1e3);
    // a ring of 8 registered frame pairs, as a streaming caller would cycle them
    {
        const int R = 8;
        unsigned char* rin[R]; unsigned char* rout[R];
        for (int k = 0; k < R; ++k) {
            rin[k] = (unsigned char*)aligned_alloc(4096, fb); rout[k] = (unsigned char*)aligned_alloc(4096, fb);
            memcpy(rin[k], in, fb); memset(rout[k], 0, fb);
            mi_host_register(rin[k], fb); mi_host_register(rout[k], fb);
        }
        for (int w = 0; w < 16; ++w) mi_equalize_hist_nv12(c, rin[w % R], rout[w % R], W, H, MI_UV_FILL128);
        t0 = now(); for (int r = 0; r < 200; ++r) mi_equalize_hist_nv12(c, rin[r % R], rout[r % R], W, H, MI_UV_FILL128);
        printf("registered ring of 8: %.3f ms/frame\n", (now() - t0) / 200 * 1e3);
        // same with vectors-style malloc'd (unaligned to pages) buffers
    }
    mi_ctx_destroy(c);
    return 0;
}
