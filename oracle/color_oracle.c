/*
 * color_oracle.c -- CPU restatement of the colour-domain NEIGHBOURS of the hot path:
 * cv::cvtColor(COLOR_BGR2YUV / COLOR_YUV2BGR), cv::split / cv::merge on CV_8UC3, as used around the luma op by
 * the reference's image benches (singlecolor.cpp:39-66, clahe1frame.cpp:83-102).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED, and more weakly anchored than lumaeq_oracle.c: OpenCV 4.4 is absent, the reference holds no
 * vectors, and the fixed-point constants below are restated from OpenCV 4.4's published
 * modules/imgproc/src/color_yuv.simd.hpp (RGB2YCrCb_i<uchar> / YCrCb2RGB_i<uchar>, isCrCb = false):
 *     yuv_shift = 14;  R2Y = 4899, G2Y = 9617, B2Y = 1868;  B2UI = 8061, R2VI = 14369;
 *     U2BI = 33292, U2GI = -6472, V2GI = -9519, V2RI = 18678;  CV_DESCALE(x,n) = (x + (1 << (n-1))) >> n
 *     Y  = DESCALE(B*B2Y + G*G2Y + R*R2Y);  U = DESCALE((B - Y)*B2UI + (128 << 14));  V = DESCALE((R - Y)*R2VI + (128 << 14))
 *     B' = Y + DESCALE((U-128)*U2BI);  G' = Y + DESCALE((U-128)*U2GI + (V-128)*V2GI);  R' = Y + DESCALE((V-128)*V2RI)
 * each stored through saturate_cast<uchar>.  tests/test_oracle_vs_opencv.py checks these against a real cv2
 * wherever one exists.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int orc_equalize_hist_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int);
int orc_clahe_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int, double, int, int);

enum { YUV_SHIFT = 14, R2Y = 4899, G2Y = 9617, B2Y = 1868, B2UI = 8061, R2VI = 14369,
       U2BI = 33292, U2GI = -6472, V2GI = -9519, V2RI = 18678 };

static inline int descale(int x) { return (x + (1 << (YUV_SHIFT - 1))) >> YUV_SHIFT; }
static inline uint8_t sat(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static inline void bgr2yuv_px(const uint8_t* s, uint8_t* d)
{
    const int b = s[0], g = s[1], r = s[2];
    const int Y = descale(b * B2Y + g * G2Y + r * R2Y);
    const int V = descale((r - Y) * R2VI + (128 << YUV_SHIFT));      /* "Cr" slot of RGB2YCrCb_i */
    const int U = descale((b - Y) * B2UI + (128 << YUV_SHIFT));      /* "Cb" slot */
    d[0] = sat(Y); d[1] = sat(U); d[2] = sat(V);                      /* yuvOrder: Y, U, V */
}

static inline void yuv2bgr_px(const uint8_t* s, uint8_t* d)
{
    const int Y = s[0], U = s[1], V = s[2];
    const int b = Y + descale((U - 128) * U2BI);
    const int g = Y + descale((U - 128) * U2GI + (V - 128) * V2GI);
    const int r = Y + descale((V - 128) * V2RI);
    d[0] = sat(b); d[1] = sat(g); d[2] = sat(r);
}

/* CV_8UC3 interleaved images; step in bytes >= 3*width.  In place allowed. */
int orc_bgr2yuv_u8(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height)
{
    if (width < 0 || height < 0) return 1;
    if (width == 0 || height == 0) return 0;
    if (!src || !dst || src_step < (size_t)width * 3 || dst_step < (size_t)width * 3) return 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x) { uint8_t o[3]; bgr2yuv_px(s + 3 * x, o); d[3 * x] = o[0]; d[3 * x + 1] = o[1]; d[3 * x + 2] = o[2]; }
    }
    return 0;
}

int orc_yuv2bgr_u8(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int width, int height)
{
    if (width < 0 || height < 0) return 1;
    if (width == 0 || height == 0) return 0;
    if (!src || !dst || src_step < (size_t)width * 3 || dst_step < (size_t)width * 3) return 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x) { uint8_t o[3]; yuv2bgr_px(s + 3 * x, o); d[3 * x] = o[0]; d[3 * x + 1] = o[1]; d[3 * x + 2] = o[2]; }
    }
    return 0;
}

/* singlecolor.cpp:39-66 (op = 0) / clahe1frame.cpp:83-102 (op = 1):
 * cvtColor(BGR2YUV) -> split -> op on planes[0] -> merge -> cvtColor(YUV2BGR), tightly packed CV_8UC3 in and out. */
int orc_bgr_luma_op(const uint8_t* bgr_in, uint8_t* bgr_out, int width, int height, int op,
                    double clip_limit, int tiles_x, int tiles_y)
{
    if (width < 0 || height < 0) return 1;
    if (width == 0 || height == 0) return 0;
    const size_t n = (size_t)width * height;
    uint8_t* yuv = (uint8_t*)malloc(n * 3);
    uint8_t* y = (uint8_t*)malloc(n);
    uint8_t* y2 = (uint8_t*)malloc(n);
    if (!yuv || !y || !y2) { free(yuv); free(y); free(y2); return 4; }
    int rc = orc_bgr2yuv_u8(bgr_in, (size_t)width * 3, yuv, (size_t)width * 3, width, height);
    for (size_t i = 0; i < n; ++i) y[i] = yuv[3 * i];                               /* split */
    if (!rc) rc = op == 0 ? orc_equalize_hist_u8(y, (size_t)width, y2, (size_t)width, width, height)
                          : orc_clahe_u8(y, (size_t)width, y2, (size_t)width, width, height, clip_limit, tiles_x, tiles_y);
    for (size_t i = 0; i < n; ++i) yuv[3 * i] = y2[i];                              /* merge */
    if (!rc) rc = orc_yuv2bgr_u8(yuv, (size_t)width * 3, bgr_out, (size_t)width * 3, width, height);
    free(yuv); free(y); free(y2);
    return rc;
}
