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

/* ------------------------------------------------------------------------------------------------------------------
 * BASELINE.json config 5 read literally (second half): an NV12 frame taken to BGR, every colour
 * channel equalized, and the result taken back to NV12.  NO file of the reference does this (ColoropenCVCwqualHist.cpp
 * equalizes Y only, :146/:165); the pipeline restated here is what a maintainer would write with OpenCV 4.4:
 *     cvtColor(nv12, bgr, COLOR_YUV2BGR_NV12); split; equalizeHist(B), (G), (R); merge;
 *     cvtColor(bgr, i420, COLOR_BGR2YUV_I420); interleave U and V into the NV12 chroma plane.
 * PARITY UNPINNED (twice over: no reference implementation and no OpenCV here).  4:2:0 arithmetic restated from OpenCV
 * 4.4 modules/imgproc/src/color_yuv.simd.hpp (YUV420sp2RGB8Invoker / RGB8toYUV420pInvoker), ITUR_BT_601_SHIFT = 20:
 *     decode: CY 1220542, CUB 2116026, CUG -409993, CVG -852492, CVR 1673527
 *        y = max(0, Y-16)*CY;  ruv = 2^19 + CVR*(V-128);  guv = 2^19 + CVG*(V-128) + CUG*(U-128);  buv = 2^19 + CUB*(U-128)
 *        R,G,B = saturate_cast<uchar>((y + {r,g,b}uv) >> 20); the four pixels of a 2x2 block share one (U,V)
 *     encode: CRY 269484, CGY 528482, CBY 102760, CRU -155188, CGU -305135, CBU 460324, CGV -385875, CBV -74448
 *        Y = sat((CRY*R + CGY*G + CBY*B + 2^19 + (16 << 20)) >> 20) for every pixel
 *        U = sat((CRU*R + CGU*G + CBU*B + 2^19 + (128 << 20)) >> 20), V = sat((CBU*R + CGV*G + CBV*B + 2^19 + (128 << 20)) >> 20)
 *        taken from the TOP-LEFT pixel of each 2x2 block (no averaging)
 * ------------------------------------------------------------------------------------------------------------------ */
enum { BT_SHIFT = 20, BT_CY = 1220542, BT_CUB = 2116026, BT_CUG = -409993, BT_CVG = -852492, BT_CVR = 1673527,
       BT_CRY = 269484, BT_CGY = 528482, BT_CBY = 102760, BT_CRU = -155188, BT_CGU = -305135, BT_CBU = 460324,
       BT_CGV = -385875, BT_CBV = -74448 };

/* NV12 (tight: Y rows of `width` bytes, then height/2 rows of interleaved U,V) -> CV_8UC3 BGR (tight). width, height even. */
int orc_nv12_to_bgr(const uint8_t* nv12, uint8_t* bgr, int width, int height)
{
    if (width < 0 || height < 0 || (width & 1) || (height & 1)) return 1;
    if (width == 0 || height == 0) return 0;
    if (!nv12 || !bgr) return 1;
    const uint8_t* uvp = nv12 + (size_t)width * height;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint8_t* yr = nv12 + (size_t)y * width;
        const uint8_t* uvr = uvp + (size_t)(y / 2) * width;
        uint8_t* d = bgr + (size_t)y * width * 3;
        for (int x = 0; x < width; ++x) {
            const int uu = (int)uvr[x & ~1] - 128, vv = (int)uvr[(x & ~1) + 1] - 128;
            const int ruv = (1 << (BT_SHIFT - 1)) + BT_CVR * vv;
            const int guv = (1 << (BT_SHIFT - 1)) + BT_CVG * vv + BT_CUG * uu;
            const int buv = (1 << (BT_SHIFT - 1)) + BT_CUB * uu;
            const int yy = (yr[x] > 16 ? (int)yr[x] - 16 : 0) * BT_CY;
            d[3 * x] = sat((yy + buv) >> BT_SHIFT); d[3 * x + 1] = sat((yy + guv) >> BT_SHIFT); d[3 * x + 2] = sat((yy + ruv) >> BT_SHIFT);
        }
    }
    return 0;
}

/* CV_8UC3 BGR (tight) -> NV12 (tight): COLOR_BGR2YUV_I420 arithmetic, U and V interleaved. */
int orc_bgr_to_nv12(const uint8_t* bgr, uint8_t* nv12, int width, int height)
{
    if (width < 0 || height < 0 || (width & 1) || (height & 1)) return 1;
    if (width == 0 || height == 0) return 0;
    if (!nv12 || !bgr) return 1;
    uint8_t* uvp = nv12 + (size_t)width * height;
    const int half = 1 << (BT_SHIFT - 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = bgr + (size_t)y * width * 3;
        uint8_t* yr = nv12 + (size_t)y * width;
        for (int x = 0; x < width; ++x) {
            const int b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
            yr[x] = sat((BT_CRY * r + BT_CGY * g + BT_CBY * b + half + (16 << BT_SHIFT)) >> BT_SHIFT);
            if (!(y & 1) && !(x & 1)) {
                uint8_t* uv = uvp + (size_t)(y / 2) * width + x;
                uv[0] = sat((BT_CRU * r + BT_CGU * g + BT_CBU * b + half + (128 << BT_SHIFT)) >> BT_SHIFT);
                uv[1] = sat((BT_CBU * r + BT_CGV * g + BT_CBV * b + half + (128 << BT_SHIFT)) >> BT_SHIFT);
            }
        }
    }
    return 0;
}

/* The whole literal config-5 frame op: NV12 in -> NV12 out (in place allowed). */
int orc_nv12_bgr_equalize(const uint8_t* nv12_in, uint8_t* nv12_out, int width, int height)
{
    if (width < 0 || height < 0 || (width & 1) || (height & 1)) return 1;
    if (width == 0 || height == 0) return 0;
    const size_t n = (size_t)width * height;
    uint8_t* bgr = (uint8_t*)malloc(n * 3);
    uint8_t* p = (uint8_t*)malloc(n);
    uint8_t* q = (uint8_t*)malloc(n);
    if (!bgr || !p || !q) { free(bgr); free(p); free(q); return 4; }
    int rc = orc_nv12_to_bgr(nv12_in, bgr, width, height);
    for (int c = 0; c < 3 && !rc; ++c) {
        for (size_t i = 0; i < n; ++i) p[i] = bgr[3 * i + c];                        /* split */
        rc = orc_equalize_hist_u8(p, (size_t)width, q, (size_t)width, width, height);
        for (size_t i = 0; i < n; ++i) bgr[3 * i + c] = q[i];                        /* merge */
    }
    if (!rc) rc = orc_bgr_to_nv12(bgr, nv12_out, width, height);
    free(bgr); free(p); free(q);
    return rc;
}

/* cv::cvtColor(bgr, yuv, COLOR_BGR2YUV_I420) as the reference calls it (1frameMeasure.cpp:32): tight CV_8UC3 in, tight
 * W x H*3/2 CV_8UC1 out = Y plane, U plane (W/2 x H/2), V plane.  Same arithmetic as orc_bgr_to_nv12, planar chroma. */
int orc_bgr_to_i420(const uint8_t* bgr, uint8_t* i420, int width, int height)
{
    if (width < 0 || height < 0 || (width & 1) || (height & 1)) return 1;
    if (width == 0 || height == 0) return 0;
    if (!bgr || !i420) return 1;
    const size_t n = (size_t)width * height;
    uint8_t* nv = (uint8_t*)malloc(n * 3 / 2);
    if (!nv) return 4;
    const int rc = orc_bgr_to_nv12(bgr, nv, width, height);
    memcpy(i420, nv, n);
    for (size_t i = 0; i < n / 4; ++i) { i420[n + i] = nv[n + 2 * i]; i420[n + n / 4 + i] = nv[n + 2 * i + 1]; }
    free(nv);
    return rc;
}
