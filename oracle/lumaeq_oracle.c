/*
 * lumaeq_oracle.c -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the *checker*, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.  The shipped path is the HIP library
 * in opencv-opencl_amd/csrc and it has no CPU fallback.
 *
 * PARITY UNPINNED.  The reference (kimkimhun3/OpenCV-OpenCL) does not contain the arithmetic
 * of this path; it calls a third-party dependency that is absent from /root/reference and from
 * this image:
 *     OpenCV 4.4  (libopencv_imgproc.so.4.4 / libopencv_core.so.4.4, pinned only by the ELF
 *                  NEEDED entries of the reference's prebuilt binaries; compile.sh:10-11 asks
 *                  pkg-config for "opencv4" without a patch version)
 * The reference holds no golden vectors or tests for the path (its one parity check,
 * 1frameMeasure.cpp:91-100, needs the FPGA board and tolerates +-1).  What is restated below is
 * OpenCV 4.4's published algorithm (modules/imgproc/src/histogram.cpp cv::equalizeHist,
 * modules/imgproc/src/clahe.cpp CLAHE_Impl::apply, core/fast_math.hpp cvRound/cvFloor,
 * core/saturate.hpp, core/src/copy.cpp borderInterpolate), x86-64 baseline semantics: IEEE
 * binary32, round-to-nearest-even, NO fused multiply-add (build with -ffp-contract=off).
 * It is anchored on the reference's own call sites:
 *     cv::equalizeHist(src,dst)        OpenCVequalHist.cpp:145, nextimprovement.cpp:168,
 *                                       AirplanMP4.cpp:90, 1frameMeasure.cpp:44
 *     cv::createCLAHE(clip,Size(t,t))  clahevideo.cpp:184/:497, clahe1frame.cpp:88
 *     CLAHE::apply(src,dst)            clahevideo.cpp:195, clahe1frame.cpp:93, CLAHECompare.cpp:150
 *     NV12 rebuild (UV=128 / UV copy)  OpenCVequalHist.cpp:160-162, ColoropenCVCwqualHist.cpp:165
 * and guarded by the hand-derived known-answer vectors of SURVEY.md Appendix B
 * (tests/golden/kat.json) and by tests/test_oracle_vs_opencv.py, which compares against a real
 * cv2 wherever one is importable (none is in this image).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_BAD_ARG 1
#define ORC_OOM 4

/* core/fast_math.hpp cvRound(float): _mm_cvtss_si32 -> nearest, ties to even under the default
 * MXCSR rounding mode.  lrintf() is the same conversion under FE_TONEAREST. */
static inline int orc_round(float v) { return (int)lrintf(v); }

/* core/fast_math.hpp cvFloor(float): i = (int)v; return i - (i > v). */
static inline int orc_floor(float v) { int i = (int)v; return i - ((float)i > v); }

/* core/saturate.hpp saturate_cast<uchar>(int) */
static inline uint8_t orc_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* core/src/copy.cpp borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int orc_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p - 1 + 1;
        else       p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* Floating-point contraction of the CLAHE interpolation (test infrastructure switch, process-wide).
 * 0 (default): every multiply and add rounded separately -- an x86-64 baseline build of OpenCV (no FMA).
 * 1: the fused multiply-adds GCC forms from clahe.cpp's expressions when the target has FMA and -ffp-contract=fast is in
 *    force (GCC's default outside ISO mode) -- what a distribution build of OpenCV 4.4 for aarch64 computes, and the
 *    reference's own binaries ARE aarch64 (ZCU106).  The pattern is decided in GCC's target-independent widening_mul pass
 *    and was read off `g++ -O2 -mfma -ffp-contract=fast -S` on the expressions below (tests/test_oracle.py re-checks it
 *    against such a build whenever the host CPU has FMA):
 *        txf = fma(x, inv_tw, -0.5f);   tyf = fma(y, inv_th, -0.5f);
 *        res = fma(fma(l11, xa1, l12 * xa), ya1, fma(l21, xa1, l22 * xa) * ya);
 *    sum * lutScale and sum * scale are single multiplies and do not change. */
static int g_fp_contract = 0;
int orc_set_fp_contract(int on) { const int old = g_fp_contract; if (on >= 0) g_fp_contract = on != 0; return old; }

static inline float orc_tf(int p, float inv)                       /* p * inv - 0.5f */
{
    return g_fp_contract ? fmaf((float)p, inv, -0.5f) : (float)p * inv - 0.5f;
}
static inline float orc_blend(float l11, float l12, float l21, float l22, float xa, float xa1, float ya, float ya1)
{
    if (g_fp_contract) return fmaf(fmaf(l11, xa1, l12 * xa), ya1, fmaf(l21, xa1, l22 * xa) * ya);
    return (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
}

/* test hooks: the two scalar expressions on their own (tests/test_oracle.py compares them with a GCC build that contracts) */
float orc_probe_tf(int p, float inv) { return orc_tf(p, inv); }
float orc_probe_blend(int l11, int l12, int l21, int l22, float xa, float ya)
{
    return orc_blend((float)l11, (float)l12, (float)l21, (float)l22, xa, 1.0f - xa, ya, 1.0f - ya);
}

/* ------------------------------------------------------------------------------------------
 * Stage A2: histogram of a CV_8UC1 image that honours `step`.
 * histogram.cpp EqualizeHistCalcHist_Invoker: exact int32 counts; threading is row-striped and
 * does not affect the result.
 * ---------------------------------------------------------------------------------------- */
int orc_hist_u8(const uint8_t* src, size_t step, int width, int height, int32_t hist[256])
{
    if (!hist || width < 0 || height < 0) return ORC_BAD_ARG;
    memset(hist, 0, 256 * sizeof(int32_t));
    if (width == 0 || height == 0) return ORC_OK;
    if (!src || step < (size_t)width) return ORC_BAD_ARG;
#ifdef _OPENMP
#pragma omp parallel
    {
        int32_t local[256];
        memset(local, 0, sizeof local);
#pragma omp for schedule(static) nowait
        for (int y = 0; y < height; ++y) {
            const uint8_t* p = src + (size_t)y * step;
            for (int x = 0; x < width; ++x) local[p[x]]++;
        }
#pragma omp critical
        for (int i = 0; i < 256; ++i) hist[i] += local[i];
    }
#else
    for (int y = 0; y < height; ++y) {
        const uint8_t* p = src + (size_t)y * step;
        for (int x = 0; x < width; ++x) hist[p[x]]++;
    }
#endif
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Stage A3: CDF -> LUT.   histogram.cpp cv::equalizeHist, after the histogram:
 *     i = first non-zero bin; if hist[i]==total -> constant image (dst.setTo(i));
 *     scale = (hist_sz - 1.f)/(total - hist[i]);  sum = 0;
 *     for (lut[i++] = 0; i < 256; ++i) { sum += hist[i]; lut[i] = saturate_cast<uchar>(sum*scale); }
 * Entries below the first non-zero bin are never read by the apply stage; they are reported as 0
 * here.  For a constant image every entry is set to i (equivalent to dst.setTo(i)).
 * Returns the index of the first non-zero bin through *first (or -1 for total==0).
 * ---------------------------------------------------------------------------------------- */
int orc_equalize_lut(const int32_t hist[256], int64_t total, uint8_t lut[256], int* first)
{
    if (!hist || !lut) return ORC_BAD_ARG;
    memset(lut, 0, 256);
    if (first) *first = -1;
    if (total <= 0) return ORC_OK;
    int i = 0;
    while (i < 256 && !hist[i]) ++i;
    if (i == 256) return ORC_BAD_ARG;
    if (first) *first = i;
    int itotal = (int)total;                  /* histogram.cpp: int total = (int)src.total() */
    if (hist[i] == itotal) { memset(lut, i, 256); return ORC_OK; }
    float scale = 255.0f / (float)(itotal - hist[i]);
    int sum = 0;
    for (lut[i++] = 0; i < 256; ++i) {
        sum += hist[i];
        lut[i] = orc_sat_u8(orc_round((float)sum * scale));
    }
    return ORC_OK;
}

/* Stage A4: dst(y,x) = lut[src(y,x)]; src/dst may alias (in place) and have step >= width. */
int orc_lut_apply_u8(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                     int width, int height, const uint8_t lut[256])
{
    if (width < 0 || height < 0) return ORC_BAD_ARG;
    if (width == 0 || height == 0) return ORC_OK;
    if (!src || !dst || !lut || src_step < (size_t)width || dst_step < (size_t)width) return ORC_BAD_ARG;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < width; ++x) d[x] = lut[s[x]];
    }
    return ORC_OK;
}

/* cv::equalizeHist on CV_8UC1 (A2+A3+A4). Empty image -> no-op. */
int orc_equalize_hist_u8(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                         int width, int height)
{
    int32_t hist[256];
    uint8_t lut[256];
    if (width < 0 || height < 0) return ORC_BAD_ARG;
    if (width == 0 || height == 0) return ORC_OK;
    int rc = orc_hist_u8(src, src_step, width, height, hist);
    if (rc) return rc;
    rc = orc_equalize_lut(hist, (int64_t)width * height, lut, NULL);
    if (rc) return rc;
    return orc_lut_apply_u8(src, src_step, dst, dst_step, width, height, lut);
}

/* ------------------------------------------------------------------------------------------
 * CLAHE geometry (clahe.cpp CLAHE_Impl::apply, first block).  When either axis is not divisible
 * BOTH pads are applied: right = tilesX - (W % tilesX), bottom = tilesY - (H % tilesY), so a
 * divisible axis gains a whole extra `tiles` pixels (KAT CL-3).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int ext_w, ext_h;      /* size of the image the LUT stage sees */
    int tile_w, tile_h;
    int clip;              /* integer clip limit, 0 = no clipping */
    float lut_scale;       /* 255.f / tile area */
} orc_clahe_geom;

int orc_clahe_geometry(int width, int height, double clip_limit, int tiles_x, int tiles_y,
                       orc_clahe_geom* g)
{
    if (!g || width <= 0 || height <= 0 || tiles_x <= 0 || tiles_y <= 0) return ORC_BAD_ARG;
    if (width % tiles_x == 0 && height % tiles_y == 0) {
        g->ext_w = width; g->ext_h = height;
    } else {
        g->ext_w = width + (tiles_x - (width % tiles_x));
        g->ext_h = height + (tiles_y - (height % tiles_y));
    }
    g->tile_w = g->ext_w / tiles_x;
    g->tile_h = g->ext_h / tiles_y;
    const int area = g->tile_w * g->tile_h;
    g->lut_scale = (float)(256 - 1) / (float)area;        /* static_cast<float>(histSize-1)/tileSizeTotal */
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * area / 256);             /* double math, truncation */
        if (clip < 1) clip = 1;
    }
    g->clip = clip;
    return ORC_OK;
}

/* clahe.cpp CLAHE_CalcLut_Body<uchar,256,0>::operator(): per tile histogram (over the REFLECT_101
 * extended image, read here by index reflection), clip, single-pass redistribute with
 * residualStep, cumulative sum -> uchar LUT.  luts = tiles_y*tiles_x rows of 256 bytes. */
int orc_clahe_tile_luts(const uint8_t* src, size_t step, int width, int height,
                        double clip_limit, int tiles_x, int tiles_y, uint8_t* luts)
{
    orc_clahe_geom g;
    int rc = orc_clahe_geometry(width, height, clip_limit, tiles_x, tiles_y, &g);
    if (rc) return rc;
    if (!src || !luts || step < (size_t)width) return ORC_BAD_ARG;
    const int ntiles = tiles_x * tiles_y;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int k = 0; k < ntiles; ++k) {
        const int ty = k / tiles_x, tx = k % tiles_x;
        int h[256];
        memset(h, 0, sizeof h);
        for (int yy = 0; yy < g.tile_h; ++yy) {
            const int y = orc_reflect101(ty * g.tile_h + yy, height);
            const uint8_t* row = src + (size_t)y * step;
            const int x0 = tx * g.tile_w;
            if (x0 + g.tile_w <= width) {
                for (int xx = 0; xx < g.tile_w; ++xx) h[row[x0 + xx]]++;
            } else {
                for (int xx = 0; xx < g.tile_w; ++xx) h[row[orc_reflect101(x0 + xx, width)]]++;
            }
        }
        if (g.clip > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; ++i)
                if (h[i] > g.clip) { clipped += h[i] - g.clip; h[i] = g.clip; }
            int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) h[i] += batch;
            if (residual != 0) {
                int rstep = 256 / residual; if (rstep < 1) rstep = 1;
                for (int i = 0; i < 256 && residual > 0; i += rstep, residual--) h[i]++;
            }
        }
        int sum = 0;
        uint8_t* lut = luts + (size_t)k * 256;
        for (int i = 0; i < 256; ++i) {
            sum += h[i];
            lut[i] = orc_sat_u8(orc_round((float)sum * g.lut_scale));
        }
    }
    return ORC_OK;
}

/* clahe.cpp CLAHE_Interpolation_Body<uchar,0>: column tables from the constructor, row weights
 * and the nine individually rounded float ops from operator().  Weights are computed BEFORE the
 * tile indices are clamped. */
int orc_clahe_interpolate(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                          int width, int height, int tiles_x, int tiles_y,
                          int tile_w, int tile_h, const uint8_t* luts)
{
    if (width <= 0 || height <= 0) return ORC_BAD_ARG;
    if (!src || !dst || !luts || src_step < (size_t)width || dst_step < (size_t)width) return ORC_BAD_ARG;
    int* ind1 = (int*)malloc(sizeof(int) * (size_t)width * 2);
    float* xa = (float*)malloc(sizeof(float) * (size_t)width * 2);
    if (!ind1 || !xa) { free(ind1); free(xa); return ORC_OOM; }
    int* ind2 = ind1 + width;
    float* xa1 = xa + width;
    const float inv_tw = 1.0f / (float)tile_w;
    for (int x = 0; x < width; ++x) {
        float txf = orc_tf(x, inv_tw);
        int tx1 = orc_floor(txf);
        int tx2 = tx1 + 1;
        xa[x] = txf - (float)tx1;
        xa1[x] = 1.0f - xa[x];
        if (tx1 < 0) tx1 = 0;
        if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
        ind1[x] = tx1 * 256;
        ind2[x] = tx2 * 256;
    }
    const float inv_th = 1.0f / (float)tile_h;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        float tyf = orc_tf(y, inv_th);
        int ty1 = orc_floor(tyf);
        int ty2 = ty1 + 1;
        float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const uint8_t* p1 = luts + (size_t)ty1 * tiles_x * 256;
        const uint8_t* p2 = luts + (size_t)ty2 * tiles_x * 256;
        for (int x = 0; x < width; ++x) {
            int v = s[x];
            int i1 = ind1[x] + v, i2 = ind2[x] + v;
            float res = orc_blend((float)p1[i1], (float)p1[i2], (float)p2[i1], (float)p2[i2], xa[x], xa1[x], ya, ya1);
            d[x] = orc_sat_u8(orc_round(res));
        }
    }
    free(ind1); free(xa);
    return ORC_OK;
}

/* cv::CLAHE::apply on CV_8UC1.  src/dst may alias: like the original (which reads `src` in the
 * interpolation pass and writes `dst` pixel by pixel after the LUTs are complete) an in-place
 * call is well defined because each output pixel depends only on the same input pixel + LUTs. */
int orc_clahe_u8(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step,
                 int width, int height, double clip_limit, int tiles_x, int tiles_y)
{
    if (width < 0 || height < 0 || tiles_x <= 0 || tiles_y <= 0) return ORC_BAD_ARG;
    if (width == 0 || height == 0) return ORC_OK;
    orc_clahe_geom g;
    int rc = orc_clahe_geometry(width, height, clip_limit, tiles_x, tiles_y, &g);
    if (rc) return rc;
    uint8_t* luts = (uint8_t*)malloc((size_t)tiles_x * tiles_y * 256);
    if (!luts) return ORC_OOM;
    rc = orc_clahe_tile_luts(src, src_step, width, height, clip_limit, tiles_x, tiles_y, luts);
    if (!rc)
        rc = orc_clahe_interpolate(src, src_step, dst, dst_step, width, height, tiles_x, tiles_y,
                                   g.tile_w, g.tile_h, luts);
    free(luts);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * CLAHE on CV_16UC1 (clahe.cpp CLAHE_CalcLut_Body<ushort,65536,0>,
 * CLAHE_Interpolation_Body<ushort,0>): the same algorithm with histSize = 65536, lutScale =
 * 65535.f/area, clip = max((int)(clipLimit*area/65536), 1), ushort LUTs, saturate_cast<ushort>.
 * Steps in BYTES.  Not used by the reference (OpenCV surface beyond it).
 * ---------------------------------------------------------------------------------------- */
static inline uint16_t orc_sat_u16(int v) { return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); }

int orc_clahe_u16(const uint16_t* src, size_t src_step, uint16_t* dst, size_t dst_step,
                  int width, int height, double clip_limit, int tiles_x, int tiles_y)
{
    enum { HS = 65536 };
    if (width < 0 || height < 0 || tiles_x <= 0 || tiles_y <= 0) return ORC_BAD_ARG;
    if (width == 0 || height == 0) return ORC_OK;
    if (!src || !dst || src_step < (size_t)width * 2 || dst_step < (size_t)width * 2) return ORC_BAD_ARG;
    int ext_w = width, ext_h = height;
    if (width % tiles_x != 0 || height % tiles_y != 0) {
        ext_w = width + (tiles_x - width % tiles_x);
        ext_h = height + (tiles_y - height % tiles_y);
    }
    const int tile_w = ext_w / tiles_x, tile_h = ext_h / tiles_y;
    const int area = tile_w * tile_h;
    const float lut_scale = (float)(HS - 1) / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * area / HS); if (clip < 1) clip = 1; }
    const int ntiles = tiles_x * tiles_y;
    uint16_t* luts = (uint16_t*)malloc((size_t)ntiles * HS * sizeof(uint16_t));
    if (!luts) return ORC_OOM;
    int oom = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int k = 0; k < ntiles; ++k) {
        int* h = (int*)calloc(HS, sizeof(int));
        if (!h) { oom = 1; continue; }
        const int ty = k / tiles_x, tx = k % tiles_x;
        for (int yy = 0; yy < tile_h; ++yy) {
            const int y = orc_reflect101(ty * tile_h + yy, height);
            const uint16_t* row = (const uint16_t*)((const uint8_t*)src + (size_t)y * src_step);
            for (int xx = 0; xx < tile_w; ++xx) h[row[orc_reflect101(tx * tile_w + xx, width)]]++;
        }
        if (clip > 0) {
            int clipped = 0;
            for (int i = 0; i < HS; ++i) if (h[i] > clip) { clipped += h[i] - clip; h[i] = clip; }
            int batch = clipped / HS, residual = clipped - batch * HS;
            for (int i = 0; i < HS; ++i) h[i] += batch;
            if (residual != 0) {
                int rstep = HS / residual; if (rstep < 1) rstep = 1;
                for (int i = 0; i < HS && residual > 0; i += rstep, residual--) h[i]++;
            }
        }
        int sum = 0;
        uint16_t* lut = luts + (size_t)k * HS;
        for (int i = 0; i < HS; ++i) { sum += h[i]; lut[i] = orc_sat_u16(orc_round((float)sum * lut_scale)); }
        free(h);
    }
    if (oom) { free(luts); return ORC_OOM; }
    const float inv_tw = 1.0f / (float)tile_w, inv_th = 1.0f / (float)tile_h;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < height; ++y) {
        const uint16_t* s = (const uint16_t*)((const uint8_t*)src + (size_t)y * src_step);
        uint16_t* d = (uint16_t*)((uint8_t*)dst + (size_t)y * dst_step);
        float tyf = orc_tf(y, inv_th);
        int ty1 = orc_floor(tyf), ty2 = ty1 + 1;
        float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const uint16_t* p1 = luts + (size_t)ty1 * tiles_x * HS;
        const uint16_t* p2 = luts + (size_t)ty2 * tiles_x * HS;
        for (int x = 0; x < width; ++x) {
            float txf = orc_tf(x, inv_tw);
            int tx1 = orc_floor(txf), tx2 = tx1 + 1;
            float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            const int v = s[x];
            float res = orc_blend((float)p1[(size_t)tx1 * HS + v], (float)p1[(size_t)tx2 * HS + v],
                                  (float)p2[(size_t)tx1 * HS + v], (float)p2[(size_t)tx2 * HS + v], xa, xa1, ya, ya1);
            d[x] = orc_sat_u16(orc_round(res));
        }
    }
    free(luts);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * A7: whole NV12 frame = op on Y + UV fill(128) (OpenCVequalHist.cpp:160-162,
 * clahevideo.cpp:200-201) or UV passthrough (ColoropenCVCwqualHist.cpp:165, improvement.cpp:163,
 * nextimprovement.cpp:160).  Tightly packed: Y = W*H bytes, UV = W*H/2 bytes (integer division
 * exactly as the callers compute uv_size, OpenCVequalHist.cpp:130).
 * uv_mode: 0 = fill 128, 1 = copy.   op: 0 = equalizeHist, 1 = CLAHE.
 * ---------------------------------------------------------------------------------------- */
int orc_nv12_frame(const uint8_t* in, uint8_t* out, int width, int height, int uv_mode, int op,
                   double clip_limit, int tiles_x, int tiles_y)
{
    if (width < 0 || height < 0 || (uv_mode != 0 && uv_mode != 1)) return ORC_BAD_ARG;
    if (width == 0 || height == 0) return ORC_OK;
    if (!in || !out) return ORC_BAD_ARG;
    const size_t y_size = (size_t)width * (size_t)height;
    const size_t uv_size = y_size / 2;
    int rc;
    if (op == 0) rc = orc_equalize_hist_u8(in, (size_t)width, out, (size_t)width, width, height);
    else         rc = orc_clahe_u8(in, (size_t)width, out, (size_t)width, width, height,
                                   clip_limit, tiles_x, tiles_y);
    if (rc) return rc;
    if (uv_mode == 0) memset(out + y_size, 128, uv_size);
    else if (out != in) memmove(out + y_size, in + y_size, uv_size);
    return ORC_OK;
}
