// color.hip.h -- cvtColor BGR2YUV / YUV2BGR and fused split/merge
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "common.hip.h"

namespace mi {
// =============================================================================================
// Colour-domain neighbours of the path: cv::cvtColor(COLOR_BGR2YUV / COLOR_YUV2BGR) on CV_8UC3
// and the split / merge around the luma op (singlecolor.cpp:39-66, clahe1frame.cpp:83-102).
// OpenCV 4.4 color_yuv.simd.hpp, 8-bit fixed point (yuv_shift = 14), restated in oracle/color_oracle.c:
//   Y = DESCALE(B*1868 + G*9617 + R*4899), U = DESCALE((B-Y)*8061 + (128<<14)), V = DESCALE((R-Y)*14369 + (128<<14))
//   B = Y + DESCALE((U-128)*33292), G = Y + DESCALE((U-128)*-6472 + (V-128)*-9519), R = Y + DESCALE((V-128)*18678)
// Pure integer work, 3 B/px streams: bound by HBM.  A lane handles 16 pixels = 3 x 16 B of interleaved data.
// =============================================================================================
struct ColorJob {
    const uint8_t* src; uint8_t* dst;        // interleaved CV_8UC3 side (src for MODE 0/1/2, dst for 0/1/3)
    long long src_step, dst_step;            // bytes between rows (interleaved side(s))
    long long src_frame, dst_frame;
    uint8_t* p0; uint8_t* p1; uint8_t* p2;   // planes (MODE 2: outputs Y,U,V; MODE 3: inputs Y,U,V), tightly packed W*H each
    long long plane_frame;                   // bytes between frames of each plane
    long long row_px;                        // pixels per row (contiguous images: W*H with rows == 1)
    int rows;
};

// CV_DESCALE(x, 14).  The empty asm keeps the shifted value opaque: hipcc (ROCm 7.2) otherwise folds pairs of
// "arithmetic shift right -> clamp to [0,255] -> pack" into gfx950's v_ashr_pk_u8_i32, and that lowering produced
// wrong bytes for the V plane here (tools/dbg_color.hip reproduces it: neighbouring bytes get OR-ed together).
__device__ __forceinline__ int yuv_descale(int x) { int r = (x + (1 << 13)) >> 14; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ uint32_t sat_u8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// (r - y) and (b - y) lie in [-255, 255] -- the three luma weights add up to 1 << 14, so y is a weighted mean of three bytes -- but the
// opaque descale hides that from the compiler, which then multiplies with v_mul_lo_u32 (quarter rate).  __mul24 says it: v_mad_i32_i24.
__device__ __forceinline__ void px_bgr2yuv(uint32_t b, uint32_t g, uint32_t r, uint32_t& Y, uint32_t& U, uint32_t& V)
{
    const int y = yuv_descale((int)b * 1868 + (int)g * 9617 + (int)r * 4899);
    V = sat_u8(yuv_descale(__mul24((int)r - y, 14369) + (128 << 14)));
    U = sat_u8(yuv_descale(__mul24((int)b - y, 8061) + (128 << 14)));
    Y = sat_u8(y);
}
// luma alone (the histogram pass needs nothing else; the descales above are volatile to the optimiser and would not be dropped)
__device__ __forceinline__ uint32_t px_bgr2y(uint32_t b, uint32_t g, uint32_t r)
{
    return sat_u8(yuv_descale((int)b * 1868 + (int)g * 9617 + (int)r * 4899));
}
__device__ __forceinline__ void px_yuv2bgr(uint32_t Y, uint32_t U, uint32_t V, uint32_t& b, uint32_t& g, uint32_t& r)
{
    const int u = (int)U - 128, v = (int)V - 128;
    b = sat_u8((int)Y + yuv_descale(u * 33292));
    g = sat_u8((int)Y + yuv_descale(u * -6472 + v * -9519));
    r = sat_u8((int)Y + yuv_descale(v * 18678));
}

// MODE 0: BGR -> YUV interleaved.  1: YUV -> BGR interleaved.  2: BGR -> planes Y,U,V (cvtColor + split).
// 3: planes Y,U,V -> BGR (merge + cvtColor).   grid = (blocks, min(rows, 65535), frames)
template <int MODE>
__global__ __launch_bounds__(kThreads) void color_kernel(ColorJob j)
{
    const int f = blockIdx.z;
    for (int row = blockIdx.y; row < j.rows; row += gridDim.y) {
        const uint8_t* s3 = MODE != 3 ? j.src + (long long)f * j.src_frame + (long long)row * j.src_step : nullptr;
        uint8_t* d3 = MODE != 2 ? j.dst + (long long)f * j.dst_frame + (long long)row * j.dst_step : nullptr;
        const long long poff = (long long)f * j.plane_frame + (long long)row * j.row_px;
        const long long groups = j.row_px >> 4;
        const bool a3s = MODE == 3 || (((uintptr_t)s3 & 15) == 0), a3d = MODE == 2 || (((uintptr_t)d3 & 15) == 0);
        const bool ap = MODE < 2 || ((((uintptr_t)j.p0 | (uintptr_t)j.p1 | (uintptr_t)j.p2 | (uintptr_t)poff) & 15) == 0);
        if (a3s && a3d && ap) {
            for (long long gidx = (long long)blockIdx.x * kThreads + threadIdx.x; gidx < groups; gidx += (long long)gridDim.x * kThreads) {
                uint32_t c0[16], c1[16], c2[16];            // channel values of 16 pixels
                if (MODE != 3) {
                    const u32x4* sp = reinterpret_cast<const u32x4*>(s3 + gidx * 48);
                    const u32x4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
                    const uint32_t w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        c0[p] = (w[(3 * p) >> 2] >> (8 * ((3 * p) & 3))) & 0xffu;
                        c1[p] = (w[(3 * p + 1) >> 2] >> (8 * ((3 * p + 1) & 3))) & 0xffu;
                        c2[p] = (w[(3 * p + 2) >> 2] >> (8 * ((3 * p + 2) & 3))) & 0xffu;
                    }
                } else {
                    const u32x4 y = *reinterpret_cast<const u32x4*>(j.p0 + poff + gidx * 16);
                    const u32x4 u = *reinterpret_cast<const u32x4*>(j.p1 + poff + gidx * 16);
                    const u32x4 v = *reinterpret_cast<const u32x4*>(j.p2 + poff + gidx * 16);
                    const uint32_t wy[4] = {y.x, y.y, y.z, y.w}, wu[4] = {u.x, u.y, u.z, u.w}, wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        c0[p] = (wy[p >> 2] >> (8 * (p & 3))) & 0xffu;
                        c1[p] = (wu[p >> 2] >> (8 * (p & 3))) & 0xffu;
                        c2[p] = (wv[p >> 2] >> (8 * (p & 3))) & 0xffu;
                    }
                }
                uint32_t o0[16], o1[16], o2[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    if (MODE == 0 || MODE == 2) px_bgr2yuv(c0[p], c1[p], c2[p], o0[p], o1[p], o2[p]);
                    else px_yuv2bgr(c0[p], c1[p], c2[p], o0[p], o1[p], o2[p]);
                }
                if (MODE != 2) {
                    uint32_t w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        w[(3 * p) >> 2] |= o0[p] << (8 * ((3 * p) & 3));
                        w[(3 * p + 1) >> 2] |= o1[p] << (8 * ((3 * p + 1) & 3));
                        w[(3 * p + 2) >> 2] |= o2[p] << (8 * ((3 * p + 2) & 3));
                    }
                    u32x4* dp = reinterpret_cast<u32x4*>(d3 + gidx * 48);
                    const u32x4 r0 = {w[0], w[1], w[2], w[3]}, r1 = {w[4], w[5], w[6], w[7]}, r2 = {w[8], w[9], w[10], w[11]};
                    dp[0] = r0; dp[1] = r1; dp[2] = r2;
                } else {
                    uint32_t wy[4], wu[4], wv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        wy[k] = o0[4 * k] | (o0[4 * k + 1] << 8) | (o0[4 * k + 2] << 16) | (o0[4 * k + 3] << 24);
                        wu[k] = o1[4 * k] | (o1[4 * k + 1] << 8) | (o1[4 * k + 2] << 16) | (o1[4 * k + 3] << 24);
                        wv[k] = o2[4 * k] | (o2[4 * k + 1] << 8) | (o2[4 * k + 2] << 16) | (o2[4 * k + 3] << 24);
                    }
                    const u32x4 ry = {wy[0], wy[1], wy[2], wy[3]}, ru = {wu[0], wu[1], wu[2], wu[3]}, rv = {wv[0], wv[1], wv[2], wv[3]};
                    *reinterpret_cast<u32x4*>(j.p0 + poff + gidx * 16) = ry;
                    *reinterpret_cast<u32x4*>(j.p1 + poff + gidx * 16) = ru;
                    *reinterpret_cast<u32x4*>(j.p2 + poff + gidx * 16) = rv;
                }
            }
        }
        // ragged tail of the row (or the whole row when something is not 16-B aligned): one pixel per lane
        const long long first = (a3s && a3d && ap) ? (groups << 4) : 0;
        for (long long x = first + (long long)blockIdx.x * kThreads + threadIdx.x; x < j.row_px; x += (long long)gridDim.x * kThreads) {
            uint32_t a, b, c, o0, o1, o2;
            if (MODE != 3) { a = s3[3 * x]; b = s3[3 * x + 1]; c = s3[3 * x + 2]; }
            else { a = j.p0[poff + x]; b = j.p1[poff + x]; c = j.p2[poff + x]; }
            if (MODE == 0 || MODE == 2) px_bgr2yuv(a, b, c, o0, o1, o2); else px_yuv2bgr(a, b, c, o0, o1, o2);
            if (MODE != 2) { d3[3 * x] = (uint8_t)o0; d3[3 * x + 1] = (uint8_t)o1; d3[3 * x + 2] = (uint8_t)o2; }
            else { j.p0[poff + x] = (uint8_t)o0; j.p1[poff + x] = (uint8_t)o1; j.p2[poff + x] = (uint8_t)o2; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused BGR luma equalization (singlecolor.cpp:39-66 in two passes over the interleaved image, 9 B/px instead of the
// 14 B/px of the planar pipeline): pass 1 converts on the fly and histograms Y; pass 2 converts, maps Y through the
// frame's LUT and converts back.  Same arithmetic per pixel as cvtColor -> split -> equalizeHist -> merge -> cvtColor
// (U and V are the saturated 8-bit values the reference would have stored in between).
// grid = (B, 1, n_frames); every workgroup walks its share of the 16-pixel groups of every row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_bgr16(const uint8_t* p, uint32_t* c0, uint32_t* c1, uint32_t* c2)
{
    const u32x4* sp = reinterpret_cast<const u32x4*>(p);
    const u32x4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
    const uint32_t w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
    for (int px = 0; px < 16; ++px) {
        c0[px] = (w[(3 * px) >> 2] >> (8 * ((3 * px) & 3))) & 0xffu;
        c1[px] = (w[(3 * px + 1) >> 2] >> (8 * ((3 * px + 1) & 3))) & 0xffu;
        c2[px] = (w[(3 * px + 2) >> 2] >> (8 * ((3 * px + 2) & 3))) & 0xffu;
    }
}

// 512-thread workgroups for both passes: eight waves share one 32 KiB LDS table (histogram / replicated LUT), 24 waves per CU at
// ~70 VGPRs instead of 20.
constexpr int kBgrThreads = 512;
__global__ __launch_bounds__(kBgrThreads) void bgr_luma_hist_kernel(ColorJob j, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[256 * kCopies];
    const int f = blockIdx.z, t = threadIdx.x;
    for (int i = t; i < 256 * kCopies; i += kBgrThreads) h[i] = 0;
    __syncthreads();
    const uint32_t copy = t & (kCopies - 1);
    for (int row = 0; row < j.rows; ++row) {
        const uint8_t* s3 = j.src + (long long)f * j.src_frame + (long long)row * j.src_step;
        const long long groups = (((uintptr_t)s3 & 15) == 0) ? (j.row_px >> 4) : 0;
        for (long long gidx = (long long)blockIdx.x * kBgrThreads + t; gidx < groups; gidx += (long long)gridDim.x * kBgrThreads) {
            uint32_t c0[16], c1[16], c2[16];
            load_bgr16(s3 + gidx * 48, c0, c1, c2);
#pragma unroll
            for (int px = 0; px < 16; ++px) lds_inc(h, (px_bgr2y(c0[px], c1[px], c2[px]) << kCopyShift) + copy);
        }
        for (long long x = (groups << 4) + (long long)blockIdx.x * kBgrThreads + t; x < j.row_px; x += (long long)gridDim.x * kBgrThreads)
            lds_inc(h, (px_bgr2y(s3[3 * x], s3[3 * x + 1], s3[3 * x + 2]) << kCopyShift) + copy);
    }
    __syncthreads();
    if (t < 256) partial[((size_t)f * gridDim.x + blockIdx.x) * 256 + t] = lds_hist_bin(h, t);
}

__global__ __launch_bounds__(kBgrThreads) void bgr_luma_apply_kernel(ColorJob j, const uint8_t* __restrict__ luts)
{
    __shared__ uint32_t lut[256 * kCopies];
    const int f = (int)gridDim.z - 1 - (int)blockIdx.z, t = threadIdx.x;      // last-to-first: what pass 1 read last is still cached
    const uint32_t copy = t & (kCopies - 1);
    if (t < 256) {
        const uint32_t v = luts[(size_t)f * 256 + t];
#pragma unroll
        for (int k = 0; k < kCopies; ++k) lut[(t << kCopyShift) + ((k + t) & (kCopies - 1))] = v;
    }
    __syncthreads();
    for (int row = 0; row < j.rows; ++row) {
        const uint8_t* s3 = j.src + (long long)f * j.src_frame + (long long)row * j.src_step;
        uint8_t* d3 = j.dst + (long long)f * j.dst_frame + (long long)row * j.dst_step;
        const long long groups = ((((uintptr_t)s3 | (uintptr_t)d3) & 15) == 0) ? (j.row_px >> 4) : 0;
        for (long long gidx = (long long)blockIdx.x * kBgrThreads + t; gidx < groups; gidx += (long long)gridDim.x * kBgrThreads) {
            uint32_t c0[16], c1[16], c2[16];
            load_bgr16(s3 + gidx * 48, c0, c1, c2);
            uint32_t w[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) w[k] = 0;
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                uint32_t Y, U, V, b, g, r;
                px_bgr2yuv(c0[px], c1[px], c2[px], Y, U, V);
                px_yuv2bgr(lut[(Y << kCopyShift) + copy], U, V, b, g, r);
                w[(3 * px) >> 2] |= b << (8 * ((3 * px) & 3));
                w[(3 * px + 1) >> 2] |= g << (8 * ((3 * px + 1) & 3));
                w[(3 * px + 2) >> 2] |= r << (8 * ((3 * px + 2) & 3));
            }
            u32x4* dp = reinterpret_cast<u32x4*>(d3 + gidx * 48);
            const u32x4 r0 = {w[0], w[1], w[2], w[3]}, r1 = {w[4], w[5], w[6], w[7]}, r2 = {w[8], w[9], w[10], w[11]};
            dp[0] = r0; dp[1] = r1; dp[2] = r2;
        }
        for (long long x = (groups << 4) + (long long)blockIdx.x * kBgrThreads + t; x < j.row_px; x += (long long)gridDim.x * kBgrThreads) {
            uint32_t Y, U, V, b, g, r;
            px_bgr2yuv(s3[3 * x], s3[3 * x + 1], s3[3 * x + 2], Y, U, V);
            px_yuv2bgr(lut[(Y << kCopyShift) + copy], U, V, b, g, r);
            d3[3 * x] = (uint8_t)b; d3[3 * x + 1] = (uint8_t)g; d3[3 * x + 2] = (uint8_t)r;
        }
    }
}

// =============================================================================================
// BASELINE.json config 5 read literally (second half): NV12 -> BGR -> equalizeHist on B, G and R ->
// NV12.  No file of the reference does this; the OpenCV 4.4 pipeline it stands for and its 4:2:0 fixed-point
// arithmetic (ITUR_BT_601_*, shift 20) are restated in oracle/color_oracle.c (orc_nv12_bgr_equalize; parity unpinned).
// The BGR image is never materialised: pass 1 decodes on the fly and builds the three channel histograms, pass 2
// decodes again, maps each channel through its LUT and encodes straight back to NV12 -- 4.5 B/px of HBM traffic
// (1.5 read twice, 1.5 written) instead of the 16.5 B/px of the literal cvtColor/split/merge sequence.
// A lane owns a 16 x 2 pixel group: two 16-B luma loads + one 16-B chroma load (8 U,V pairs).
// =============================================================================================
struct Nv12Job {
    const uint8_t* in; uint8_t* out;          // tight NV12 frames: W*H luma bytes, then H/2 rows of W interleaved U,V bytes
    long long in_frame, out_frame;            // bytes between frames
    int width, height;                        // both even
    int vec;                                  // 1: W % 16 == 0 and every row start is 16-B aligned -> vector path
};

constexpr int kChCopies = 16, kChCopyShift = 4;     // 3 channel histograms x 256 bins x 16 copies = 48 KiB of LDS

// (x >> 20) kept opaque for the same reason as yuv_descale(): no v_ashr_pk_u8_i32 pairing.
__device__ __forceinline__ int bt_shift(int x) { int r = x >> 20; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ void bt601_uv_terms(uint32_t U, uint32_t V, int& ruv, int& guv, int& buv)
{
    const int uu = (int)U - 128, vv = (int)V - 128;
    ruv = (1 << 19) + 1673527 * vv;
    guv = (1 << 19) - 852492 * vv - 409993 * uu;
    buv = (1 << 19) + 2116026 * uu;
}
__device__ __forceinline__ void bt601_px_bgr(uint32_t Y, int ruv, int guv, int buv, uint32_t& b, uint32_t& g, uint32_t& r)
{
    const int yy = max(0, (int)Y - 16) * 1220542;
    b = sat_u8(bt_shift(yy + buv)); g = sat_u8(bt_shift(yy + guv)); r = sat_u8(bt_shift(yy + ruv));
}
__device__ __forceinline__ uint32_t bt601_y(uint32_t b, uint32_t g, uint32_t r)
{
    return sat_u8(bt_shift(269484 * (int)r + 528482 * (int)g + 102760 * (int)b + (1 << 19) + (16 << 20)));
}
__device__ __forceinline__ void bt601_uv(uint32_t b, uint32_t g, uint32_t r, uint32_t& U, uint32_t& V)
{
    U = sat_u8(bt_shift(-155188 * (int)r - 305135 * (int)g + 460324 * (int)b + (1 << 19) + (128 << 20)));
    V = sat_u8(bt_shift(460324 * (int)r - 385875 * (int)g - 74448 * (int)b + (1 << 19) + (128 << 20)));
}
__device__ __forceinline__ uint32_t byte_of(const u32x4& q, int i)       // i is a compile-time constant after unrolling
{
    const uint32_t w = i < 4 ? q.x : (i < 8 ? q.y : (i < 12 ? q.z : q.w));
    return (w >> (8 * (i & 3))) & 0xffu;
}

// Both passes run 512-thread workgroups: their LDS tables (48 KiB of channel histograms / 32 KiB of LUTs) are per workgroup, so eight
// waves sharing one table put 24 / 32 waves on a CU where 256-thread workgroups gave 12 / 20 -- these kernels wait on memory and LDS
// latency (105 s_waitcnt per 32 pixels in the apply pass), not on issue slots.
constexpr int kNv12Threads = 512;
// pass 1: partial[((f*3 + ch) * B + part) * 256 + bin], ch = 0 (B), 1 (G), 2 (R).  grid = (B, n_frames)
__global__ __launch_bounds__(kNv12Threads) void nv12_bgr_hist_kernel(Nv12Job j, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[3 * 256 * kChCopies];       // (32 copies / 1024 threads / 96 KiB, conflict-free but one workgroup per CU, measured 7 % slower)
    const int t = threadIdx.x, f = blockIdx.y;
    for (int i = t; i < 3 * 256 * kChCopies; i += kNv12Threads) h[i] = 0;
    __syncthreads();
    uint32_t* hb = h; uint32_t* hg = h + 256 * kChCopies; uint32_t* hr = h + 2 * 256 * kChCopies;
    const uint32_t copy = t & (kChCopies - 1);
    const uint8_t* yp = j.in + (long long)f * j.in_frame;
    const uint8_t* uvp = yp + (long long)j.width * j.height;
    auto add = [&](uint32_t Y, int ruv, int guv, int buv) {
        uint32_t b, g, r;
        bt601_px_bgr(Y, ruv, guv, buv, b, g, r);
        lds_inc(hb, (b << kChCopyShift) + copy); lds_inc(hg, (g << kChCopyShift) + copy); lds_inc(hr, (r << kChCopyShift) + copy);
    };
    if (j.vec) {
        const int gx_n = j.width >> 4;
        const int groups = gx_n * (j.height >> 1);            // < 2^27 (W*H < 2^31)
        // (block row, 16-pixel group) walked incrementally: the 64-bit division per 32 pixels this loop used to do cost 7 % of the pass
        const int stride = (int)gridDim.x * kNv12Threads, dby = stride / gx_n, dgx = stride - dby * gx_n;
        int gi = (int)blockIdx.x * kNv12Threads + t;
        int by = gi / gx_n, gx = gi - by * gx_n;
        for (; gi < groups; gi += stride, by += dby, gx += dgx) {
            if (gx >= gx_n) { gx -= gx_n; ++by; }
            const u32x4 y0 = *reinterpret_cast<const u32x4*>(yp + (long long)(2 * by) * j.width + (gx << 4));
            const u32x4 y1 = *reinterpret_cast<const u32x4*>(yp + (long long)(2 * by + 1) * j.width + (gx << 4));
            const u32x4 uv = *reinterpret_cast<const u32x4*>(uvp + (long long)by * j.width + (gx << 4));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int ruv, guv, buv;
                bt601_uv_terms(byte_of(uv, 2 * k), byte_of(uv, 2 * k + 1), ruv, guv, buv);
                add(byte_of(y0, 2 * k), ruv, guv, buv); add(byte_of(y0, 2 * k + 1), ruv, guv, buv);
                add(byte_of(y1, 2 * k), ruv, guv, buv); add(byte_of(y1, 2 * k + 1), ruv, guv, buv);
            }
        }
    } else {                                            // one 2x2 block per lane
        const int bx_n = j.width >> 1;
        const long long blocks = (long long)bx_n * (j.height >> 1);
        for (long long bi = (long long)blockIdx.x * kNv12Threads + t; bi < blocks; bi += (long long)gridDim.x * kNv12Threads) {
            const int by = (int)(bi / bx_n), bx = (int)(bi - (long long)by * bx_n);
            const uint8_t* r0 = yp + (long long)(2 * by) * j.width + 2 * bx;
            const uint8_t* uv = uvp + (long long)by * j.width + 2 * bx;
            int ruv, guv, buv;
            bt601_uv_terms(uv[0], uv[1], ruv, guv, buv);
            add(r0[0], ruv, guv, buv); add(r0[1], ruv, guv, buv);
            add(r0[j.width], ruv, guv, buv); add(r0[j.width + 1], ruv, guv, buv);
        }
    }
    __syncthreads();
    if (t >= 256) return;                               // one bin per thread from here on
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kChCopies; ++k) s += h[ch * 256 * kChCopies + (t << kChCopyShift) + ((k + t) & (kChCopies - 1))];
        partial[(((size_t)f * 3 + ch) * gridDim.x + blockIdx.x) * 256 + t] = s;
    }
}

// pass 2: luts[(f*3 + ch) * 256 + v].  grid = (B, n_frames).  in == out allowed (a lane reads its group before writing it).
__global__ __launch_bounds__(kNv12Threads) void nv12_bgr_apply_kernel(Nv12Job j, const uint8_t* __restrict__ luts)
{
    __shared__ uint32_t lut3[256 * kCopies];            // entry v: lutB[v] | lutG[v] << 8 | lutR[v] << 16, 32 copies
    const int t = threadIdx.x, f = (int)gridDim.y - 1 - (int)blockIdx.y;      // last-to-first: what pass 1 read last is still cached
    const uint32_t copy = t & (kCopies - 1);
    if (t < 256) {
        const uint8_t* lf = luts + (size_t)f * 3 * 256;
        const uint32_t v = (uint32_t)lf[t] | ((uint32_t)lf[256 + t] << 8) | ((uint32_t)lf[512 + t] << 16);
#pragma unroll
        for (int k = 0; k < kCopies; ++k) lut3[(t << kCopyShift) + ((k + t) & (kCopies - 1))] = v;
    }
    __syncthreads();
    const uint8_t* yp = j.in + (long long)f * j.in_frame;
    const uint8_t* uvp = yp + (long long)j.width * j.height;
    uint8_t* yo = j.out + (long long)f * j.out_frame;
    uint8_t* uvo = yo + (long long)j.width * j.height;
    // decode -> per-channel LUT -> luma of the equalized pixel; (b, g, r) returned for the chroma of a block's first pixel
    auto px = [&](uint32_t Y, int ruv, int guv, int buv, uint32_t& b, uint32_t& g, uint32_t& r) -> uint32_t {
        bt601_px_bgr(Y, ruv, guv, buv, b, g, r);
        b = lut3[(b << kCopyShift) + copy] & 0xffu;
        g = (lut3[(g << kCopyShift) + copy] >> 8) & 0xffu;
        r = (lut3[(r << kCopyShift) + copy] >> 16) & 0xffu;
        return bt601_y(b, g, r);
    };
    if (j.vec) {
        const int gx_n = j.width >> 4;
        const int groups = gx_n * (j.height >> 1);
        const int stride = (int)gridDim.x * kNv12Threads, dby = stride / gx_n, dgx = stride - dby * gx_n;
        int gi = (int)blockIdx.x * kNv12Threads + t;
        int by = gi / gx_n, gx = gi - by * gx_n;
        for (; gi < groups; gi += stride, by += dby, gx += dgx) {
            if (gx >= gx_n) { gx -= gx_n; ++by; }
            const long long o0 = (long long)(2 * by) * j.width + (gx << 4), o1 = o0 + j.width, ouv = (long long)by * j.width + (gx << 4);
            const u32x4 y0 = *reinterpret_cast<const u32x4*>(yp + o0);
            const u32x4 y1 = *reinterpret_cast<const u32x4*>(yp + o1);
            const u32x4 uv = *reinterpret_cast<const u32x4*>(uvp + ouv);
            uint32_t w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, wuv[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int ruv, guv, buv;
                bt601_uv_terms(byte_of(uv, 2 * k), byte_of(uv, 2 * k + 1), ruv, guv, buv);
                uint32_t b0, g0, r0, b, g, r, U, V;
                const uint32_t a00 = px(byte_of(y0, 2 * k), ruv, guv, buv, b0, g0, r0);
                const uint32_t a01 = px(byte_of(y0, 2 * k + 1), ruv, guv, buv, b, g, r);
                const uint32_t a10 = px(byte_of(y1, 2 * k), ruv, guv, buv, b, g, r);
                const uint32_t a11 = px(byte_of(y1, 2 * k + 1), ruv, guv, buv, b, g, r);
                bt601_uv(b0, g0, r0, U, V);
                w0[k >> 1] |= (a00 | (a01 << 8)) << (16 * (k & 1));
                w1[k >> 1] |= (a10 | (a11 << 8)) << (16 * (k & 1));
                wuv[k >> 1] |= (U | (V << 8)) << (16 * (k & 1));
            }
            const u32x4 q0 = {w0[0], w0[1], w0[2], w0[3]}, q1 = {w1[0], w1[1], w1[2], w1[3]}, q2 = {wuv[0], wuv[1], wuv[2], wuv[3]};
            *reinterpret_cast<u32x4*>(yo + o0) = q0;
            *reinterpret_cast<u32x4*>(yo + o1) = q1;
            *reinterpret_cast<u32x4*>(uvo + ouv) = q2;
        }
    } else {
        const int bx_n = j.width >> 1;
        const long long blocks = (long long)bx_n * (j.height >> 1);
        for (long long bi = (long long)blockIdx.x * kNv12Threads + t; bi < blocks; bi += (long long)gridDim.x * kNv12Threads) {
            const int by = (int)(bi / bx_n), bx = (int)(bi - (long long)by * bx_n);
            const long long o0 = (long long)(2 * by) * j.width + 2 * bx, ouv = (long long)by * j.width + 2 * bx;
            const uint32_t Y00 = yp[o0], Y01 = yp[o0 + 1], Y10 = yp[o0 + j.width], Y11 = yp[o0 + j.width + 1];
            int ruv, guv, buv;
            bt601_uv_terms(uvp[ouv], uvp[ouv + 1], ruv, guv, buv);
            uint32_t b0, g0, r0, b, g, r, U, V;
            const uint32_t a00 = px(Y00, ruv, guv, buv, b0, g0, r0);
            const uint32_t a01 = px(Y01, ruv, guv, buv, b, g, r);
            const uint32_t a10 = px(Y10, ruv, guv, buv, b, g, r);
            const uint32_t a11 = px(Y11, ruv, guv, buv, b, g, r);
            bt601_uv(b0, g0, r0, U, V);
            yo[o0] = (uint8_t)a00; yo[o0 + 1] = (uint8_t)a01; yo[o0 + j.width] = (uint8_t)a10; yo[o0 + j.width + 1] = (uint8_t)a11;
            uvo[ouv] = (uint8_t)U; uvo[ouv + 1] = (uint8_t)V;
        }
    }
}

// cv::cvtColor 4:2:0 codes as stand-alone conversions (same arithmetic as above, one 2x2 block per lane):
// MODE 0: COLOR_BGR2YUV_I420 (1frameMeasure.cpp:32 prepares its input with it): CV_8UC3 -> Y plane, U plane, V plane (tight)
// MODE 1: COLOR_YUV2BGR_NV12: tight NV12 -> CV_8UC3.        grid = (blocks, n_frames)
struct Cvt420Job {
    const uint8_t* src; uint8_t* dst;
    long long c3_step, c3_frame;               // the interleaved CV_8UC3 side (src for MODE 0, dst for MODE 1)
    long long planar_frame;                    // bytes between frames of the tight planar side
    int width, height;
};

// 16 BGR pixels packed into 48 bytes (three 16-byte stores; p 16-byte aligned)
__device__ __forceinline__ void store_bgr16(uint8_t* p, const uint32_t* b, const uint32_t* g, const uint32_t* r)
{
    uint32_t w[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = 0;
#pragma unroll
    for (int px = 0; px < 16; ++px) {
        w[(3 * px) >> 2] |= b[px] << (8 * ((3 * px) & 3));
        w[(3 * px + 1) >> 2] |= g[px] << (8 * ((3 * px + 1) & 3));
        w[(3 * px + 2) >> 2] |= r[px] << (8 * ((3 * px + 2) & 3));
    }
    u32x4* dp = reinterpret_cast<u32x4*>(p);
    const u32x4 r0 = {w[0], w[1], w[2], w[3]}, r1 = {w[4], w[5], w[6], w[7]}, r2 = {w[8], w[9], w[10], w[11]};
    dp[0] = r0; dp[1] = r1; dp[2] = r2;
}

// `vec` (host: width % 16 == 0, every base / pitch / frame stride a multiple of 16): a lane owns a 16 x 2 pixel group -- six 16-byte
// loads of BGR (or two of luma and one of chroma) and 16- / 8-byte stores -- instead of one 2 x 2 block with byte accesses
// (4K, 16 frames: 2.2 -> 5.5 TB/s of the 4.5 B/px for BGR2YUV_I420, 2.4 -> 4.1 for YUV2BGR_NV12).
template <int MODE>
__global__ __launch_bounds__(kThreads) void cvt420_kernel(Cvt420Job j, int vec)
{
    const int f = blockIdx.y, bx_n = j.width >> 1;
    const long long blocks = (long long)bx_n * (j.height >> 1), ysz = (long long)j.width * j.height;
    if (vec) {
        const int gx_n = j.width >> 4;
        const int groups = gx_n * (j.height >> 1);
        const int stride = (int)gridDim.x * kThreads, dby = stride / gx_n, dgx = stride - dby * gx_n;
        int gi = (int)blockIdx.x * kThreads + (int)threadIdx.x;
        int by = gi / gx_n, gx = gi - by * gx_n;
        for (; gi < groups; gi += stride, by += dby, gx += dgx) {
            if (gx >= gx_n) { gx -= gx_n; ++by; }
            if (MODE == 0) {
                const uint8_t* r0 = j.src + (long long)f * j.c3_frame + (long long)(2 * by) * j.c3_step + 48 * gx;
                uint8_t* pl = j.dst + (long long)f * j.planar_frame;
                uint32_t b0[16], g0[16], q0[16], b1[16], g1[16], q1[16];
                load_bgr16(r0, b0, g0, q0);
                load_bgr16(r0 + j.c3_step, b1, g1, q1);
                uint32_t y0[4] = {0, 0, 0, 0}, y1[4] = {0, 0, 0, 0}, uu[2] = {0, 0}, vv[2] = {0, 0};
#pragma unroll
                for (int px = 0; px < 16; ++px) {
                    y0[px >> 2] |= bt601_y(b0[px], g0[px], q0[px]) << (8 * (px & 3));
                    y1[px >> 2] |= bt601_y(b1[px], g1[px], q1[px]) << (8 * (px & 3));
                    if ((px & 1) == 0) {                           // chroma from the top-left pixel of each 2 x 2 block
                        uint32_t U, V;
                        bt601_uv(b0[px], g0[px], q0[px], U, V);
                        uu[px >> 3] |= U << (8 * ((px >> 1) & 3)); vv[px >> 3] |= V << (8 * ((px >> 1) & 3));
                    }
                }
                const u32x4 o0 = {y0[0], y0[1], y0[2], y0[3]}, o1 = {y1[0], y1[1], y1[2], y1[3]};
                *reinterpret_cast<u32x4*>(pl + (long long)(2 * by) * j.width + (gx << 4)) = o0;
                *reinterpret_cast<u32x4*>(pl + (long long)(2 * by + 1) * j.width + (gx << 4)) = o1;
                uint8_t* uo = pl + ysz + (long long)by * bx_n + (gx << 3);
                *reinterpret_cast<uint2*>(uo) = make_uint2(uu[0], uu[1]);
                *reinterpret_cast<uint2*>(uo + (ysz >> 2)) = make_uint2(vv[0], vv[1]);
            } else {
                const uint8_t* pl = j.src + (long long)f * j.planar_frame;
                const u32x4 y0 = *reinterpret_cast<const u32x4*>(pl + (long long)(2 * by) * j.width + (gx << 4));
                const u32x4 y1 = *reinterpret_cast<const u32x4*>(pl + (long long)(2 * by + 1) * j.width + (gx << 4));
                const u32x4 uv = *reinterpret_cast<const u32x4*>(pl + ysz + (long long)by * j.width + (gx << 4));
                uint32_t b0[16], g0[16], q0[16], b1[16], g1[16], q1[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int ruv, guv, buv;
                    bt601_uv_terms(byte_of(uv, 2 * k), byte_of(uv, 2 * k + 1), ruv, guv, buv);
                    bt601_px_bgr(byte_of(y0, 2 * k), ruv, guv, buv, b0[2 * k], g0[2 * k], q0[2 * k]);
                    bt601_px_bgr(byte_of(y0, 2 * k + 1), ruv, guv, buv, b0[2 * k + 1], g0[2 * k + 1], q0[2 * k + 1]);
                    bt601_px_bgr(byte_of(y1, 2 * k), ruv, guv, buv, b1[2 * k], g1[2 * k], q1[2 * k]);
                    bt601_px_bgr(byte_of(y1, 2 * k + 1), ruv, guv, buv, b1[2 * k + 1], g1[2 * k + 1], q1[2 * k + 1]);
                }
                uint8_t* d0 = j.dst + (long long)f * j.c3_frame + (long long)(2 * by) * j.c3_step + 48 * gx;
                store_bgr16(d0, b0, g0, q0);
                store_bgr16(d0 + j.c3_step, b1, g1, q1);
            }
        }
        return;
    }
    for (long long bi = (long long)blockIdx.x * kThreads + threadIdx.x; bi < blocks; bi += (long long)gridDim.x * kThreads) {
        const int by = (int)(bi / bx_n), bx = (int)(bi - (long long)by * bx_n);
        if (MODE == 0) {
            const uint8_t* r0 = j.src + (long long)f * j.c3_frame + (long long)(2 * by) * j.c3_step + 6 * bx;
            const uint8_t* r1 = r0 + j.c3_step;
            uint8_t* yo = j.dst + (long long)f * j.planar_frame + (long long)(2 * by) * j.width + 2 * bx;
            uint8_t* uo = j.dst + (long long)f * j.planar_frame + ysz + (long long)by * bx_n + bx;
            uint32_t U, V;
            bt601_uv(r0[0], r0[1], r0[2], U, V);
            yo[0] = (uint8_t)bt601_y(r0[0], r0[1], r0[2]); yo[1] = (uint8_t)bt601_y(r0[3], r0[4], r0[5]);
            yo[j.width] = (uint8_t)bt601_y(r1[0], r1[1], r1[2]); yo[j.width + 1] = (uint8_t)bt601_y(r1[3], r1[4], r1[5]);
            uo[0] = (uint8_t)U; uo[ysz >> 2] = (uint8_t)V;
        } else {
            const uint8_t* yp = j.src + (long long)f * j.planar_frame + (long long)(2 * by) * j.width + 2 * bx;
            const uint8_t* uv = j.src + (long long)f * j.planar_frame + ysz + (long long)by * j.width + 2 * bx;
            uint8_t* d0 = j.dst + (long long)f * j.c3_frame + (long long)(2 * by) * j.c3_step + 6 * bx;
            uint8_t* d1 = d0 + j.c3_step;
            int ruv, guv, buv;
            bt601_uv_terms(uv[0], uv[1], ruv, guv, buv);
            uint32_t b, g, r;
            bt601_px_bgr(yp[0], ruv, guv, buv, b, g, r); d0[0] = (uint8_t)b; d0[1] = (uint8_t)g; d0[2] = (uint8_t)r;
            bt601_px_bgr(yp[1], ruv, guv, buv, b, g, r); d0[3] = (uint8_t)b; d0[4] = (uint8_t)g; d0[5] = (uint8_t)r;
            bt601_px_bgr(yp[j.width], ruv, guv, buv, b, g, r); d1[0] = (uint8_t)b; d1[1] = (uint8_t)g; d1[2] = (uint8_t)r;
            bt601_px_bgr(yp[j.width + 1], ruv, guv, buv, b, g, r); d1[3] = (uint8_t)b; d1[4] = (uint8_t)g; d1[5] = (uint8_t)r;
        }
    }
}

}  // namespace mi
