// color_clahe.hip.h -- CLAHE on the luma of interleaved BGR images without planar intermediates
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "clahe.hip.h"
#include "color.hip.h"

namespace mi {
// =============================================================================================
// clahe1frame.cpp:83-102 -- cvtColor(BGR2YUV) -> split -> CLAHE::apply on Y -> merge -> cvtColor(YUV2BGR) -- in two passes
// over the interleaved image (9 B/px) instead of through Y/U/V/Y' planes (14 B/px): pass 1 converts on the fly and builds
// the per-tile Y histograms, pass 2 converts, blends Y through the tile LUTs and converts back.  Same arithmetic per
// pixel as the planar sequence (U and V are the saturated bytes the reference stores in between).
// Taken only for the common shape: no REFLECT_101 padding (W % tiles_x == 0, H % tiles_y == 0), tile_w % 16 == 0,
// 16-byte aligned rows, tiles_x <= 14 (f32 pair tables); everything else keeps the planar path.
// =============================================================================================

// 16 interleaved pixels (48 B, 16-B aligned) -> 16 luma bytes packed in a u32x4 (and, optionally, U and V packed likewise)
template <bool WANT_UV>
__device__ __forceinline__ u32x4 bgr16_to_y(const uint8_t* p, u32x4* up, u32x4* vp)
{
    uint32_t c0[16], c1[16], c2[16];
    load_bgr16(p, c0, c1, c2);
    uint32_t wy[4] = {0, 0, 0, 0}, wu[4] = {0, 0, 0, 0}, wv[4] = {0, 0, 0, 0};
#pragma unroll
    for (int px = 0; px < 16; ++px) {
        if (WANT_UV) {
            uint32_t Y, U, V;
            px_bgr2yuv(c0[px], c1[px], c2[px], Y, U, V);
            wy[px >> 2] |= Y << (8 * (px & 3));
            wu[px >> 2] |= U << (8 * (px & 3)); wv[px >> 2] |= V << (8 * (px & 3));
        } else {
            wy[px >> 2] |= px_bgr2y(c0[px], c1[px], c2[px]) << (8 * (px & 3));      // luma alone: the opaque descales of U and V would not be dropped
        }
    }
    if (WANT_UV) {
        u32x4 u = {wu[0], wu[1], wu[2], wu[3]}, v = {wv[0], wv[1], wv[2], wv[3]};
        *up = u; *vp = v;
    }
    u32x4 y = {wy[0], wy[1], wy[2], wy[3]};
    return y;
}

// pass 1: per-tile luma histogram partials straight from BGR.  grid = (S, tiles, n_frames), as tile_hist_kernel: NT = 512 threads
// share the tile's histogram (32 waves per CU instead of 16), (row, slot) items are walked incrementally (no division per item),
// two 48-byte items in flight per lane; waves 4..7 leave before the 256-thread fold / LUT stage.
template <int NT>
__global__ __launch_bounds__(NT) void bgr_tile_hist_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                          ClaheGeom g, uint32_t* __restrict__ partial, uint8_t* __restrict__ luts)
{
    __shared__ uint32_t h[256 * kCopies];
    __shared__ uint32_t s_wave[4];
    const int t = threadIdx.x;
    for (int i = t; i < 256 * kCopies; i += NT) h[i] = 0;
    __syncthreads();
    const uint32_t copy = t & (kCopies - 1);
    const int S = gridDim.x, s = blockIdx.x, tile = blockIdx.y, f = blockIdx.z;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const int r0 = (int)((long long)g.tile_h * s / S), r1 = (int)((long long)g.tile_h * (s + 1) / S);
    const uint8_t* src = src_base + (long long)f * frame_stride + (long long)tx * g.tile_w * 3 + (long long)(ty * g.tile_h + r0) * step;
    const int slots = g.tile_w >> 4;                               // 16-pixel groups per tile row (tile_w % 16 == 0)
    const int items = (r1 - r0) * slots;                           // < 2^27: the tile area is < 2^31
    int row = t / slots, slot = t - row * slots;
    const int drow = NT / slots, dslot = NT - drow * slots;
    auto item_ptr = [&]() -> const uint8_t* {
        const uint8_t* p = src + (long long)row * step + slot * 48;
        row += drow; slot += dslot;
        if (slot >= slots) { slot -= slots; ++row; }
        return p;
    };
    for (int it = t; it < items; it += 2 * NT) {
        const uint8_t* p0 = item_ptr();
        const uint8_t* p1 = item_ptr();
        const bool two = it + NT < items;
        const u32x4 y0 = bgr16_to_y<false>(p0, nullptr, nullptr);
        if (two) {
            const u32x4 y1 = bgr16_to_y<false>(p1, nullptr, nullptr);
            hist_add_vec(h, y0, copy);
            hist_add_vec(h, y1, copy);
        } else {
            hist_add_vec(h, y0, copy);
        }
    }
    __syncthreads();
    if (NT > kThreads && t >= kThreads) return;
    const uint32_t bin = lds_hist_bin(h, t);
    if (luts) luts[((size_t)f * gridDim.y + tile) * 256 + t] = tile_lut_value(bin, g, s_wave);     // one workgroup per tile: LUT in place
    else partial[(((size_t)f * gridDim.y + tile) * S + s) * 256 + t] = bin;
}

// pass 2: convert, blend Y through the four neighbouring tile LUTs (f32 pair tables in LDS, as clahe_interp_kernel<true>),
// convert back.  grid = (bands*subs, n_frames, col_segments); src == dst allowed (a lane reads its 16 pixels before
// writing them).
__global__ __launch_bounds__(kThreads) void bgr_clahe_interp_kernel(ColorJob j, ClaheGeom g, const uint8_t* __restrict__ luts, int subs, int groups)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t quad[];
    f32x4* quadf = reinterpret_cast<f32x4*>(quad);
    const int t = threadIdx.x, f = (int)gridDim.y - 1 - (int)blockIdx.y;      // last-to-first, see clahe_interp_kernel
    const int band = blockIdx.x / subs, sub = blockIdx.x - band * subs;
    const int ty1u = band - 1;
    const int ty1 = max(ty1u, 0), ty2 = min(ty1u + 1, g.tiles_y - 1);
    const uint8_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * 256;
    const uint8_t* l1 = lf + (size_t)ty1 * g.tiles_x * 256;
    const uint8_t* l2 = lf + (size_t)ty2 * g.tiles_x * 256;
    const int npairs = g.tiles_x + 1;
    for (int i = t; i < npairs * 256; i += kThreads) {
        const int pr = i >> 8, v = i & 255;
        const int ta = max(pr - 1, 0), tb = min(pr, g.tiles_x - 1);
        const f32x4 e = {(float)l1[ta * 256 + v], (float)l2[ta * 256 + v], (float)l1[tb * 256 + v], (float)l2[tb * 256 + v]};   // {a, c, b, d}
        quadf[i] = e;
    }
    __syncthreads();
    const int y_lo_band = (int)max(0LL, ((long long)(2 * band - 1) * g.tile_h) / 2 - kBandMargin);
    const int y_hi_band = (int)min((long long)g.height, ((long long)(2 * band + 1) * g.tile_h + 1) / 2 + kBandMargin);
    const int nrows = max(0, y_hi_band - y_lo_band);
    const int y_lo = y_lo_band + (int)((long long)nrows * sub / subs);
    const int y_hi = y_lo_band + (int)((long long)nrows * (sub + 1) / subs);
    const int phases = kThreads / groups;
    const int grp = t % groups, phase = t / groups;
    const int x0 = (blockIdx.z * groups + grp) * kInterpPx;
    if (phase >= phases || x0 >= g.width) return;
    f32x2 xw[kInterpPx];
    int poff[kInterpPx];
#pragma unroll
    for (int k = 0; k < kInterpPx; ++k) {
        const float txf = __fsub_rn(__fmul_rn((float)(x0 + k), g.inv_tw), 0.5f);
        const int tx1 = floor_f32_to_int(txf);
        const float xa = __fsub_rn(txf, (float)tx1);
        xw[k].x = __fsub_rn(1.0f, xa); xw[k].y = xa;
        int pr = tx1 + 1;
        pr = pr < 0 ? 0 : (pr > g.tiles_x ? g.tiles_x : pr);
        poff[k] = pr << 8;
    }
    const uint8_t* src = j.src + (long long)f * j.src_frame + (long long)x0 * 3;
    uint8_t* dst = j.dst + (long long)f * j.dst_frame + (long long)x0 * 3;
    auto ty1_of = [&](int y) { return floor_f32_to_int(__fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f)); };
    int ya_lo = y_lo, ya_hi = y_hi;
    while (ya_lo < ya_hi && ty1_of(ya_lo) != ty1u) ++ya_lo;
    while (ya_hi > ya_lo && ty1_of(ya_hi - 1) != ty1u) --ya_hi;
    for (int y = ya_lo + ((phase - (ya_lo - y_lo) % phases) % phases + phases) % phases; y < ya_hi; y += phases) {
        u32x4 u, v;
        const u32x4 yq = bgr16_to_y<true>(src + (long long)y * j.src_step, &u, &v);
        const float tyf = __fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f);
        const float ya = __fsub_rn(tyf, (float)ty1u), ya1 = __fsub_rn(1.0f, ya);
        const u32x4 yo = clahe_vec16_f32<false>(quadf, yq, poff, xw, ya, ya1);       // the host takes the planar path for ClaheGeom::contract
        uint32_t w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            uint32_t b, gg, r;
            px_yuv2bgr(byte_of(yo, px), byte_of(u, px), byte_of(v, px), b, gg, r);
            w[(3 * px) >> 2] |= b << (8 * ((3 * px) & 3));
            w[(3 * px + 1) >> 2] |= gg << (8 * ((3 * px + 1) & 3));
            w[(3 * px + 2) >> 2] |= r << (8 * ((3 * px + 2) & 3));
        }
        u32x4* dp = reinterpret_cast<u32x4*>(dst + (long long)y * j.dst_step);
        const u32x4 o0 = {w[0], w[1], w[2], w[3]}, o1 = {w[4], w[5], w[6], w[7]}, o2 = {w[8], w[9], w[10], w[11]};
        dp[0] = o0; dp[1] = o1; dp[2] = o2;
    }
}

}  // namespace mi
