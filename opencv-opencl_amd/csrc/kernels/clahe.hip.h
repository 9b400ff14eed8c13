// clahe.hip.h -- K4 tile histograms, K5 clip/redistribute/LUT, K6 bilinear LUT interpolation (8-bit CLAHE)
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "common.hip.h"
#include "equalize.hip.h"

namespace mi {
// =============================================================================================
// CLAHE  (oracle: orc_clahe_tile_luts / orc_clahe_interpolate)
// =============================================================================================
struct ClaheGeom {
    int width, height;          // unpadded image
    int tiles_x, tiles_y;
    int tile_w, tile_h;         // tile size on the REFLECT_101-extended image
    int clip;                   // integer clip limit (0 = off)
    float lut_scale;            // 255.f / (tile_w*tile_h), computed on the host (IEEE division)
    float inv_tw, inv_th;       // 1.f/tile_w, 1.f/tile_h, computed on the host
    int contract;               // 0: separately rounded mul/add (x86-64 baseline OpenCV).  1: the FMAs GCC forms from clahe.cpp's
                                // expressions on FMA targets (distribution OpenCV on aarch64, the reference's board); see
                                // oracle/lumaeq_oracle.c orc_set_fp_contract
};

// core/src/copy.cpp borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// K5's arithmetic for bin t = threadIdx.x given this bin's count c (all 256 threads call it): clip, redistribute, CDF -> LUT.
// clahe.cpp CLAHE_CalcLut_Body: the sequential residual loop
//     for (i = 0; i < 256 && residual > 0; i += step, --residual) ++h[i];
// increments bin b iff b % step == 0 and b / step < residual.
__device__ __forceinline__ uint8_t tile_lut_value(uint32_t c, const ClaheGeom& g, uint32_t* s_wave /*[4]*/)
{
    const int t = threadIdx.x;
    int hv = (int)c;
    if (g.clip > 0) {
        const uint32_t excess = hv > g.clip ? (uint32_t)(hv - g.clip) : 0u;
        uint32_t clipped;
        block_incl_scan(excess, s_wave, &clipped);
        if (hv > g.clip) hv = g.clip;
        const int batch = (int)clipped / 256;
        int residual = (int)clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            int rstep = 256 / residual; if (rstep < 1) rstep = 1;
            if (t % rstep == 0 && t / rstep < residual) ++hv;
        }
    }
    const uint32_t sum = block_incl_scan((uint32_t)hv, s_wave, nullptr);
    int r = __float2int_rn(__fmul_rn((float)(int)sum, g.lut_scale));
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    return (uint8_t)r;
}

// ---------------------------------------------------------------------------------------------
// K4  per-tile histogram partials.  grid = (S, tiles, n_frames); partial[f][tile][s][256].
// The padded image is never materialised: rows/columns beyond the frame are read by index
// reflection.  Work items are (row, 16-byte slot) pairs walked incrementally so short tile rows
// (480 B at 4K 8x8) still give every lane a vector load.
// ---------------------------------------------------------------------------------------------
// With one workgroup per tile (gridDim.x == 1, the batch case) the finished histogram never leaves the CU: the LUT is
// computed in place and written to `luts` (K5 folded in, no partials round trip, one launch fewer).
// XCD-aware order (xcd_map, speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2, and with the
// plain order the horizontally adjacent tiles of an 8-wide grid land on 8 different XCDs -- every 128-byte line cut by a tile edge
// (tile rows are 480 bytes at 4K 8x8) is then fetched from HBM twice.  With the map, dispatch slot i of a frame works on tile
// (i % 8) * (tiles / 8) + i / 8: each XCD owns a contiguous row-major run of tiles and walks it in order, so both halves of a cut
// line are requested through the same L2 within microseconds of each other.
// columns of a tile that are not covered by whole 16-byte slots: the ragged right edge of the in-frame part and the reflected columns
// of a right-border tile (byte loads; rows r0..r1 of the tile)
template <int NT>
__device__ __forceinline__ void tile_hist_edges(uint32_t* h, const uint8_t* src, long long step, const ClaheGeom& g, int tx, int ty,
                                                int r0, int r1, uint32_t copy)
{
    const int t = threadIdx.x;
    const int x0 = tx * g.tile_w;
    const int in_w = max(0, min(g.tile_w, g.width - x0));
    const int slots = in_w >> 4;
    if ((in_w & 15) != 0) {                                     // ragged right edge of the in-frame part: byte loads
        const int pw = in_w & 15, xs = x0 + (slots << 4);
        const long long items = (long long)(r1 - r0) * pw;
        for (long long it = t; it < items; it += NT) {
            const int row = (int)(it / pw), c = (int)(it - (long long)row * pw);
            const int y = reflect101(ty * g.tile_h + r0 + row, g.height);
            lds_inc(h, ((uint32_t)src[(long long)y * step + xs + c] << kCopyShift) + copy);
        }
    }
    if (in_w < g.tile_w) {                                      // reflected columns (right border tiles only)
        const int pw = g.tile_w - in_w;
        const long long items = (long long)(r1 - r0) * pw;
        for (long long it = t; it < items; it += NT) {
            const int row = (int)(it / pw), c = (int)(it - (long long)row * pw);
            const int y = reflect101(ty * g.tile_h + r0 + row, g.height);
            const int x = reflect101(x0 + in_w + c, g.width);
            lds_inc(h, ((uint32_t)src[(long long)y * step + x] << kCopyShift) + copy);
        }
    }
}

// NT = 256 or 512 threads: the histogram is shared by the whole workgroup either way (32 KiB), so 512 threads put 32 waves on a CU
// (4 workgroups) instead of 20 (5 workgroups of 4 waves); waves 4..7 leave before the 256-thread fold / LUT stage.
template <int NT>
__global__ __launch_bounds__(NT) void tile_hist_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                      ClaheGeom g, uint32_t* __restrict__ partial, uint8_t* __restrict__ luts, int xcd_map)
{
    __shared__ uint32_t h[256 * kCopies];                       // exactly 32 KiB: five workgroups per CU (a 16-byte scan scratch
    uint32_t* const s_wave = h;                                 // next to it made it four); the scans reuse h[0..3] once h is folded
    const int t = threadIdx.x;
    for (int i = t; i < 256 * kCopies; i += NT) h[i] = 0;
    __syncthreads();
    const uint32_t copy = t & (kCopies - 1);
    const int S = gridDim.x, s = blockIdx.x, f = blockIdx.z;
    const int ntiles = gridDim.y;
    const int tile = xcd_map ? ((int)(blockIdx.y & 7) * (ntiles >> 3) + (int)(blockIdx.y >> 3)) : (int)blockIdx.y;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    const int r0 = (int)((long long)g.tile_h * s / S), r1 = (int)((long long)g.tile_h * (s + 1) / S);
    const int x0 = tx * g.tile_w;
    const int in_w = max(0, min(g.tile_w, g.width - x0));     // columns of this tile that lie inside the frame
    const int slots = in_w >> 4;                               // full 16-byte slots per row
    if (slots > 0) {
        const int rows = r1 - r0;
        const long long items = (long long)rows * slots;
        int row = t / slots, slot = t - row * slots;
        const int drow = NT / slots, dslot = NT - drow * slots;
        auto item_ptr = [&]() -> const u32x4_u* {             // address of the current (row, slot), then advance by 256 items
            const int y = reflect101(ty * g.tile_h + r0 + row, g.height);
            const u32x4_u* p = reinterpret_cast<const u32x4_u*>(src + (long long)y * step + x0 + (slot << 4));
            row += drow; slot += dslot;
            if (slot >= slots) { slot -= slots; ++row; }
            return p;
        };
        // groups of four 16-byte loads per lane, each predicated on its own bound: the ragged end of a tile (and all of a small tile:
        // a 240 x 135 tile is four vectors per lane) still has four loads in flight instead of a serial tail -- tile histograms of
        // 1080p 8x8 / 4K 16x16: -12 % / -16 %, 4K 8x8 unchanged (profiles/r02_n_clahe_ab_tail.txt)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (long long it = t; it < items; it += 4 * NT) {
            u32x4 cur[4]; bool cv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { cv[k] = it + (long long)k * NT < items; const u32x4_u* q = item_ptr(); cur[k] = cv[k] ? *q : zero; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (cv[k]) hist_add_vec(h, cur[k], copy);
        }
    }
    tile_hist_edges<NT>(h, src, step, g, tx, ty, r0, r1, copy);
    __syncthreads();
    if (NT > kThreads && t >= kThreads) return;                 // the fold and the LUT are 256-thread stages (terminated waves leave the barriers)
    const uint32_t bin = lds_hist_bin(h, t);
    __syncthreads();                                            // everybody has folded its bin: h[0..3] becomes the scan scratch
    if (luts) luts[((size_t)f * gridDim.y + tile) * 256 + t] = tile_lut_value(bin, g, s_wave);     // host passes luts only when S == 1
    else partial[(((size_t)f * gridDim.y + tile) * S + s) * 256 + t] = bin;
}

// ---------------------------------------------------------------------------------------------
// K4m  K4 for batches of SMALL tiles: one workgroup walks K consecutive tiles of its XCD's run, LUT included.
// Workgroups are dispatched at ~4 ns apiece (measured with the pixel loop switched off: 17 us per 4096, 61 us per 16384,
// profiles/r02_n_clahe_ab_no_pixel_loop.txt), so a 720p 8x8 batch of 576 frames -- the same bytes as 64 4K frames, in 36 864 tiles
// of 160 x 90 pixels -- spends 150 of its 188 us being dispatched.  K tiles per workgroup divide that by K; the price is that fold,
// scans and LUT of a tile now sit between two pixel loops of the same workgroup (the other three workgroups of the CU cover it).
// grid = (1, tiles / K, frames), 512 threads.  Waves 4..7 zero the histogram for the next tile while waves 0..3 scan; the two sides
// of that branch execute different s_barrier instructions, hence the scalar condition.
// Tile order: dispatch slot g of a frame (XCD g % 8) -> tiles (g % 8) * (tiles / 8) + (g / 8) * K + k, k = 0..K-1, which needs
// (tiles / 8) % K == 0; without the XCD map: g * K + k.
// ---------------------------------------------------------------------------------------------
constexpr int kTileMultiThreads = 512;
__global__ __launch_bounds__(kTileMultiThreads) void tile_hist_multi_kernel(const uint8_t* __restrict__ src_base, long long step,
                                                                            long long frame_stride, ClaheGeom g,
                                                                            uint8_t* __restrict__ luts, int ntiles, int K, int xcd_map)
{
    constexpr int NT = kTileMultiThreads;
    __shared__ uint32_t h[256 * kCopies];
    __shared__ uint32_t s_wave[4];
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const bool lower = __builtin_amdgcn_readfirstlane(t >> 6) < kThreads / 64;      // waves 0..3: the 256 threads of the fold / LUT stage
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int i = t; i < 256 * kCopies / 4; i += NT) reinterpret_cast<u32x4*>(h)[i] = zero;
    const int f = blockIdx.z, grp = blockIdx.y;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    const int first = xcd_map ? ((grp & 7) * (ntiles >> 3) + (grp >> 3) * K) : grp * K;
    for (int k = 0; k < K; ++k) {
        __syncthreads();                                        // histogram zeroed (and the previous tile's scans done with s_wave)
        const int tile = first + k;
        const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
        const int x0 = tx * g.tile_w;
        const int in_w = max(0, min(g.tile_w, g.width - x0));
        const int slots = in_w >> 4;
        if (slots > 0) {
            const int items = g.tile_h * slots;
            int row = t / slots, slot = t - row * slots;
            const int drow = NT / slots, dslot = NT - drow * slots;
            auto item_ptr = [&]() -> const u32x4_u* {
                const int y = reflect101(ty * g.tile_h + row, g.height);
                const u32x4_u* p = reinterpret_cast<const u32x4_u*>(src + (long long)y * step + x0 + (slot << 4));
                row += drow; slot += dslot;
                if (slot >= slots) { slot -= slots; ++row; }
                return p;
            };
            for (int it = t; it < items; it += 4 * NT) {
                u32x4 cur[4]; bool cv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { cv[q] = it + q * NT < items; const u32x4_u* p = item_ptr(); cur[q] = cv[q] ? *p : zero; }
#pragma unroll
                for (int q = 0; q < 4; ++q) if (cv[q]) hist_add_vec(h, cur[q], copy);
            }
        }
        tile_hist_edges<NT>(h, src, step, g, tx, ty, 0, g.tile_h, copy);
        __syncthreads();                                        // A: every pixel of the tile counted
        uint32_t bin = 0;
        if (lower) {
#pragma unroll 1
            for (int k0 = 0; k0 < kCopies; k0 += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) bin += h[(t << kCopyShift) + ((k0 + q + t) & (kCopies - 1))];
            }
        }
        __syncthreads();                                        // B: folded
        if (lower) {
            luts[((size_t)f * ntiles + tile) * 256 + t] = tile_lut_value(bin, g, s_wave);      // two barriers per scan inside
        } else {
            if (k + 1 < K)
                for (int q = t - kThreads; q < 256 * kCopies / 4; q += NT - kThreads) reinterpret_cast<u32x4*>(h)[q] = zero;
            if (g.clip > 0) { __syncthreads(); __syncthreads(); }
            __syncthreads(); __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5  per-tile clip + redistribute + CDF -> uchar LUT.  grid = (tiles, n_frames), 256 threads = bins.
// clahe.cpp CLAHE_CalcLut_Body: the sequential residual loop
//     for (i = 0; i < 256 && residual > 0; i += step, --residual) ++h[i];
// increments bin b iff b % step == 0 and b / step < residual.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void tile_lut_kernel(const uint32_t* __restrict__ partial, int S, ClaheGeom g,
                                                           uint8_t* __restrict__ luts)
{
    __shared__ uint32_t s_wave[4];
    const int t = threadIdx.x;
    const size_t tile_id = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t* pp = partial + tile_id * S * 256 + t;
    uint32_t c = 0;
    for (int s = 0; s < S; ++s) c += pp[(size_t)s * 256];
    luts[tile_id * 256 + t] = tile_lut_value(c, g, s_wave);
}

// ---------------------------------------------------------------------------------------------
// K6  bilinear interpolation of the four neighbouring tile LUTs (clahe.cpp CLAHE_Interpolation_Body).
// grid = (bands*subs, n_frames, col_segments).  A "band" is the set of rows with the same unclamped
// ty1 (= band-1), so the two LUT rows a workgroup needs are fixed; it stages, for every
// horizontal tile pair p (unclamped tx1 = p-1), quad[p][v] = {LUT[ty1][tx1][v], LUT[ty1][tx2][v],
// LUT[ty2][tx1][v], LUT[ty2][tx2][v]} as one dword in LDS, so a pixel costs ONE ds_read_b32.
// A lane owns 16 fixed columns (their xa/xa1/pair are lane constants) and walks down the rows.
// Float ops: nine individually rounded f32 ops per pixel, no FMA (App. A.2 step 5).
// ---------------------------------------------------------------------------------------------
constexpr int kInterpPx = 16;           // pixels per lane per row
constexpr int kMaxPairsLdsF32 = 15;     // float tables: (tiles_x + 1) * 4 KiB of LDS (<= 60 KiB)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxPairsLds = 63;        // (tiles_x + 1) KiB of LDS (<= 64 KiB dynamic); wider grids use the global-LUT kernel
constexpr int kBandMargin = 4;          // rows; covers the f32 rounding of y*inv_th - 0.5 for any height <= 2^24

__device__ __forceinline__ int floor_f32_to_int(float v) { const int i = (int)v; return i - ((float)i > v); }   // cvFloor

// p * inv - 0.5f in the two arithmetic modes (ClaheGeom::contract)
template <bool FMA>
__device__ __forceinline__ float tile_coord(int p, float inv) { return FMA ? __fmaf_rn((float)p, inv, -0.5f) : __fsub_rn(__fmul_rn((float)p, inv), 0.5f); }
__device__ __forceinline__ float tile_coord(int p, float inv, int contract) { return contract ? tile_coord<true>(p, inv) : tile_coord<false>(p, inv); }

// res = (a*xa1 + b*xa)*ya1 + (c*xa1 + d*xa)*ya, nine individually rounded f32 ops, then round half to even.
// contracted form (GCC, FMA target): fma(fma(a, xa1, b*xa), ya1, fma(c, xa1, d*xa) * ya)
template <bool FMA = false>
__device__ __forceinline__ float clahe_blend_f(float a, float b, float c, float d, float xa, float xa1, float ya, float ya1)
{
    if (FMA) return __fmaf_rn(__fmaf_rn(a, xa1, __fmul_rn(b, xa)), ya1, __fmul_rn(__fmaf_rn(c, xa1, __fmul_rn(d, xa)), ya));
    const float top = __fmul_rn(__fadd_rn(__fmul_rn(a, xa1), __fmul_rn(b, xa)), ya1);
    const float bot = __fmul_rn(__fadd_rn(__fmul_rn(c, xa1), __fmul_rn(d, xa)), ya);
    return __fadd_rn(top, bot);
}
template <bool FMA = false>
__device__ __forceinline__ float clahe_blend(uint32_t q, float xa, float xa1, float ya, float ya1)
{
    const float a = (float)(q & 0xffu), b = (float)((q >> 8) & 0xffu), c = (float)((q >> 16) & 0xffu), d = (float)(q >> 24);
    return rintf(clahe_blend_f<FMA>(a, b, c, d, xa, xa1, ya, ya1));  // v_rndne_f32: cvRound
}
template <bool FMA = false>
__device__ __forceinline__ uint32_t clahe_px(uint32_t q, float xa, float xa1, float ya, float ya1)
{
    int r = (int)clahe_blend<FMA>(q, xa, xa1, ya, ya1);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);                             // saturate_cast<uchar>
    return (uint32_t)r;
}
// 16 pixels of one row: one ds_read_b32 per pixel, v_cvt_pk_u8_f32 (saturating, input already integral) packs the bytes
template <bool FMA = false>
__device__ __forceinline__ u32x4 clahe_vec16(const uint32_t* quad, u32x4 q, const int* poff, const float* xa, const float* xa1, float ya, float ya1)
{
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = k * 4 + b;
            const uint32_t v = (w[k] >> (8 * b)) & 0xffu;
            acc = __builtin_amdgcn_cvt_pk_u8_f32(clahe_blend<FMA>(quad[poff[j] + v], xa[j], xa1[j], ya, ya1), b, acc);
        }
        ow[k] = acc;
    }
    u32x4 o; o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
    return o;
}

// Float-table variant of the 16-pixel body: the LDS entry is {a, c, b, d} as f32, so one ds_read_b128 delivers
// two register pairs that feed v_pk_mul_f32 / v_pk_add_f32 directly (each lane of a packed op is an ordinary
// individually rounded f32 op): 4 packed ops + 1 add per pixel, no byte->float converts.
// The column weights live as ONE register pair {xa1, xa} per pixel; op_sel broadcasts one half of it to both lanes of
// the packed multiply (the compiler would otherwise keep {xa1, xa1} and {xa, xa}: 32 more VGPRs per lane).
__device__ __forceinline__ f32x2 pk_mul_bcast_lo(f32x2 a, f32x2 x)      // {a.x * x.x, a.y * x.x}
{
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(x));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul_bcast_hi(f32x2 a, f32x2 x)      // {a.x * x.y, a.y * x.y}
{
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(x));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_bcast_lo(f32x2 a, f32x2 x, f32x2 c)      // {a.x * x.x + c.x, a.y * x.x + c.y}, one rounding each
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(x), "v"(c));
    return d;
}
template <bool FMA = false>
__device__ __forceinline__ u32x4 clahe_vec16_f32(const f32x4* quadf, u32x4 q, const int* poff, const f32x2* xw, float ya, float ya1)
{
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    const f32x2 yv = {ya1, ya};
    uint32_t ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // the four LDS reads of a dword first (16 VGPRs in flight), then four independent blend chains: keeps
        // the packed ops of different pixels interleaved instead of one LDS round trip + dependent chain per pixel
        f32x4 e[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) e[b] = quadf[poff[k * 4 + b] + ((w[k] >> (8 * b)) & 0xffu)];
        f32x2 tb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = k * 4 + b;
            const f32x2 ac = {e[b].x, e[b].y}, bd = {e[b].z, e[b].w};
            if (FMA) tb[b] = pk_fma_bcast_lo(ac, xw[j], pk_mul_bcast_hi(bd, xw[j]));     // {fma(a,xa1,b*xa), fma(c,xa1,d*xa)}
            else tb[b] = (pk_mul_bcast_lo(ac, xw[j]) + pk_mul_bcast_hi(bd, xw[j])) * yv;  // pk_mul, pk_mul, pk_add, pk_mul
        }
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float r = FMA ? __fmaf_rn(tb[b].x, ya1, __fmul_rn(tb[b].y, ya)) : __fadd_rn(tb[b].x, tb[b].y);
            acc = __builtin_amdgcn_cvt_pk_u8_f32(rintf(r), b, acc);
        }
        ow[k] = acc;
    }
    u32x4 o; o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
    return o;
}

// pair_cap: pairs the LDS table holds.  With tiles_x + 1 <= pair_cap the table covers every pair of the frame; otherwise each column
// segment (blockIdx.z, `groups` 16-pixel groups wide -- the host sizes it so that a segment touches at most pair_cap pairs) stages only
// the pairs ITS columns use, first pair = p0 below, and the float tables serve grids of up to 63 tiles across (16 x 16 on 4K: the
// interpolation 322 -> see DESIGN.md).
template <bool FT, bool FMA>
__global__ __launch_bounds__(kThreads) void clahe_interp_kernel(PlaneBatch p, ClaheGeom g, const uint8_t* __restrict__ luts,
                                                               int subs, int groups, UVJob uv, int pair_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t quad[];   // [(tiles_x + 1)][256] u32 quads, or f32x4 when FT
    f32x4* quadf = reinterpret_cast<f32x4*>(quad);
    // frames are walked last-to-first: the histogram pass has just streamed the batch first-to-last, so its tail is what the
    // memory-side Infinity Cache still holds
    const int t = threadIdx.x, f = (int)gridDim.y - 1 - (int)blockIdx.y;
    const int band = blockIdx.x / subs, sub = blockIdx.x - band * subs;
    const int ty1u = band - 1;                                // unclamped ty1 of every row of the band
    const int ty1 = max(ty1u, 0), ty2 = min(ty1u + 1, g.tiles_y - 1);
    const uint8_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * 256;
    const uint8_t* l1 = lf + (size_t)ty1 * g.tiles_x * 256;
    const uint8_t* l2 = lf + (size_t)ty2 * g.tiles_x * 256;
    int p0 = 0, npairs = g.tiles_x + 1;
    if (npairs > pair_cap) {                                  // the pairs of this column segment only
        const int xs = (int)blockIdx.z * groups * kInterpPx;
        const int xe = min(g.width, xs + groups * kInterpPx) - 1;
        auto pair_of = [&](int x) { const int q = floor_f32_to_int(tile_coord<FMA>(x, g.inv_tw)) + 1; return q < 0 ? 0 : (q > g.tiles_x ? g.tiles_x : q); };
        p0 = pair_of(xs);
        npairs = min(pair_of(max(xe, xs)) - p0 + 1, pair_cap);
    }
    for (int i = t; i < npairs * 256; i += kThreads) {
        const int pr = p0 + (i >> 8), v = i & 255;
        const int ta = max(pr - 1, 0), tb = min(pr, g.tiles_x - 1);
        if (FT) {
            const f32x4 e = {(float)l1[ta * 256 + v], (float)l2[ta * 256 + v], (float)l1[tb * 256 + v], (float)l2[tb * 256 + v]};   // {a, c, b, d}
            quadf[i] = e;
        } else {
            quad[i] = (uint32_t)l1[ta * 256 + v] | ((uint32_t)l1[tb * 256 + v] << 8) |
                      ((uint32_t)l2[ta * 256 + v] << 16) | ((uint32_t)l2[tb * 256 + v] << 24);
        }
    }
    __syncthreads();

    // rows of this band: ideal range [(band-0.5)*th, (band+0.5)*th), widened by kBandMargin rows each side and
    // filtered by the float-computed ty1 so the decision is exactly the reference's.
    const int y_lo_band = (int)max(0LL, ((long long)(2 * band - 1) * g.tile_h) / 2 - kBandMargin);
    const int y_hi_band = (int)min((long long)g.height, ((long long)(2 * band + 1) * g.tile_h + 1) / 2 + kBandMargin);
    const int nrows = max(0, y_hi_band - y_lo_band);
    const int y_lo = y_lo_band + (int)((long long)nrows * sub / subs);
    const int y_hi = y_lo_band + (int)((long long)nrows * (sub + 1) / subs);

    const int phases = kThreads / groups;
    const int grp = t % groups, phase = t / groups;
    const int x0 = (blockIdx.z * groups + grp) * kInterpPx;
    if (phase < phases && x0 < g.width) {
        float xa[kInterpPx], xa1[kInterpPx];
        f32x2 xw[kInterpPx];                                   // {xa1, xa} pairs for the packed float-table body
        int poff[kInterpPx];
#pragma unroll
        for (int j = 0; j < kInterpPx; ++j) {
            const float txf = tile_coord<FMA>(x0 + j, g.inv_tw);
            const int tx1 = floor_f32_to_int(txf);
            xa[j] = __fsub_rn(txf, (float)tx1);
            xa1[j] = __fsub_rn(1.0f, xa[j]);
            xw[j].x = xa1[j]; xw[j].y = xa[j];
            int pr = tx1 + 1;                                  // pair index; columns beyond the frame are never used
            pr = pr < 0 ? 0 : (pr > g.tiles_x ? g.tiles_x : pr);
            pr -= p0;                                          // position in this workgroup's table
            pr = pr < 0 ? 0 : (pr >= npairs ? npairs - 1 : pr);
            poff[j] = pr << 8;
        }
        const uint8_t* src = p.src + (long long)f * p.src_frame;
        uint8_t* dst = p.dst + (long long)f * p.dst_frame;
        const bool full = x0 + kInterpPx <= g.width;
        // ty1 is monotone in y: trim the widened range to the rows that really belong to this band, using the
        // reference's own float expression (at most kBandMargin+1 steps per end)
        auto ty1_of = [&](int y) { return floor_f32_to_int(tile_coord<FMA>(y, g.inv_th)); };
        int ya_lo = y_lo, ya_hi = y_hi;
        while (ya_lo < ya_hi && ty1_of(ya_lo) != ty1u) ++ya_lo;
        while (ya_hi > ya_lo && ty1_of(ya_hi - 1) != ty1u) --ya_hi;
        // rows of this lane: ya_lo + phase, + phases, ...  (sub-ranges are contiguous per block, phases interleave inside)
        int y = ya_lo + ((phase - (ya_lo - y_lo) % phases) % phases + phases) % phases;
        if (full) {
            // The loop is VALU-issue bound (~290 instructions per 16 pixels: 64 byte->float converts, 144 blend
            // flops, 16 LDS reads); an explicit 2-row software pipeline measured 11 % SLOWER than letting the
            // other resident waves cover the load latency, so the row loop stays simple.
            auto do_row = [&](int yy, const u32x4& q) {
                const float tyf = tile_coord<FMA>(yy, g.inv_th);
                const float ya = __fsub_rn(tyf, (float)ty1u), ya1 = __fsub_rn(1.0f, ya);
                *reinterpret_cast<u32x4_u*>(dst + (long long)yy * p.dst_step + x0) =
                    FT ? clahe_vec16_f32<FMA>(quadf, q, poff, xw, ya, ya1) : clahe_vec16<FMA>(quad, q, poff, xa, xa1, ya, ya1);
            };
            constexpr int kRowsInFlight = FT ? 4 : 2;         // register budget: stay at 4 waves/SIMD (<= 128 VGPRs)
            // kRowsInFlight rows are loaded before the first is blended: a lane then has 64 B in flight instead of 16, and a
            // wave pays the HBM latency once per four rows (the kernel sits at 4 waves/SIMD, too few to hide it otherwise)
            for (; y + (kRowsInFlight - 1) * phases < ya_hi; y += kRowsInFlight * phases) {
                u32x4 q[kRowsInFlight];
#pragma unroll
                for (int k = 0; k < kRowsInFlight; ++k) q[k] = *reinterpret_cast<const u32x4_u*>(src + (long long)(y + k * phases) * p.src_step + x0);
#pragma unroll
                for (int k = 0; k < kRowsInFlight; ++k) { do_row(y + k * phases, q[k]); __builtin_amdgcn_sched_barrier(0); }
            }
            for (; y < ya_hi; y += phases) {
                const u32x4 q = *reinterpret_cast<const u32x4_u*>(src + (long long)y * p.src_step + x0);
                do_row(y, q);
            }
        } else {
            for (; y < ya_hi; y += phases) {
                const float tyf = tile_coord<FMA>(y, g.inv_th);
                const float ya = __fsub_rn(tyf, (float)ty1u), ya1 = __fsub_rn(1.0f, ya);
                const uint8_t* sr = src + (long long)y * p.src_step + x0;
                uint8_t* dr = dst + (long long)y * p.dst_step + x0;
#pragma unroll
                for (int j = 0; j < kInterpPx; ++j)
                    if (x0 + j < g.width) {
                        uint32_t e;
                        if (FT) {
                            const f32x4 fe = quadf[poff[j] + sr[j]];
                            e = (uint32_t)fe.x | ((uint32_t)fe.z << 8) | ((uint32_t)fe.y << 16) | ((uint32_t)fe.w << 24);
                        } else {
                            e = quad[poff[j] + sr[j]];
                        }
                        dr[j] = (uint8_t)clahe_px<FMA>(e, xa[j], xa1[j], ya, ya1);
                    }
            }
        }
    }
    if (uv.bytes > 0 && blockIdx.z == 0)
        uv_flat(uv.src + (long long)f * uv.src_frame, uv.dst + (long long)f * uv.dst_frame, uv.bytes, uv.mode, blockIdx.x, gridDim.x);
}

// Fallback for tile grids too wide for the LDS pair table: LUTs gathered from global memory (L2).
__global__ __launch_bounds__(kThreads) void clahe_interp_global_kernel(PlaneBatch p, ClaheGeom g, const uint8_t* __restrict__ luts)
{
    const int f = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= g.width) return;
    const uint8_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * 256;
    const float txf = tile_coord(x, g.inv_tw, g.contract);
    int tx1 = floor_f32_to_int(txf);
    const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
    int tx2 = tx1 + 1; tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
    const float tyf = tile_coord(y, g.inv_th, g.contract);
    int ty1 = floor_f32_to_int(tyf);
    const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
    int ty2 = ty1 + 1; ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
    const uint32_t v = p.src[(long long)f * p.src_frame + (long long)y * p.src_step + x];
    const uint32_t q = (uint32_t)lf[((size_t)ty1 * g.tiles_x + tx1) * 256 + v] |
                       ((uint32_t)lf[((size_t)ty1 * g.tiles_x + tx2) * 256 + v] << 8) |
                       ((uint32_t)lf[((size_t)ty2 * g.tiles_x + tx1) * 256 + v] << 16) |
                       ((uint32_t)lf[((size_t)ty2 * g.tiles_x + tx2) * 256 + v] << 24);
    p.dst[(long long)f * p.dst_frame + (long long)y * p.dst_step + x] =
        (uint8_t)(g.contract ? clahe_px<true>(q, xa, xa1, ya, ya1) : clahe_px<false>(q, xa, xa1, ya, ya1));
}

// UV-only launch (used when the Y kernel cannot carry the UV job).
__global__ __launch_bounds__(kThreads) void uv_kernel(UVJob uv)
{
    const int f = blockIdx.y;
    uv_flat(uv.src + (long long)f * uv.src_frame, uv.dst + (long long)f * uv.dst_frame, uv.bytes, uv.mode, blockIdx.x, gridDim.x);
}


}  // namespace mi
