// clahe16.hip.h -- CLAHE on CV_16UC1 (65536 bins), SURVEY 8f N4
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "common.hip.h"
#include "clahe.hip.h"

namespace mi {
// =============================================================================================
// CLAHE on CV_16UC1 (clahe.cpp CLAHE_CalcLut_Body<ushort,65536,0> / CLAHE_Interpolation_Body<ushort,0>).
// Not on the reference's path (OpenCV surface beyond it).  65 536 u32 bins do not fit LDS but half of them do, so a
// tile's histogram is built in two LDS passes by one workgroup; the clip / redistribute / scan walks the bins in
// coalesced chunks of 1024; the interpolation gathers its four ushort LUT entries from L2.
// =============================================================================================
constexpr int kHist16 = 65536;

// grid = (tiles, frames), 1024 threads, one workgroup per tile.  65 536 u32 counters do not fit LDS, half of them do:
// two passes over the tile (the second one is served by L2), each histogramming one half of the value range in
// 128 KiB of LDS and storing it -- no global atomics, no zeroing of the output.  steps in BYTES.
constexpr int kHalf16 = 32768;
__global__ __launch_bounds__(1024) void tile_hist16_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                          ClaheGeom g, uint32_t* __restrict__ hist)
{
    extern __shared__ uint32_t h16[];                            // [32768]
    const int t = threadIdx.x;
    const int tile = blockIdx.x, f = blockIdx.y;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    uint32_t* out = hist + ((size_t)f * gridDim.x + tile) * kHist16;
    const long long items = (long long)g.tile_h * g.tile_w;
    const int drow = 1024 / g.tile_w, dcol = 1024 - drow * g.tile_w;
    for (int half = 0; half < 2; ++half) {
        for (int i = t; i < kHalf16; i += 1024) h16[i] = 0;
        __syncthreads();
        int row = t / g.tile_w, col = t - row * g.tile_w;
        for (long long it = t; it < items; it += 1024) {
            const int y = reflect101(ty * g.tile_h + row, g.height);
            const int x = reflect101(tx * g.tile_w + col, g.width);
            const uint32_t v = *reinterpret_cast<const uint16_t*>(src + (long long)y * step + 2 * (long long)x);
            if ((int)(v >> 15) == half) lds_inc(h16, v & (kHalf16 - 1));
            row += drow; col += dcol;
            if (col >= g.tile_w) { col -= g.tile_w; ++row; }
        }
        __syncthreads();
        for (int i = t; i < kHalf16; i += 1024) out[half * kHalf16 + i] = h16[i];
        __syncthreads();
    }
}

// grid = (tiles, frames), 1024 threads.  The 65 536 bins are walked in 64 chunks of 1024 (coalesced): a first sweep
// sums the clipped excess, a second applies clip + redistribute and scans (block scan per chunk + running offset).
__global__ __launch_bounds__(1024) void tile_lut16_kernel(const uint32_t* __restrict__ hist, ClaheGeom g, float lut_scale16, int clip16,
                                                         uint16_t* __restrict__ luts)
{
    __shared__ uint32_t s_w[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const size_t tile_id = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t* h = hist + tile_id * kHist16;
    uint16_t* lut = luts + tile_id * kHist16;
    auto block_scan = [&](uint32_t v, uint32_t& total) -> uint32_t {     // inclusive prefix of v over the 1024 threads
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        __syncthreads();
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { const uint32_t x = s_w[k]; if (k < w) off += x; tot += x; }
        total = tot;
        return off + incl;
    };
    int batch = 0, residual = 0, rstep = 1;
    if (clip16 > 0) {
        uint32_t excess = 0;
        for (int c = 0; c < 64; ++c) { const int v = (int)h[c * 1024 + t]; if (v > clip16) excess += (uint32_t)(v - clip16); }
        uint32_t clipped;
        (void)block_scan(excess, clipped);
        batch = (int)clipped / kHist16;
        residual = (int)clipped - batch * kHist16;
        if (residual != 0) { rstep = kHist16 / residual; if (rstep < 1) rstep = 1; }
    }
    uint32_t running = 0;
    for (int c = 0; c < 64; ++c) {
        const int b = c * 1024 + t;
        int v = (int)h[b];
        if (clip16 > 0) {
            if (v > clip16) v = clip16;
            v += batch;
            if (residual != 0 && b % rstep == 0 && b / rstep < residual) ++v;
        }
        uint32_t total;
        const uint32_t sum = running + block_scan((uint32_t)v, total);
        running += total;
        int r = __float2int_rn(__fmul_rn((float)(int)sum, lut_scale16));
        r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
        lut[b] = (uint16_t)r;
    }
}

// grid = (ceil(W/256), H, frames): one pixel per lane, four ushort gathers from the per-tile LUTs (L2).
__global__ __launch_bounds__(kThreads) void clahe_interp16_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                 uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                 ClaheGeom g, const uint16_t* __restrict__ luts)
{
    const int f = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= g.width) return;
    const uint16_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * kHist16;
    const float txf = __fsub_rn(__fmul_rn((float)x, g.inv_tw), 0.5f);
    int tx1 = floor_f32_to_int(txf);
    const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
    int tx2 = tx1 + 1; tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
    const float tyf = __fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f);
    int ty1 = floor_f32_to_int(tyf);
    const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
    int ty2 = ty1 + 1; ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
    const uint32_t v = *reinterpret_cast<const uint16_t*>(src_base + (long long)f * src_frame + (long long)y * src_step + 2 * (long long)x);
    const float a = (float)lf[((size_t)ty1 * g.tiles_x + tx1) * kHist16 + v], b = (float)lf[((size_t)ty1 * g.tiles_x + tx2) * kHist16 + v];
    const float c = (float)lf[((size_t)ty2 * g.tiles_x + tx1) * kHist16 + v], d = (float)lf[((size_t)ty2 * g.tiles_x + tx2) * kHist16 + v];
    const float top = __fmul_rn(__fadd_rn(__fmul_rn(a, xa1), __fmul_rn(b, xa)), ya1);
    const float bot = __fmul_rn(__fadd_rn(__fmul_rn(c, xa1), __fmul_rn(d, xa)), ya);
    int r = __float2int_rn(__fadd_rn(top, bot));
    r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
    *reinterpret_cast<uint16_t*>(dst_base + (long long)f * dst_frame + (long long)y * dst_step + 2 * (long long)x) = (uint16_t)r;
}

}  // namespace mi
