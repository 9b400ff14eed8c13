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
__device__ __forceinline__ void hist16_add_dword(uint32_t* h16, uint32_t w, int half)
{
    const uint32_t a = w & 0xffffu, b = w >> 16;
    if ((int)(a >> 15) == half) lds_inc(h16, a & (kHalf16 - 1));
    if ((int)(b >> 15) == half) lds_inc(h16, b & (kHalf16 - 1));
}

// `vec` (host: no REFLECT_101 padding, tile_w % 8 == 0, 16-B aligned rows): a lane takes 8 pixels per 16-byte load with
// four loads in flight; otherwise one pixel per lane per step with index reflection.
__global__ __launch_bounds__(1024) void tile_hist16_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                          ClaheGeom g, uint32_t* __restrict__ hist, int vec)
{
    extern __shared__ uint32_t h16[];                            // [32768]
    const int t = threadIdx.x;
    const int tile = blockIdx.x, f = blockIdx.y;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    uint32_t* out = hist + ((size_t)f * gridDim.x + tile) * kHist16;
    const long long items = (long long)g.tile_h * g.tile_w;
    const int drow = 1024 / g.tile_w, dcol = 1024 - drow * g.tile_w;
    const int slots = g.tile_w >> 3;                              // 8-pixel groups per tile row (vector path)
    const int vitems = g.tile_h * slots;
    const uint8_t* tbase = src + (long long)ty * g.tile_h * step + (long long)tx * g.tile_w * 2;
    auto vload = [&](int it) -> u32x4 {
        const int row = it / slots, slot = it - row * slots;
        return *reinterpret_cast<const u32x4*>(tbase + (long long)row * step + (slot << 4));
    };
    auto vadd = [&](const u32x4& q, int half) {
        hist16_add_dword(h16, q.x, half); hist16_add_dword(h16, q.y, half);
        hist16_add_dword(h16, q.z, half); hist16_add_dword(h16, q.w, half);
    };
    for (int half = 0; half < 2; ++half) {
        for (int i = t; i < kHalf16; i += 1024) h16[i] = 0;
        __syncthreads();
        if (vec) {
            int it = t;
            for (; it + 3 * 1024 < vitems; it += 4 * 1024) {
                const u32x4 a = vload(it), b = vload(it + 1024), c = vload(it + 2048), d = vload(it + 3072);
                vadd(a, half); vadd(b, half); vadd(c, half); vadd(d, half);
            }
            for (; it < vitems; it += 1024) vadd(vload(it), half);
        } else {
            int row = t / g.tile_w, col = t - row * g.tile_w;
            for (long long it = t; it < items; it += 1024) {
                const int y = reflect101(ty * g.tile_h + row, g.height);
                const int x = reflect101(tx * g.tile_w + col, g.width);
                const uint32_t v = *reinterpret_cast<const uint16_t*>(src + (long long)y * step + 2 * (long long)x);
                if ((int)(v >> 15) == half) lds_inc(h16, v & (kHalf16 - 1));
                row += drow; col += dcol;
                if (col >= g.tile_w) { col -= g.tile_w; ++row; }
            }
        }
        __syncthreads();
        for (int i = t; i < kHalf16; i += 1024) out[half * kHalf16 + i] = h16[i];
        __syncthreads();
    }
}

// grid = (tiles, frames), 1024 threads.  The 65 536 bins are walked in 16 chunks of 4096, four consecutive bins per
// thread (one 16-byte load, one 8-byte store): a first sweep sums the clipped excess, a second applies clip +
// redistribute and scans (serial over a thread's four bins, block scan of the four-bin sums, running offset per chunk).
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
        for (int c = 0; c < 16; ++c) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(h + c * 4096 + t * 4);
            const int v[4] = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) if (v[k] > clip16) excess += (uint32_t)(v[k] - clip16);
        }
        uint32_t clipped;
        (void)block_scan(excess, clipped);
        batch = (int)clipped / kHist16;
        residual = (int)clipped - batch * kHist16;
        if (residual != 0) { rstep = kHist16 / residual; if (rstep < 1) rstep = 1; }
    }
    uint32_t running = 0;
    for (int c = 0; c < 16; ++c) {
        const int b0 = c * 4096 + t * 4;
        const u32x4 q = *reinterpret_cast<const u32x4*>(h + b0);
        int v[4] = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
        uint32_t local = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (clip16 > 0) {
                if (v[k] > clip16) v[k] = clip16;
                v[k] += batch;
                const int b = b0 + k;
                if (residual != 0 && b % rstep == 0 && b / rstep < residual) ++v[k];
            }
            local += (uint32_t)v[k];
            v[k] = (int)local;                                       // inclusive prefix within the thread's four bins
        }
        uint32_t total;
        const uint32_t before = running + block_scan(local, total) - local;     // everything before this thread's first bin
        running += total;
        uint32_t packed[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int r = __float2int_rn(__fmul_rn((float)(int)(before + (uint32_t)v[k]), lut_scale16));
            r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
            if (k & 1) packed[k >> 1] |= (uint32_t)r << 16; else packed[k >> 1] = (uint32_t)r;
        }
        *reinterpret_cast<uint2*>(lut + b0) = make_uint2(packed[0], packed[1]);
    }
}

// grid = (ceil(W/256), H, frames): one pixel per lane, four ushort gathers from the per-tile LUTs (L2).  Bound by the
// divergent gathers themselves (up to 64 cache lines per wave instruction): giving each XCD one eighth of the rows, so
// that its L2 only has to hold two tile rows of LUTs, measured 4 % SLOWER.
__global__ __launch_bounds__(kThreads) void clahe_interp16_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                 uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                 ClaheGeom g, const uint16_t* __restrict__ luts)
{
    const int f = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= g.width) return;
    const uint16_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * kHist16;
    const float txf = tile_coord(x, g.inv_tw, g.contract);
    int tx1 = floor_f32_to_int(txf);
    const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
    int tx2 = tx1 + 1; tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
    const float tyf = tile_coord(y, g.inv_th, g.contract);
    int ty1 = floor_f32_to_int(tyf);
    const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
    int ty2 = ty1 + 1; ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
    const uint32_t v = *reinterpret_cast<const uint16_t*>(src_base + (long long)f * src_frame + (long long)y * src_step + 2 * (long long)x);
    const float a = (float)lf[((size_t)ty1 * g.tiles_x + tx1) * kHist16 + v], b = (float)lf[((size_t)ty1 * g.tiles_x + tx2) * kHist16 + v];
    const float c = (float)lf[((size_t)ty2 * g.tiles_x + tx1) * kHist16 + v], d = (float)lf[((size_t)ty2 * g.tiles_x + tx2) * kHist16 + v];
    int r = __float2int_rn(g.contract ? clahe_blend_f<true>(a, b, c, d, xa, xa1, ya, ya1) : clahe_blend_f<false>(a, b, c, d, xa, xa1, ya, ya1));
    r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
    *reinterpret_cast<uint16_t*>(dst_base + (long long)f * dst_frame + (long long)y * dst_step + 2 * (long long)x) = (uint16_t)r;
}

// ---- value-major LUT layout for the interpolation ---------------------------------------------------------------
// The four entries a pixel needs -- LUT[ty1][tx1][v], [ty1][tx2][v], [ty2][tx1][v], [ty2][tx2][v] -- sit in four different
// 128 KiB tables, so a wave of 64 pixels pulls up to 256 cache lines for 512 useful bytes and the kernel is bound by the
// L2 -> L1 fills.  Transposed to lutT[v][tile] (tiles <= 64: one 128-byte line per value at 8x8) all four come from ONE
// line.  transpose_lut16_kernel does it through LDS with coalesced reads and writes (16.8 MB per 4K frame).
// Measured on 4 x 4K frames: full-range noise 447 -> 296 us (+ 41 us for the transpose), but 400-level content 125 -> 241 us:
// a line now carries all 64 tiles' entries for a value and only four are used, so content with a narrow local range --
// the usual case for 10/12-bit sensors -- overflows L1 sixteen times sooner.  Hence an OPTION ("clahe16_transposed"), off by
// default.
// grid = (65536 / 256, frames); 256 threads; LDS = tiles * 256 ushorts.
__global__ __launch_bounds__(kThreads) void transpose_lut16_kernel(const uint16_t* __restrict__ luts, uint16_t* __restrict__ lutT, int tiles)
{
    extern __shared__ uint16_t tr[];                             // [tiles][256]
    const int t = threadIdx.x, f = blockIdx.y, v0 = blockIdx.x * 256;
    const uint16_t* src = luts + (size_t)f * tiles * kHist16 + v0;
    for (int k = 0; k < tiles; ++k) tr[k * 256 + t] = src[(size_t)k * kHist16 + t];
    __syncthreads();
    uint16_t* dst = lutT + ((size_t)f * kHist16 + v0) * tiles;
    for (int i = t; i < tiles * 256; i += kThreads) {             // i = v_local * tiles + tile, consecutive in memory
        const int vl = i / tiles, k = i - vl * tiles;
        dst[i] = tr[k * 256 + vl];
    }
}

// as clahe_interp16_kernel, gathering from the value-major layout
__global__ __launch_bounds__(kThreads) void clahe_interp16T_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                  uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                  ClaheGeom g, const uint16_t* __restrict__ lutT)
{
    const int f = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= g.width) return;
    const int tiles = g.tiles_x * g.tiles_y;
    const float txf = tile_coord(x, g.inv_tw, g.contract);
    int tx1 = floor_f32_to_int(txf);
    const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
    int tx2 = tx1 + 1; tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
    const float tyf = tile_coord(y, g.inv_th, g.contract);
    int ty1 = floor_f32_to_int(tyf);
    const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
    int ty2 = ty1 + 1; ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
    const uint32_t v = *reinterpret_cast<const uint16_t*>(src_base + (long long)f * src_frame + (long long)y * src_step + 2 * (long long)x);
    const uint16_t* e = lutT + ((size_t)f * kHist16 + v) * tiles;
    const float a = (float)e[ty1 * g.tiles_x + tx1], b = (float)e[ty1 * g.tiles_x + tx2];
    const float c = (float)e[ty2 * g.tiles_x + tx1], d = (float)e[ty2 * g.tiles_x + tx2];
    int r = __float2int_rn(g.contract ? clahe_blend_f<true>(a, b, c, d, xa, xa1, ya, ya1) : clahe_blend_f<false>(a, b, c, d, xa, xa1, ya, ya1));
    r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
    *reinterpret_cast<uint16_t*>(dst_base + (long long)f * dst_frame + (long long)y * dst_step + 2 * (long long)x) = (uint16_t)r;
}

}  // namespace mi
