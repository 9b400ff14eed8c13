// clahe16_wide.hip.h -- CLAHE on CV_16UC1 for content that populates MORE than 8192 values (14-bit sensors, full-range words,
// a 12-bit frame with a hot pixel): round 6's two kernels.  Part of the gfx950 kernel set of libmi_lumaeq; see clahe16.hip.h for the
// 10 / 12-bit paths, which these leave untouched.  Cost model and measurements: docs/experiments.md R6.1-R6.3.
//
// What the round-3 kernels paid on such content (16 4K frames, full range: 255 + 154 + 1276 us):
//   * a tile that loses tile_hist12_kernel's bet was swept once per 16384 values (four sweeps), its 65536 u32 counters went to HBM
//     (256 KiB per tile) and tile_lut16_kernel read them back chunk by chunk;
//   * the interpolation walked windows of 8192 table entries and, per window, re-read the workgroup's pixels from L2, ran the WHOLE
//     blend for every pixel of a wave as soon as one lane's pixel fell into the window, and stored results two bytes at a time.
// Here:
//   tile_hist16p_kernel   ONE sweep: 65536 counters of 16 bits, two per LDS word (128 KiB, one workgroup of 1024 threads per CU), the
//                         counter of value v in half (v >> 15) of word (v & 32767) -- neighbouring values in neighbouring banks.
//                         16 bits do not hold a tile (4K 8x8: 129 600 pixels): a counter that wraps loses 65535 or 65536 from the sum
//                         of all counters, so "sum == pixels" proves that none did; otherwise the tile is redone by the careful
//                         sweeps (tile_hist16_careful) in the same workgroup.  The LUT stage is folded in, over all 65536 values, raw
//                         domain: nothing but the LUT and the tile's range leaves the CU.
//   clahe_interp16_acc_kernel   a lane HOLDS its pixels (eight rows of eight) in registers over all table windows (16384 entries of
//                         {a | b << 16, c | d << 16}: 128 KiB, 512 threads with up to 256 VGPRs each).  Per window and pixel it does one subtract, one compare and
//                         one EXEC-masked ds_read_b64 into the pixel's accumulator -- lanes whose pixel lies in another window keep
//                         what they have -- and the blend runs ONCE per pixel, after the last window, all lanes busy, followed by
//                         16-byte stores.  In place is safe: every pixel a workgroup writes it has read before, and nobody else reads it.
#pragma once
#include "clahe16.hip.h"

namespace mi {

constexpr int kWideThreads = 1024;
constexpr int kWideWords = kHalf16;                  // 32768 LDS words = 65536 packed 16-bit counters
constexpr int kAccEntries = 16384;                   // table entries per window (8 bytes each: 128 KiB)
constexpr int kAccThreads = 512;                     // ONE workgroup per CU (128 KiB of LDS), two waves per SIMD: 256 VGPRs per lane
constexpr int kAccRows = 7;                          // rows of eight pixels a lane holds: 28 + 112 VGPRs (1024 threads x 4 rows spilled 76 VGPRs; a 4K band
                                                     // of 270 rows is five blocks of 54 = 7 rows x 8 phases either way, and eight rows spilled 29)

// one pixel: word v & 32767, low or high half by bit 15
__device__ __forceinline__ void hist16p_px(uint32_t* h, uint32_t v, uint32_t n)
{
    lds_add(h, v & 0x7fffu, n << ((v >> 11) & 16u));
}
__device__ __forceinline__ void hist16p_vec(uint32_t* h, const u32x4& q)
{
    const uint32_t v0 = q.x & 0xffffu;
    const bool flat = q.x == q.y && q.y == q.z && q.z == q.w && v0 == (q.x >> 16);
    if (__builtin_expect(flat, 0)) {                                // flat regions never reach the LDS pixel by pixel (as hist12_vec)
        const unsigned long long active = __ballot(1);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0);
        if (__ballot(v0 == first) == active) {
            if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(active)) hist16p_px(h, v0, 8u * (uint32_t)__builtin_popcountll(active));
        } else {
            hist16p_px(h, v0, 8u);
        }
        return;
    }
    hist16p_px(h, q.x & 0xffffu, 1u); hist16p_px(h, q.x >> 16, 1u); hist16p_px(h, q.y & 0xffffu, 1u); hist16p_px(h, q.y >> 16, 1u);
    hist16p_px(h, q.z & 0xffffu, 1u); hist16p_px(h, q.z >> 16, 1u); hist16p_px(h, q.w & 0xffffu, 1u); hist16p_px(h, q.w >> 16, 1u);
}

// PERSISTENT: grid = min(tiles x frames, CUs) workgroups of 1024 threads with 128 KiB of dynamic LDS (one per CU is all that fits);
// work items are (tile, frame) pairs.  Vector geometry only: tile_hist12_kernel ran before it and left the tiles that lost its bet
// marked kWideTodo, and frame_done[f] says whether a frame has any (2: all of its tiles, 3: some) -- frames without are skipped on one
// scalar load, so on 12-bit content this launch is 256 workgroups that look at a few words.
// `sync`: the per-frame 64-bit word NEXT to tile_hist12_kernel's (zero between launches): bits 48..63 how many of the frame's tiles
// tile_hist12_kernel left to this kernel (complete when this kernel starts), bits 0..15 how many of them have arrived, 16..31 how many
// wrote their LUT here, 32..47 which 4096-value buckets hold a tile's lowest / highest value.  A frame without left tiles is skipped
// on one scalar load.  The last left tile of a frame to arrive zeroes the word; if EVERY tile of the frame was left and all of them
// wrote their LUT here, it also settles the frame's range and frame_done = 1, so that tile_lut16_kernel leaves such frames on one scalar
// load.  Mixed frames go through tile_lut16_kernel, which returns early for the tiles written here (kLutFull) and extends the LUTs of
// their neighbours from the neighbours' own histograms.
__global__ __launch_bounds__(kWideThreads) void tile_hist16p_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                                   ClaheGeom g, uint32_t* __restrict__ hist, Range16* __restrict__ ranges,
                                                                   float lut_scale16, int clip16, uint16_t* __restrict__ luts,
                                                                   uint32_t* __restrict__ sync, Range16* __restrict__ frame_ranges,
                                                                   uint32_t* __restrict__ frame_done, uint32_t* __restrict__ shift_hint,
                                                                   int tiles, int n_frames)
{
    constexpr int NT = kWideThreads, NW = NT / 64;
    extern __shared__ uint32_t h16[];                               // [32768]
    __shared__ uint32_t s_lo, s_hi, s_or;
    __shared__ uint32_t s_tot[NW], s_exc[NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const int slots = g.tile_w >> 3;
    const int vitems = g.tile_h * slots;
    for (long long id = blockIdx.x; id < (long long)tiles * n_frames; id += gridDim.x) {
    const int f = (int)(id / tiles), tile = (int)(id - (long long)f * tiles);
    unsigned long long* const sy = reinterpret_cast<unsigned long long*>(sync) + 2 * (size_t)f + 1;
    const uint32_t left = (uint32_t)(__hip_atomic_load(sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 48);
    if (left == 0u) continue;                                       // uniform: no tile of this frame was left to this kernel
    const size_t tile_id = (size_t)f * tiles + tile;
    if (!(ranges[tile_id].hi & kWideTodo)) continue;                // uniform: not a tile that was left to this kernel
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    const uint8_t* tbase = src + (long long)ty * g.tile_h * step + (long long)tx * g.tile_w * 2;
    __syncthreads();                                                // the previous item's counters and sums are no longer read
    if (t == 0) { s_lo = 0xffffu; s_hi = 0u; }
    for (int i = t; i < kWideWords / 4; i += NT) reinterpret_cast<u32x4*>(h16)[i] = zero;
    __syncthreads();
    // ---- the sweep: (row, slot) items walked incrementally, two sets of four predicated 16-byte loads (as tile_hist12_kernel)
    {
        int row = t / slots, slot = t - row * slots;
        const int vdrow = NT / slots, vdslot = NT - vdrow * slots;
        auto load_set = [&](int it, u32x4* q, bool* qv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qv[k] = it + k * NT < vitems;
                const u32x4* ptr = reinterpret_cast<const u32x4*>(tbase + (long long)row * step + (slot << 4));
                q[k] = qv[k] ? *ptr : zero;
                row += vdrow; slot += vdslot;
                if (slot >= slots) { slot -= slots; ++row; }
            }
        };
        u32x4 cur[4], nxt[4]; bool cv[4], nv[4];
        load_set(t, cur, cv);
        for (int it = t; it < vitems; it += 4 * NT) {
            const bool more = it + 4 * NT < vitems;
            if (more) load_set(it + 4 * NT, nxt, nv);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (cv[k]) hist16p_vec(h16, cur[k]);
            if (more) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { cur[k] = nxt[k]; cv[k] = nv[k]; }
            }
        }
    }
    __syncthreads();
    // ---- pass 1: wave w owns values 4096 w .. 4096 w + 4095 (half w >> 3 of words 4096 (w & 7) ...): their total, their clip excess,
    // the lowest / highest populated value.  Lane-consecutive words: conflict-free.
    const uint32_t hsh = (uint32_t)(wv >> 3) << 4;                  // 0 or 16: which half of a word this wave's values live in
    const uint32_t wbase = (uint32_t)(wv & 7) * 4096u;
    {
        uint32_t tot = 0, exc = 0, first = 0xffffu, last = 0u;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            const uint32_t i = (uint32_t)lane + 64u * (uint32_t)k;
            const uint32_t c = (h16[wbase + i] >> hsh) & 0xffffu;
            tot += c;
            if (clip16 > 0 && (int)c > clip16) exc += c - (uint32_t)clip16;
            if (c) { const uint32_t b = (uint32_t)wv * 4096u + i; first = min(first, b); last = b; }      // i ascends with k
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            tot += (uint32_t)__shfl_xor((int)tot, d, 64); exc += (uint32_t)__shfl_xor((int)exc, d, 64);
            first = min(first, (uint32_t)__shfl_xor((int)first, d, 64)); last = max(last, (uint32_t)__shfl_xor((int)last, d, 64));
        }
        if (lane == 0) {
            s_tot[wv] = tot; s_exc[wv] = exc;
            if (tot) {
                __hip_atomic_fetch_min(&s_lo, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&s_hi, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    uint32_t all = 0, clipped = 0, before = 0, mask = 0;            // uniform
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t a = s_tot[k], e = s_exc[k];
        all += a; clipped += e;
        if (k < wv) before += a - e;                                // clipped counts of the values below this wave's
        if (a) mask |= 1u << k;
    }
    auto arrive_and_settle = [&](bool ok, uint32_t lo, uint32_t hi) {   // thread 0
        const uint32_t bits = ok ? (1u << (lo >> 12)) | (1u << (hi >> 12)) : 0u;
        if (ok) __hip_atomic_fetch_or(sy, (unsigned long long)bits << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b64 = __hip_atomic_fetch_add(sy, ok ? 0x10001ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(b64 & 0xffffu) != left - 1u) return;         // not the last of the frame's left tiles
        __hip_atomic_store(sy, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t nd = (uint32_t)((b64 >> 16) & 0xffffu) + (ok ? 1u : 0u);
        if (left != (uint32_t)tiles || nd != (uint32_t)tiles) return;   // a mixed frame, or a tile fell back: tile_lut16_kernel does the frame
        const uint32_t buckets = ((uint32_t)(b64 >> 32) | bits) & 0xffffu;
        Range16 r;
        r.lo = (uint32_t)__builtin_ctz(buckets) << 12;
        r.hi = ((31u - (uint32_t)__builtin_clz(buckets)) << 12) | 4095u;      // shift 0
        frame_ranges[f] = r;
        frame_done[f] = 1u;
        if (shift_hint) hint_out(shift_hint, 0u);
    };
    if (all != (uint32_t)(g.tile_w * g.tile_h)) {
        // a 16-bit counter wrapped (more than 65535 pixels of one value): the careful sweeps, 32768 u32 counters at a time, histogram to
        // memory, LUT by tile_lut16_kernel
        __syncthreads();
        tile_hist16_careful<15, NT>(h16, s_lo, s_hi, s_or, src_base, step, frame_stride, g, hist, ranges, 1, tile, f, tiles);
        if (t == 0) arrive_and_settle(false, 0u, 0u);
        continue;
    }
    // ---- clip, redistribute, prefix sum, scale: clahe.cpp for histSize 65536 (the closed form of tile_lut16_kernel, shift 0)
    int batch = 0, residual = 0;
    uint32_t rmagic = 0;                                            // floor(b / rstep) = mulhi(b, rmagic) for b < 65536 (rstep >= 2)
    bool rstep1 = false;
    if (clip16 > 0) {
        batch = (int)clipped / kHist16;
        residual = (int)clipped - batch * kHist16;
        if (residual != 0) {
            int rstep = kHist16 / residual; if (rstep < 1) rstep = 1;
            rstep1 = rstep == 1;
            rmagic = rstep1 ? 0u : 0xffffffffu / (uint32_t)rstep + 1u;
        }
    }
    uint16_t* const lut = luts + tile_id * kHist16;
    uint32_t running = before;                                      // clipped counts below the step's first value (uniform per wave)
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {                                  // 256 values per step, four consecutive ones per lane
        const uint32_t i0 = (uint32_t)s * 256u + (uint32_t)lane * 4u;
        const u32x4 q = *reinterpret_cast<const u32x4*>(h16 + wbase + i0);
        uint32_t c[4] = {(q.x >> hsh) & 0xffffu, (q.y >> hsh) & 0xffffu, (q.z >> hsh) & 0xffffu, (q.w >> hsh) & 0xffffu};
        uint32_t local = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (clip16 > 0 && (int)c[k] > clip16) c[k] = (uint32_t)clip16;
            local += c[k];
            c[k] = local;                                           // inclusive prefix within the lane's four values
        }
        const uint32_t incl = wave_incl_scan(local);
        const uint32_t base = running + incl - local;
        running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t packed[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t sum = base + c[k];
            if (clip16 > 0) {
                const uint32_t b = (uint32_t)wv * 4096u + i0 + (uint32_t)k;
                sum += (uint32_t)batch * (b + 1u);
                if (residual != 0) sum += min((uint32_t)residual, (rstep1 ? b : __umulhi(b, rmagic)) + 1u);
            }
            int r = __float2int_rn(__fmul_rn((float)(int)sum, lut_scale16));
            r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
            if (k & 1) packed[k >> 1] |= (uint32_t)r << 16; else packed[k >> 1] = (uint32_t)r;
        }
        *reinterpret_cast<uint2*>(lut + (uint32_t)wv * 4096u + i0) = make_uint2(packed[0], packed[1]);
    }
    if (t == 0) {
        const uint32_t lo = s_lo, hi = s_hi;
        Range16 r; r.lo = lo | ((mask & chunk_bits(lo, hi)) << 16); r.hi = hi | kLutFull; ranges[tile_id] = r;      // shift 0
        arrive_and_settle(true, lo, hi);
    }
    }
}

// ---- interpolation of the wide rectangles ------------------------------------------------------------------------------------------
// The table of a window is PLANAR -- four arrays of kAccEntries ushorts, one per tile LUT -- because that is what LDS-DMA can build:
// a wave copies 1 KiB of ONE LUT per global_load_lds_dwordx4, no VGPR in between, so all 128 KiB of a window are in flight at once
// and a window costs one trip to wherever the LUTs live (another XCD's tile kernel wrote them: Infinity Cache or HBM, ~2 us).  Staged
// through registers in {a | b << 16, c | d << 16} order the same window took four dependent trips (the 160 VGPRs of pixels and
// accumulators leave room for a quarter of it in flight): 8 us per window, the largest item of a block's 21 us (R6.2).
struct AccRect {
    size_t off[4];                                                  // the four tiles' LUTs, as element offsets into `luts`
    uint32_t sft, start, hi;
    int ty1u, phases, phase, x0;
    uint32_t own;
};

// one window [w0, w0 + n_w) of the four LUTs -> tab (planar).  Every wave issues its share of 1-KiB pieces; the caller's barrier
// waits for them (hipcc drains vmcnt before a __syncthreads).  A piece may run past the range's end: its source is clamped to stay
// inside the 65536-entry LUT, and what lands beyond n_w is never looked up.
__device__ __forceinline__ void acc_stage_window(uint16_t* tab, const uint16_t* __restrict__ luts, const AccRect& rc, uint32_t w0, uint32_t n_w)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t pieces = (n_w + 511u) >> 9;                      // per LUT
    for (uint32_t c = (uint32_t)wv; c < 4u * pieces; c += (uint32_t)(kAccThreads / 64)) {
        const uint32_t x = c & 3u, piece = c >> 2;
        const size_t off = x == 0u ? rc.off[0] : x == 1u ? rc.off[1] : x == 2u ? rc.off[2] : rc.off[3];      // uniform per wave
        const uint32_t v = min(w0 + piece * 512u + (uint32_t)lane * 8u, (uint32_t)kHist16 - 8u);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(luts + off + v),
                                         (void __attribute__((address_space(3)))*)(tab + x * (uint32_t)kAccEntries + piece * 512u), 16, 0, 0);
    }
}

// The four LUT values of one pixel from the planar table: four ds_read_u16 issued back to back, results NOT waited for.  Inline asm
// on purpose: written as C++ loads, `in ? table[i] : acc` came back from the compiler as a branch around the loads with an
// `s_waitcnt lgkmcnt(0)` after every pair -- 64 pixels x 2 full LDS round trips per window and wave.  The compiler does not know these
// registers are still in flight: lds_landed16() is the wait, and carries them as operands so that no use can be scheduled above it.
__device__ __forceinline__ void lds_read4_u16(uint32_t addr, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    asm volatile("ds_read_u16 %0, %4\n\tds_read_u16 %1, %4 offset:32768\n\tds_read_u16 %2, %5\n\tds_read_u16 %3, %5 offset:32768"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr), "v"(addr + 65536u));
}
#define MI_LANDED8(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
__device__ __forceinline__ void lds_landed16(uint32_t (&p)[8], uint32_t (&q)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : MI_LANDED8(p), MI_LANDED8(q));
}

// An item's rows are walked in blocks of what a workgroup holds: up to kAccRows rows of one 8-pixel group per lane (r_used of them in
// use, uniform).  Every lane's group is owned whole or not at all (the host launches this kernel only for geometries whose pair edges
// fall on multiples of eight pixels: 4K, 1080p, 720p at 8x8), so there is no per-pixel ownership anywhere: 16-byte loads and stores;
// rows beyond a block's end are loaded as zeros, looked up for nothing and neither blended nor stored.
__device__ __forceinline__ void acc_load_block(u32x4 (&q)[kAccRows], const AccRect& rc, const uint8_t* src, long long src_step, int yb, int y_end)
{
#pragma unroll
    for (int r = 0; r < kAccRows; ++r) {
        const int y = yb + rc.phase + r * rc.phases;
        q[r] = u32x4{0u, 0u, 0u, 0u};
        if (rc.own && y < y_end) q[r] = *reinterpret_cast<const u32x4*>(src + (long long)y * src_step + 2 * (long long)rc.x0);
    }
}

// SINGLE (uniform): the rectangle's whole range is ONE window, which the caller has staged once for all of the item's blocks.
// The pixels of the NEXT block are requested as soon as this block's last look-up is done (their registers are free from then on) and
// arrive while this block is blended.
template <bool FMA, bool SINGLE>
__device__ __forceinline__ void acc_item(uint16_t* tab, uint32_t* s_windows, const AccRect& rc, const ClaheGeom& g, const uint16_t* __restrict__ luts,
                                         const uint8_t* src, long long src_step, uint8_t* dst, long long dst_step,
                                         int y_lo, int y_hi, int rows_blk, int r_used)
{
    constexpr int R = kAccRows;
    const int t = threadIdx.x;
    const uint32_t wid = 16u - rc.sft;                              // a pixel >> shift: ONE v_bfe_u32 from the packed pair
    const uint32_t tab_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)tab;      // LDS byte address of the table
    u32x4 q[R];
    acc_load_block(q, rc, src, src_step, y_lo, min(y_lo + rows_blk, y_hi));
    if (SINGLE) __syncthreads();                                    // the single window's pieces, issued by the caller, have landed (vmcnt is drained
                                                                    // before a barrier) -- together with the first block's pixels
    for (int yb = y_lo; yb < y_hi; yb += rows_blk) {
        const int y_end = min(yb + rows_blk, y_hi);
        // ---- which windows do the block's pixels populate?  (a locally smooth image needs one or two of the four)
        uint32_t windows = 1u;
        if (!SINGLE) {
            if (t == 0) *s_windows = 0;
            __syncthreads();
            uint32_t seen = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = yb + rc.phase + r * rc.phases;
                if (!rc.own || y >= y_end) continue;
                const uint32_t w4[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    seen |= 1u << (((__builtin_amdgcn_ubfe(w4[j >> 1], (j & 1 ? 16u : 0u) + rc.sft, wid) - rc.start) / (uint32_t)kAccEntries) & 31u);
            }
            if (seen) __hip_atomic_fetch_or(s_windows, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            windows = *s_windows;
        }
        uint2 acc[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] = make_uint2(0u, 0u);
        for (uint32_t w0 = rc.start, wi = 0; w0 <= rc.hi; w0 += (uint32_t)kAccEntries, ++wi) {
            if (!((windows >> wi) & 1u)) continue;                  // uniform: none of the block's pixels lives in that window
            const uint32_t n_w = min(rc.hi - w0 + 1u, (uint32_t)kAccEntries);
            if (!SINGLE) {
                __syncthreads();                                    // the previous window's (or block's) table is no longer read
                acc_stage_window(tab, luts, rc, w0, n_w);
                __syncthreads();
            }
            // every lane reads its four entries, clamped into the window, and keeps them if its pixel lies there: a row's 32 reads in
            // flight, one wait, selects (no EXEC-masked loads, no branches)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r < r_used) {                                   // uniform
                    const uint32_t w4[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
                    uint32_t la[8], lb[8], lc[8], ld[8], idx[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        idx[j] = __builtin_amdgcn_ubfe(w4[j >> 1], (j & 1 ? 16u : 0u) + rc.sft, wid) - w0;
                        lds_read4_u16(tab_base + 2u * min(idx[j], n_w - 1u), la[j], lb[j], lc[j], ld[j]);
                    }
                    lds_landed16(la, lb);
                    lds_landed16(lc, ld);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t ex = la[j] | (lb[j] << 16), ey = lc[j] | (ld[j] << 16);
                        if (SINGLE) {
                            acc[r][j] = make_uint2(ex, ey);         // every pixel of the rectangle lies in this window
                        } else {
                            const bool in = idx[j] < n_w;
                            acc[r][j].x = in ? ex : acc[r][j].x; acc[r][j].y = in ? ey : acc[r][j].y;
                        }
                    }
                }
            }
        }
        // ---- the next block's pixels: on their way while this one is blended
        if (yb + rows_blk < y_hi) acc_load_block(q, rc, src, src_step, yb + rows_blk, min(yb + 2 * rows_blk, y_hi));
        // ---- the blend, once per pixel: {a, c} and {b, d} as float pairs, v_pk_mul / v_pk_add, every product and sum rounded on its own
        if (rc.own) {
            // the column weights are recomputed per block ON PURPOSE (56 instructions): hoisted out of the block loop they are 16 more
            // registers alive across the look-ups, which then spill (the empty asm hides from the compiler that x0 does not change)
            int x0v = rc.x0;
            asm volatile("" : "+v"(x0v));
            f32x2 xw[8];                                            // {xa1, xa} of the lane's eight columns
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float txf = tile_coord<FMA>(x0v + j, g.inv_tw);
                const float xa = __fsub_rn(txf, (float)floor_f32_to_int(txf));
                xw[j].x = __fsub_rn(1.0f, xa); xw[j].y = xa;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = yb + rc.phase + r * rc.phases;
                if (y < y_end) {
                    const float tyf = tile_coord<FMA>(y, g.inv_th);
                    const float ya = __fsub_rn(tyf, (float)rc.ty1u), ya1 = __fsub_rn(1.0f, ya);
                    const f32x2 yv = {ya1, ya};
                    uint32_t res[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint2 e = acc[r][j];
                        const f32x2 ac = {(float)(e.x & 0xffffu), (float)(e.y & 0xffffu)}, bd = {(float)(e.x >> 16), (float)(e.y >> 16)};
                        float v;
                        if (FMA) {
                            const f32x2 tb = pk_fma_bcast_lo(ac, xw[j], pk_mul_bcast_hi(bd, xw[j]));      // {fma(a,xa1,b*xa), fma(c,xa1,d*xa)}
                            v = __fmaf_rn(tb.x, ya1, __fmul_rn(tb.y, ya));
                        } else {
                            const f32x2 tb = (pk_mul_bcast_lo(ac, xw[j]) + pk_mul_bcast_hi(bd, xw[j])) * yv;  // nine individually rounded operations
                            v = __fadd_rn(tb.x, tb.y);
                        }
                        const int ri = __float2int_rn(v);
                        res[j] = (uint32_t)(ri < 0 ? 0 : (ri > 65535 ? 65535 : ri));
                    }
                    u32x4 o;
                    o.x = res[0] | (res[1] << 16); o.y = res[2] | (res[3] << 16); o.z = res[4] | (res[5] << 16); o.w = res[6] | (res[7] << 16);
                    *reinterpret_cast<u32x4*>(dst + (long long)y * dst_step + 2 * (long long)rc.x0) = o;
                }
                __builtin_amdgcn_sched_barrier(0);                  // one row at a time: the scheduler otherwise converts several rows' entries at once
            }
        }
    }
}

// PERSISTENT: grid = min(work items, CUs rounded to a multiple of 8) workgroups of 512 threads with 128 KiB of dynamic LDS -- one per CU
// is all that fits, and a launch of thousands of such workgroups that only return cost 23 us on 12-bit content.  Work items are
// those of clahe_interp16_kernel -- (tile pair, band, sub-band) rows of pairs, each row's pairs on ONE XCD (item id & 7; the grid is
// a multiple of 8, so a workgroup stays on its XCD) -- with this kernel's own, coarser `subs`: an item is walked in blocks of the rows a
// workgroup holds, and a rectangle whose whole range is ONE window (14-bit content) stages its table once for all of them.
// Which rectangles are this kernel's: rect_goes_wide() (clahe16.hip.h); clahe_interp16_kernel asks the same question and leaves them
// alone.  Frames whose whole range fits the small table are known from a bit mask built once per workgroup.
// The host launches it only when every plane and pitch is 16-byte aligned, every pair edge falls on a multiple of eight pixels and
// a pair has at most kAccThreads 8-pixel groups (clahe16.inc.hpp: acc_geometry_ok).
template <bool FMA>
__global__ __launch_bounds__(kAccThreads) void clahe_interp16_acc_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                        uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                        ClaheGeom g, const uint16_t* __restrict__ luts,
                                                                        const Range16* __restrict__ frame_ranges, int subs, int n_frames,
                                                                        const Range16* __restrict__ tile_ranges)
{
    static_assert(kAccEntries == kInterp16AccEntries, "rect_goes_wide() and the table agree on a window");
    constexpr int NT = kAccThreads, R = kAccRows;
    extern __shared__ __attribute__((aligned(16))) uint16_t tab16[];   // [4][kAccEntries]: the window's stretch of the four LUTs
    uint16_t* const tab = tab16;
    __shared__ uint32_t s_windows;
    __shared__ unsigned long long s_wide[16];                        // bit f % 64 of word f / 64: frame f may hold a wide rectangle (<= 1024 frames per launch)
    const int t = threadIdx.x;
    const int npairs = g.tiles_x + 1, bands = g.tiles_y + 1;
    const long long rows_total = (long long)bands * subs * n_frames;
    const long long items = (rows_total + 7) / 8 * 8 * npairs;
    // which frames can hold a wide rectangle at all?  (one vector load per 512 frames instead of a dependent scalar load per item: on
    // 12-bit content this launch is 256 workgroups that look at 16 words and leave)
    for (int f0 = 0; f0 < n_frames; f0 += NT) {
        const int f = f0 + t;
        bool wide = false;
        if (f < n_frames) {
            const Range16 fr = frame_ranges[f];
            const uint32_t sft = range_shift(fr.hi);
            wide = (range_hi(fr.hi) >> sft) - ((fr.lo >> sft) & ~3u) >= (uint32_t)kInterp16Entries;
        }
        const unsigned long long m = __ballot(wide);
        if ((t & 63) == 0 && f0 + t < 1024) s_wide[(f0 + t) >> 6] = m;
    }
    __syncthreads();
    for (long long id = blockIdx.x; id < items; id += gridDim.x) {
        const int xcd = (int)(id & 7);
        const long long k8 = id >> 3;
        const int pr = (int)(k8 % npairs);
        const long long row = (k8 / npairs) * 8 + xcd;
        if (row >= rows_total) continue;
        const int sub = (int)(row % subs), band = (int)((row / subs) % bands), f = n_frames - 1 - (int)(row / ((long long)subs * bands));
        if (!((s_wide[(f >> 6) & 15] >> (f & 63)) & 1ull)) continue;  // uniform: no wide rectangle in this frame
        AccRect rc;
        rc.sft = range_shift(frame_ranges[f].hi);
        rc.ty1u = band - 1;
        const int ty1 = max(rc.ty1u, 0), ty2 = min(rc.ty1u + 1, g.tiles_y - 1);
        const int tx1 = max(pr - 1, 0), tx2 = min(pr, g.tiles_x - 1);
        uint32_t lo;
        bool any_full;
        {
            const Range16* tr = tile_ranges + (size_t)f * g.tiles_x * g.tiles_y;
            const Range16 r00 = tr[ty1 * g.tiles_x + tx1], r01 = tr[ty1 * g.tiles_x + tx2], r10 = tr[ty2 * g.tiles_x + tx1], r11 = tr[ty2 * g.tiles_x + tx2];
            lo = min(min(range_lo(r00.lo), range_lo(r01.lo)), min(range_lo(r10.lo), range_lo(r11.lo))) >> rc.sft;
            rc.hi = max(max(range_hi(r00.hi), range_hi(r01.hi)), max(range_hi(r10.hi), range_hi(r11.hi))) >> rc.sft;
            any_full = ((r00.hi | r01.hi | r10.hi | r11.hi) & kLutFull) != 0u;
        }
        if (!rect_goes_wide(lo & ~3u, rc.hi, any_full, src_base == dst_base)) continue;     // clahe_interp16_kernel's rectangle (the same question, the same numbers)
        rc.start = lo & ~7u;                                        // windows start at a multiple of eight values: 16-byte aligned pieces
        // (uniform offsets into `luts`, made scalar by hand: left to itself the compiler kept four pointers in VGPR pairs and spilled them)
        auto scalar_off = [](size_t v) {
            const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
            return ((size_t)hi32 << 32) | lo32;
        };
        const size_t lf = (size_t)f * g.tiles_x * g.tiles_y * kHist16;
        rc.off[0] = scalar_off(lf + ((size_t)ty1 * g.tiles_x + tx1) * kHist16);
        rc.off[1] = scalar_off(lf + ((size_t)ty1 * g.tiles_x + tx2) * kHist16);
        rc.off[2] = scalar_off(lf + ((size_t)ty2 * g.tiles_x + tx1) * kHist16);
        rc.off[3] = scalar_off(lf + ((size_t)ty2 * g.tiles_x + tx2) * kHist16);
        // rows of the band and columns of the pair, exactly as clahe_interp16_kernel finds them
        const int y_lo_band = (int)max(0LL, ((long long)(2 * band - 1) * g.tile_h) / 2 - kBandMargin);
        const int y_hi_band = (int)min((long long)g.height, ((long long)(2 * band + 1) * g.tile_h + 1) / 2 + kBandMargin);
        const int nrows = max(0, y_hi_band - y_lo_band);
        int y_lo = y_lo_band + (int)((long long)nrows * sub / subs);
        int y_hi = y_lo_band + (int)((long long)nrows * (sub + 1) / subs);
        auto ty1_of = [&](int y) { return floor_f32_to_int(tile_coord<FMA>(y, g.inv_th)); };
        while (y_lo < y_hi && ty1_of(y_lo) != rc.ty1u) ++y_lo;
        while (y_hi > y_lo && ty1_of(y_hi - 1) != rc.ty1u) --y_hi;
        const int x_lo = (int)max(0LL, ((long long)(2 * pr - 1) * g.tile_w) / 2 - kBandMargin);
        const int x_hi = (int)min((long long)g.width, ((long long)(2 * pr + 1) * g.tile_w + 1) / 2 + kBandMargin);
        if (x_lo >= x_hi || y_lo >= y_hi) continue;                 // uniform over the workgroup
        const int g_lo = x_lo >> 3, ngroups = min(((x_hi + 7) >> 3) - g_lo, NT);
        rc.phases = max(1, NT / ngroups);
        const uint8_t* src = src_base + (long long)f * src_frame;
        uint8_t* dst = dst_base + (long long)f * dst_frame;
        const bool single = rc.hi - rc.start < (uint32_t)kAccEntries;     // uniform
        __syncthreads();                                            // the previous item's table is no longer read
        if (single) acc_stage_window(tab, luts, rc, rc.start, rc.hi - rc.start + 1u);      // in flight while the first block's pixels are fetched
        // the item's rows in equal blocks of at most R * phases (what a workgroup holds)
        const int rows_item = y_hi - y_lo, rows_cap = R * rc.phases;
        const int nblocks = (rows_item + rows_cap - 1) / rows_cap;
        const int rows_blk = (rows_item + nblocks - 1) / nblocks;
        const int r_used = (rows_blk + rc.phases - 1) / rc.phases;
        const int gi = t % ngroups;
        rc.phase = t / ngroups;
        rc.x0 = (g_lo + gi) << 3;
        rc.own = 0;
        if (rc.phase < rc.phases) {                                 // does this pair own the lane's group?  (whole or not at all)
            int q = floor_f32_to_int(tile_coord<FMA>(rc.x0, g.inv_tw)) + 1;
            q = q < 0 ? 0 : (q > g.tiles_x ? g.tiles_x : q);
            if (q == pr && rc.x0 + 8 <= g.width) rc.own = 0xffu;
        }
        if (single) {
            acc_item<FMA, true>(tab, &s_windows, rc, g, luts, src, src_step, dst, dst_step, y_lo, y_hi, rows_blk, r_used);
        } else {
            acc_item<FMA, false>(tab, &s_windows, rc, g, luts, src, src_step, dst, dst_step, y_lo, y_hi, rows_blk, r_used);
        }
    }
}

}  // namespace mi
