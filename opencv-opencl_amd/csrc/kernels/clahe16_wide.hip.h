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
constexpr int kAccRows = 8;                          // rows of eight pixels a lane holds: 32 + 128 VGPRs (1024 threads x 4 rows spilled 76)

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

// grid = (tiles, frames), 1024 threads, 128 KiB of dynamic LDS; vector geometry only (tile_hist12_kernel ran before it on the same
// grid and left the tiles that lost its bet marked kWideTodo; everybody else returns on one scalar load).
// `sync`: the per-frame 64-bit word NEXT to tile_hist12_kernel's (zero between launches): bits 0..15 arrivals, 16..31 tiles that
// wrote their LUT here, 32..47 which 4096-value buckets hold a tile's lowest / highest value.  Used only for frames in which EVERY
// tile was left to this kernel (frame_done == 2): their last tile to arrive settles the frame's range and frame_done = 1, so that
// tile_lut16_kernel leaves such frames on one scalar load.  Mixed frames go through tile_lut16_kernel, which returns early for the
// tiles written here (kLutFull) and extends the LUTs of their 12-bit neighbours from the neighbours' own histograms.
__global__ __launch_bounds__(kWideThreads) void tile_hist16p_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                                   ClaheGeom g, uint32_t* __restrict__ hist, Range16* __restrict__ ranges,
                                                                   float lut_scale16, int clip16, uint16_t* __restrict__ luts,
                                                                   uint32_t* __restrict__ sync, Range16* __restrict__ frame_ranges,
                                                                   uint32_t* __restrict__ frame_done, uint32_t* __restrict__ shift_hint)
{
    constexpr int NT = kWideThreads, NW = NT / 64;
    extern __shared__ uint32_t h16[];                               // [32768]
    __shared__ uint32_t s_lo, s_hi, s_or;
    __shared__ uint32_t s_tot[NW], s_exc[NW];
    const int tile = blockIdx.x, f = blockIdx.y;
    const size_t tile_id = (size_t)f * gridDim.x + tile;
    if (!(ranges[tile_id].hi & kWideTodo)) return;                  // uniform: not a tile that was left to this kernel
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    const int slots = g.tile_w >> 3;
    const int vitems = g.tile_h * slots;
    const uint8_t* tbase = src + (long long)ty * g.tile_h * step + (long long)tx * g.tile_w * 2;
    const u32x4 zero = {0u, 0u, 0u, 0u};
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
    unsigned long long* const sy = reinterpret_cast<unsigned long long*>(sync) + 2 * (size_t)f + 1;
    const bool frame_mine = frame_done[f] == 2u;                    // every tile of the frame is here: the last one settles it
    auto arrive_and_settle = [&](bool ok, uint32_t lo, uint32_t hi) {   // thread 0
        if (!frame_mine) return;
        const uint32_t bits = ok ? (1u << (lo >> 12)) | (1u << (hi >> 12)) : 0u;
        if (ok) __hip_atomic_fetch_or(sy, (unsigned long long)bits << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b64 = __hip_atomic_fetch_add(sy, ok ? 0x10001ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(b64 & 0xffffu) != gridDim.x - 1) return;
        __hip_atomic_store(sy, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t nd = (uint32_t)((b64 >> 16) & 0xffffu) + (ok ? 1u : 0u);
        if (nd != gridDim.x) { frame_done[f] = 0u; return; }        // some tile fell back: tile_lut16_kernel does the frame
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
        tile_hist16_careful<15, NT>(h16, s_lo, s_hi, s_or, src_base, step, frame_stride, g, hist, ranges, 1);
        if (t == 0) arrive_and_settle(false, 0u, 0u);
        return;
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

// grid = 8 * ceil(rows / 8) * (tiles_x + 1) workgroups as clahe_interp16_kernel (rows of pairs dealt to XCDs whole), with its own
// `subs`; 512 threads, 128 KiB of dynamic LDS.  Takes the rectangles whose four tiles populate a range of 8192 values or more and
// leaves the others to clahe_interp16_kernel (which is told to leave these alone).  The host launches it only when every plane and
// pitch is 16-byte aligned.
template <bool FMA>
__global__ __launch_bounds__(kAccThreads) void clahe_interp16_acc_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                        uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                        ClaheGeom g, const uint16_t* __restrict__ luts,
                                                                        const Range16* __restrict__ frame_ranges, int subs, int n_frames,
                                                                        const Range16* __restrict__ tile_ranges)
{
    constexpr int NT = kAccThreads, R = kAccRows;
    extern __shared__ __attribute__((aligned(16))) uint2 tab[];      // [kAccEntries] {a | b << 16, c | d << 16}
    __shared__ uint32_t s_windows;
    const int t = threadIdx.x;
    const int npairs = g.tiles_x + 1, bands = g.tiles_y + 1;
    const long long id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const long long k8 = id >> 3;
    const int pr = (int)(k8 % npairs);
    const long long row = (k8 / npairs) * 8 + xcd;
    if (row >= (long long)bands * subs * n_frames) return;
    const int sub = (int)(row % subs), band = (int)((row / subs) % bands), f = n_frames - 1 - (int)(row / ((long long)subs * bands));
    const int ty1u = band - 1;
    const int ty1 = max(ty1u, 0), ty2 = min(ty1u + 1, g.tiles_y - 1);
    const int tx1 = max(pr - 1, 0), tx2 = min(pr, g.tiles_x - 1);
    const uint32_t sft = range_shift(frame_ranges[f].hi);
    uint32_t lo, hi;
    {
        const Range16* tr = tile_ranges + (size_t)f * g.tiles_x * g.tiles_y;
        const Range16 r00 = tr[ty1 * g.tiles_x + tx1], r01 = tr[ty1 * g.tiles_x + tx2], r10 = tr[ty2 * g.tiles_x + tx1], r11 = tr[ty2 * g.tiles_x + tx2];
        lo = min(min(range_lo(r00.lo), range_lo(r01.lo)), min(range_lo(r10.lo), range_lo(r11.lo))) >> sft;
        hi = max(max(range_hi(r00.hi), range_hi(r01.hi)), max(range_hi(r10.hi), range_hi(r11.hi))) >> sft;
    }
    const uint32_t start = lo & ~3u;
    if (hi - start < (uint32_t)kInterp16Entries) return;             // one window of the small table: clahe_interp16_kernel's rectangle
    const uint16_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * kHist16;
    const uint16_t* la = lf + ((size_t)ty1 * g.tiles_x + tx1) * kHist16;
    const uint16_t* lb = lf + ((size_t)ty1 * g.tiles_x + tx2) * kHist16;
    const uint16_t* lc = lf + ((size_t)ty2 * g.tiles_x + tx1) * kHist16;
    const uint16_t* ld = lf + ((size_t)ty2 * g.tiles_x + tx2) * kHist16;

    // rows of the band and columns of the pair, exactly as clahe_interp16_kernel finds them
    const int y_lo_band = (int)max(0LL, ((long long)(2 * band - 1) * g.tile_h) / 2 - kBandMargin);
    const int y_hi_band = (int)min((long long)g.height, ((long long)(2 * band + 1) * g.tile_h + 1) / 2 + kBandMargin);
    const int nrows = max(0, y_hi_band - y_lo_band);
    int y_lo = y_lo_band + (int)((long long)nrows * sub / subs);
    int y_hi = y_lo_band + (int)((long long)nrows * (sub + 1) / subs);
    auto ty1_of = [&](int y) { return floor_f32_to_int(tile_coord<FMA>(y, g.inv_th)); };
    while (y_lo < y_hi && ty1_of(y_lo) != ty1u) ++y_lo;
    while (y_hi > y_lo && ty1_of(y_hi - 1) != ty1u) --y_hi;
    const int x_lo = (int)max(0LL, ((long long)(2 * pr - 1) * g.tile_w) / 2 - kBandMargin);
    const int x_hi = (int)min((long long)g.width, ((long long)(2 * pr + 1) * g.tile_w + 1) / 2 + kBandMargin);
    if (x_lo >= x_hi || y_lo >= y_hi) return;                       // uniform over the workgroup
    const int g_lo = x_lo >> 3, ngroups = ((x_hi + 7) >> 3) - g_lo;
    const int phases = max(1, NT / ngroups);
    const uint8_t* src = src_base + (long long)f * src_frame;
    uint8_t* dst = dst_base + (long long)f * dst_frame;

    for (int gbase = 0; gbase < ngroups; gbase += NT) {             // more than one pass only for pairs wider than 8192 pixels
        const int gi = gbase + (ngroups > NT ? t : t % ngroups);
        const int phase = ngroups > NT ? 0 : t / ngroups;
        const int x0 = (g_lo + gi) << 3;
        uint32_t own = 0;
        if (gi < ngroups && phase < phases) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int q = floor_f32_to_int(tile_coord<FMA>(x0 + j, g.inv_tw)) + 1;
                q = q < 0 ? 0 : (q > g.tiles_x ? g.tiles_x : q);
                if (q == pr && x0 + j < g.width) own |= 1u << j;
            }
        }
        const bool full = own == 0xffu;                              // (then x0 + 8 <= width: all eight lie inside the frame)
        for (int yb = y_lo; yb < y_hi; yb += R * phases) {          // more than one block only if the host gave the workgroup more rows than it holds
            // ---- the lane's pixels: R rows of one 8-pixel group
            u32x4 q[R];
            uint32_t rowok = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = yb + phase + r * phases;
                q[r] = u32x4{0u, 0u, 0u, 0u};
                if (!own || y >= y_hi) continue;
                rowok |= 1u << r;
                const uint8_t* sp = src + (long long)y * src_step + 2 * (long long)x0;
                if (full) {
                    q[r] = *reinterpret_cast<const u32x4*>(sp);
                } else {
                    uint32_t px[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) px[j] = (own >> j) & 1u ? (uint32_t)*reinterpret_cast<const uint16_t*>(sp + 2 * j) : 0u;
                    q[r] = u32x4{px[0] | (px[1] << 16), px[2] | (px[3] << 16), px[4] | (px[5] << 16), px[6] | (px[7] << 16)};
                }
            }
            // ---- which windows do they populate?  (a locally smooth image needs one or two of the four)
            if (t == 0) s_windows = 0;
            __syncthreads();
            {
                uint32_t seen = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!((rowok >> r) & 1u)) continue;
                    const uint32_t w4[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pv = (j & 1) ? (w4[j >> 1] >> 16) : (w4[j >> 1] & 0xffffu);
                        if ((own >> j) & 1u) seen |= 1u << ((((pv >> sft) - start) / (uint32_t)kAccEntries) & 31u);
                    }
                }
                if (seen) __hip_atomic_fetch_or(&s_windows, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            const uint32_t windows = s_windows;
            uint2 acc[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r][j] = make_uint2(0u, 0u);
            for (uint32_t w0 = start, wi = 0; w0 <= hi; w0 += (uint32_t)kAccEntries, ++wi) {
                if (!((windows >> wi) & 1u)) continue;              // uniform: none of the block's pixels lives in that window
                __syncthreads();                                    // the previous window's table is no longer read
                const uint32_t n_w = min(hi - w0 + 1u, (uint32_t)kAccEntries);
                {
                    // four entries per lane and step from four 8-byte loads, written as two 16-byte stores; a window starts at a multiple
                    // of four values and ends at most three entries past the range (still inside the 65536-entry LUTs, never looked up)
                    const uint32_t n4 = (n_w + 3u) & ~3u;
#pragma unroll 2
                    for (uint32_t i0 = (uint32_t)t * 4u; i0 < n4; i0 += (uint32_t)NT * 4u) {
                        const uint32_t v = w0 + i0;
                        const uint2 A = *reinterpret_cast<const uint2*>(la + v), B = *reinterpret_cast<const uint2*>(lb + v);
                        const uint2 C = *reinterpret_cast<const uint2*>(lc + v), D = *reinterpret_cast<const uint2*>(ld + v);
                        u32x4 e0, e1;
                        e0.x = (A.x & 0xffffu) | (B.x << 16);         e0.y = (C.x & 0xffffu) | (D.x << 16);
                        e0.z = (A.x >> 16) | (B.x & 0xffff0000u);     e0.w = (C.x >> 16) | (D.x & 0xffff0000u);
                        e1.x = (A.y & 0xffffu) | (B.y << 16);         e1.y = (C.y & 0xffffu) | (D.y << 16);
                        e1.z = (A.y >> 16) | (B.y & 0xffff0000u);     e1.w = (C.y >> 16) | (D.y & 0xffff0000u);
                        u32x4* o = reinterpret_cast<u32x4*>(tab + i0);
                        o[0] = e0; o[1] = e1;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!((rowok >> r) & 1u)) continue;
                    const uint32_t w4[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pv = (j & 1) ? (w4[j >> 1] >> 16) : (w4[j >> 1] & 0xffffu);
                        const uint32_t idx = (pv >> sft) - w0;
                        if (idx < n_w) acc[r][j] = tab[idx];        // EXEC-masked ds_read_b64: the other lanes keep what they have
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- the blend, once per pixel
            float xa[8], xa1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float txf = tile_coord<FMA>(x0 + j, g.inv_tw);
                xa[j] = __fsub_rn(txf, (float)floor_f32_to_int(txf));
                xa1[j] = __fsub_rn(1.0f, xa[j]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!((rowok >> r) & 1u)) continue;
                const int y = yb + phase + r * phases;
                const float tyf = tile_coord<FMA>(y, g.inv_th);
                const float ya = __fsub_rn(tyf, (float)ty1u), ya1 = __fsub_rn(1.0f, ya);
                uint32_t res[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint2 e = acc[r][j];
                    const float a = (float)(e.x & 0xffffu), b = (float)(e.x >> 16), c = (float)(e.y & 0xffffu), d = (float)(e.y >> 16);
                    int v = __float2int_rn(clahe_blend_f<FMA>(a, b, c, d, xa[j], xa1[j], ya, ya1));
                    res[j] = (uint32_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
                }
                uint8_t* dp = dst + (long long)y * dst_step + 2 * (long long)x0;
                if (full) {
                    u32x4 o;
                    o.x = res[0] | (res[1] << 16); o.y = res[2] | (res[3] << 16); o.z = res[4] | (res[5] << 16); o.w = res[6] | (res[7] << 16);
                    *reinterpret_cast<u32x4*>(dp) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if ((own >> j) & 1u) *reinterpret_cast<uint16_t*>(dp + 2 * j) = (uint16_t)res[j];
                }
                __builtin_amdgcn_sched_barrier(0);                  // one row at a time: the scheduler otherwise converts several rows' entries at once
            }
        }
    }
}

}  // namespace mi
