// clahe16.hip.h -- CLAHE on CV_16UC1 (65536 bins), SURVEY 8f N4
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "common.hip.h"
#include "clahe.hip.h"

namespace mi {
// =============================================================================================
// CLAHE on CV_16UC1 (clahe.cpp CLAHE_CalcLut_Body<ushort,65536,0> / CLAHE_Interpolation_Body<ushort,0>).
// Not on the reference's path (OpenCV surface beyond it).  Everything is RANGE-ADAPTIVE: 16-bit video carries 10 or 12 bits,
// so the three kernels only ever touch the bins a frame populates.
//   tile_hist12   (vector geometry, the default) one workgroup per tile BETS on values < 4096: 4096 bins x 4 LDS copies, the whole
//                 LUT stage done from the counters in LDS; a tile that loses is redone by the careful sweeps of tile_hist16 in the
//                 same workgroup.  The last workgroup of a frame to arrive writes the frame's range and "every LUT written".
//   tile_hist16   one workgroup per tile; ONE pass builds the histogram of values < 32768 in 128 KiB of LDS and tracks the
//                 tile's min / max; a second pass runs only if the tile holds values >= 32768.  Only bins [lo, hi] are stored,
//                 with the tile's range next to them (unwritten bins are never read by anyone).
//   tile_lut16    leaves at once for a frame tile_hist12 has finished.  Otherwise: frame range = union of its tiles' ranges; the
//                 clip excess is summed over the tile's own bins; the scan walks [frame lo, frame hi] only, starting from the
//                 closed-form prefix of the empty bins below it (they still receive `batch` and their share of the residual
//                 increments, exactly as the sequential loops would give them).
//   clahe_interp16  one workgroup per (tile pair, band, sub-band), the pairs of a row on one XCD: stages {LUT[ty1][tx1][v],
//                 [ty1][tx2][v], [ty2][tx1][v], [ty2][tx2][v]} for v in the frame range as ONE LDS entry -- four floats when the
//                 range has at most 4096 values, else 8 bytes -- so a pixel costs one LDS read instead of four L2 gathers.
//                 kInterp16Entries 8-byte entries fit at a time (every 13-bit source in one go); a wider range
//                 is walked in windows of that size, each pixel finished in the window its value falls into (the workgroup's
//                 pixels are re-read once per window, from L2; windows none of them falls into are skipped -- a locally smooth
//                 image needs one or two of the eight).  Only the in-place RECTANGLES that need several windows still gather from
//                 L2 (clahe_interp16_wide_kernel): re-reading pixels that earlier windows have overwritten is not an option.
//   clahe_interp16_mid  (round 6) the SAME body with a 16384-entry table and 1024 threads, persistent, for the rectangles whose range
//                 needs 8193..16384 entries (every rectangle of a 14-bit frame: one window, the vector path) and for dense wider ones
//                 (half as many window passes); launched only while such content was seen lately (WideHint).
// =============================================================================================
constexpr int kHist16 = 65536;
constexpr int kHalf16 = 32768;
constexpr int kInterp16F32Entries = 4096;                  // ... or 4096 entries of four floats
constexpr int kInterp16Entries = 8192;               // LDS pair-table entries (64 KiB): two workgroups of 512 threads per CU
constexpr int kInterp16Threads = 512;
// ... and, since round 6, the same kernel body with a table TWICE that size for the rectangles whose range needs 8193..16384 entries
// (14-bit sensors: thermal, medical -- every rectangle of such a frame): ONE window of 128 KiB and the vector path instead of two
// windows of the small table and the scalar multi-window path (466 us -> see docs/experiments.md R6.3).  128 KiB of LDS is one
// workgroup of 1024 threads per CU (the same 16 waves per CU), persistent: clahe_interp16_mid_kernel.
constexpr int kInterp16MidEntries = 16384;
constexpr int kInterp16MidThreads = 1024;
// "A 14-bit rectangle was seen": what lets the HOST decide, without waiting for anything, whether a call should launch
// clahe_interp16_mid_kernel -- on narrower content that launch, 256 workgroups with 128 KiB of LDS that look at a few words and
// leave, costs ~8 us of a 12-bit call's 190.  A workgroup of clahe_interp16_kernel that meets such a rectangle stamps the call's
// sequence number into a device word, and the first one of the call to do so also into word 0 of two words of pinned host memory;
// tile_lut16_kernel, part of every call, stamps the number into host word 1 ("executed so far"); the host launches the mid kernel
// while the last stamped call lies at most eight EXECUTED calls back (counted in the device's progress, not the host's: a caller
// that enqueues twenty calls ahead must not see its own hint expire).  A hint: what a call launches never changes a byte of its result.
struct WideHint { uint32_t* dev; uint32_t* host; uint32_t seq; };
__device__ __forceinline__ void wide_seen(const WideHint& h)        // one lane
{
    if (!h.dev) return;
    // (look before exchanging: a thousand workgroups exchanging on one word cost a call 100 us; a thousand loads of it cost nothing)
    if (__hip_atomic_load(h.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == h.seq) return;
    if (__hip_atomic_exchange(h.dev, h.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != h.seq)
        __hip_atomic_store(h.host, h.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// which rectangles are the mid kernel's when it runs beside clahe_interp16_kernel (both ask, with the same numbers): span = highest
// value - lowest value rounded down to a multiple of four, in the frame's domain
// ... 8193..16384 entries: always (one window of the large table).  More: if the 4096-value stretches of the value range that the
// four tiles populate (presence bits of Range16.lo; the careful sweeps set them per window of four) have NO HOLE between the lowest
// and the highest -- dense wide content, where the large table halves the number of window passes (full range: 1695 -> 1319 us per
// 16 frames).  A 12-bit rectangle with a hot pixel (stretches 0..3 and 12..15) keeps the small table's windows, which skip the empty
// ones and do not wait for a persistent kernel's tail: taken by the mid kernel it lost 15 % (R6.3).
__device__ __forceinline__ bool rect_is_mid(uint32_t span, uint32_t presence)
{
    if (span < (uint32_t)kInterp16Entries) return false;
    if (span < (uint32_t)kInterp16MidEntries) return true;
    const uint32_t m = presence >> (presence ? __builtin_ctz(presence) : 0);     // lowest populated stretch at bit 0
    return m != 0u && (m & (m + 1u)) == 0u;                                        // ... and ones without a hole above it
}

// IN PLACE (src == dst) a rectangle that needs several windows cannot be done from a table (a window's pass would re-read pixels an earlier
// one has overwritten): 0 = the small table's (clahe_interp16_kernel), 1 = ONE window of the mid kernel's table, if it runs,
// 2 = clahe_interp16_wide_kernel gathers its pixels from the LUTs in L2.  All three kernels ask this with the same numbers, so a
// frame is shared out by rectangles: a 12-bit frame with a hot pixel leaves four of its 81 rectangles to the gathers, not all.
__device__ __forceinline__ int rect_owner_in_place(uint32_t span, int mid_runs)
{
    if (span < (uint32_t)kInterp16Entries) return 0;
    return (mid_runs && span < (uint32_t)kInterp16MidEntries) ? 1 : 2;
}

struct Range16 { uint32_t lo, hi; };                 // populated value range of a tile / frame (lo > hi: empty -- cannot happen, a tile has pixels)
// hi carries more than the bound: bits 0..15 the highest value, bits 16..19 a SHIFT -- every value of the tile (frame) is a multiple
// of 1 << shift (10- or 12-bit samples in the high bits of the word, as P010 / P016 video stores them) -- and bit 31 kLutDone.
// A frame with a shift runs in the COMPRESSED domain j = value >> shift from the LUT kernel on: its LUTs are stored at index j and
// the interpolation looks pixels up at (pixel >> shift), so MSB-aligned 10- and 12-bit content needs one 1024- / 4096-entry table
// like LSB-aligned content instead of windows over the whole 16-bit range.
// lo of a TILE carries, above its 16 bits, a presence mask: bit c set if the tile may hold a value in [4096 c, 4096 c + 4095] (conservative).
// tile_lut16_kernel writes a LUT only over the 4096-value stretches some tile of the 3 x 3 neighbourhood populates.
__device__ __forceinline__ uint32_t range_lo(uint32_t lo) { return lo & 0xffffu; }
__device__ __forceinline__ uint32_t range_mask(uint32_t lo) { return lo >> 16; }
__device__ __forceinline__ uint32_t chunk_bits(uint32_t vlo, uint32_t vhi)      // presence bits of the 4096-value stretches [vlo >> 12, vhi >> 12]
{
    const uint32_t a = vlo >> 12, b = min(vhi, 0xffffu) >> 12;
    return ((2u << b) - 1u) & ~((1u << a) - 1u);
}
constexpr uint32_t kRangeHiMask = 0xffffu;
__device__ __forceinline__ uint32_t range_hi(uint32_t hi) { return hi & kRangeHiMask; }
__device__ __forceinline__ uint32_t range_shift(uint32_t hi) { return (hi >> 16) & 15u; }
// shift of a set of values from the OR of all of them (packed pairs allowed): trailing zero bits, 15 when every value is 0
__device__ __forceinline__ uint32_t shift_of_or(uint32_t packed_or)
{
    const uint32_t o = (packed_or | (packed_or >> 16)) & 0xffffu;
    return o ? (uint32_t)__builtin_ctz(o) : 15u;
}

__device__ __forceinline__ void lds_add(uint32_t* h, uint32_t idx, uint32_t n)
{
    __hip_atomic_fetch_add(h + idx, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32
}

template <int kWinBits>
__device__ __forceinline__ void hist16_add_dword(uint32_t* h16, uint32_t w, int half, uint32_t& lmin, uint32_t& lmax, uint32_t& lor)
{
    const uint32_t a = w & 0xffffu, b = w >> 16;
    lmin = min(lmin, min(a, b)); lmax = max(lmax, max(a, b)); lor |= w;
    if (a == b) { if ((int)(a   >> kWinBits) == half) lds_add(h16, a & ((1u << kWinBits) - 1), 2u); return; }
    if ((int)(a   >> kWinBits) == half) lds_inc(h16, a & ((1u << kWinBits) - 1));
    if ((int)(b   >> kWinBits) == half) lds_inc(h16, b & ((1u << kWinBits) - 1));
}

// Eight pixels of one 16-byte load.  There is no room to replicate 32 768 counters per LDS bank, so equal values meeting in one
// ds_add serialise; flat image regions (borders, saturated areas) are the bad case and are caught before they reach the LDS:
// all eight pixels equal -> one add of 8; the same value in every active lane of the wave -> one lane adds for the whole wave.
template <int kWinBits>
__device__ __forceinline__ void hist16_add_vec(uint32_t* h16, const u32x4& q, int half, uint32_t& lmin, uint32_t& lmax, uint32_t& lor)
{
    const uint32_t v0 = q.x & 0xffffu;
    const bool flat = q.x == q.y && q.y == q.z && q.z == q.w && v0 == (q.x >> 16);
    if (__builtin_expect(flat, 0)) {
        lmin = min(lmin, v0); lmax = max(lmax, v0); lor |= v0;
        const unsigned long long active = __ballot(1);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0);
        if (__ballot(v0 == first) == active) {                      // wave-uniform value (only lanes with flat vectors are here)
            if ((int)(v0   >> kWinBits) == half && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(active))
                lds_add(h16, v0 & ((1u << kWinBits) - 1), 8u * (uint32_t)__builtin_popcountll(active));
        } else if ((int)(v0   >> kWinBits) == half) {
            lds_add(h16, v0 & ((1u << kWinBits) - 1), 8u);
        }
        return;
    }
    hist16_add_dword<kWinBits>(h16, q.x, half, lmin, lmax, lor); hist16_add_dword<kWinBits>(h16, q.y, half, lmin, lmax, lor);
    hist16_add_dword<kWinBits>(h16, q.z, half, lmin, lmax, lor); hist16_add_dword<kWinBits>(h16, q.w, half, lmin, lmax, lor);
}

// OPTIMISTIC sweep (vector path): count the pixels of window 0 (values below 1 << kWinBits) with one test per PAIR of pixels -- no
// bit above the window in either half -- and track the range with packed 16-bit min / max (one v_pk_min_u16 / v_pk_max_u16 per two
// pixels).  For everything that fits the window this sweep is the whole job at ~5 VALU instructions per pixel instead of ~14.
// A value beyond the window is not counted but its window is noted (wmask): the counters stay exact for window 0, and only the
// windows that hold something are swept afterwards, one sweep each -- a 12-bit tile with one hot pixel costs two sweeps, not five.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int kWinBits>
__device__ __forceinline__ void hist16_fast_dword(uint32_t* h16, uint32_t w, u16x2& pmin, u16x2& pmax, uint32_t& lor, uint32_t& wmask)
{
    const u16x2 v = __builtin_bit_cast(u16x2, w);
    pmin = __builtin_elementwise_min(pmin, v); pmax = __builtin_elementwise_max(pmax, v); lor |= w;
    constexpr uint32_t kHigh = ((0xffffu << kWinBits) & 0xffffu) * 0x10001u;          // the bits a value of window 0 does not have
    if (__builtin_expect((w & kHigh) == 0u, 1)) {
        lds_inc(h16, w & ((1u << kWinBits) - 1));
        lds_inc(h16, w >> 16);
        return;
    }
    // a value beyond window 0 is NOT counted (the counters stay exact for window 0); its window is noted for a sweep of its own
    const uint32_t a = w & 0xffffu, b = w >> 16;
    if (a < (1u << kWinBits)) lds_inc(h16, a); else wmask |= 1u << (a >> kWinBits);
    if (b < (1u << kWinBits)) lds_inc(h16, b); else wmask |= 1u << (b >> kWinBits);
}
template <int kWinBits>
__device__ __forceinline__ void hist16_fast_vec(uint32_t* h16, const u32x4& q, u16x2& pmin, u16x2& pmax, uint32_t& lor, uint32_t& wmask)
{
    const uint32_t v0 = q.x & 0xffffu;
    const bool flat = q.x == q.y && q.y == q.z && q.z == q.w && v0 == (q.x >> 16);
    if (__builtin_expect(flat, 0)) {                                // as hist16_add_vec: flat regions never reach the LDS pixel by pixel
        const u16x2 v = __builtin_bit_cast(u16x2, q.x);
        pmin = __builtin_elementwise_min(pmin, v); pmax = __builtin_elementwise_max(pmax, v); lor |= q.x;
        if (v0 >= (1u << kWinBits)) { wmask |= 1u << (v0 >> kWinBits); return; }     // beyond window 0: noted, not counted
        const unsigned long long active = __ballot(1);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0);
        if (__ballot(v0 == first) == active) {
            if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(active)) lds_add(h16, v0, 8u * (uint32_t)__builtin_popcountll(active));
        } else {
            lds_add(h16, v0, 8u);
        }
        return;
    }
    hist16_fast_dword<kWinBits>(h16, q.x, pmin, pmax, lor, wmask); hist16_fast_dword<kWinBits>(h16, q.y, pmin, pmax, lor, wmask);
    hist16_fast_dword<kWinBits>(h16, q.z, pmin, pmax, lor, wmask); hist16_fast_dword<kWinBits>(h16, q.w, pmin, pmax, lor, wmask);
}

// grid = (tiles, frames), NT threads (1024 in tile_hist16_kernel), (4 << kWinBits) bytes of dynamic LDS.  steps in BYTES.
// `vec` (host: no REFLECT_101 padding, tile_w % 8 == 0, 16-B aligned rows): a lane takes 8 pixels per 16-byte load with
// four loads in flight; otherwise one pixel per lane per step with index reflection.
template <int kWinBits, int NT = 1024>
__device__ __forceinline__ void tile_hist16_careful(uint32_t* h16 /* [1 << kWinBits] LDS */, uint32_t& s_lo, uint32_t& s_hi, uint32_t& s_or,
                                                    const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                    const ClaheGeom& g, uint32_t* __restrict__ hist, Range16* __restrict__ ranges, int vec)
{
    constexpr int kWin = 1 << kWinBits;
    const int t = threadIdx.x;
    const int tile = blockIdx.x, f = blockIdx.y;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    const size_t tile_id = (size_t)f * gridDim.x + tile;
    uint32_t* out = hist + tile_id * kHist16;
    const long long items = (long long)g.tile_h * g.tile_w;
    const int drow = NT / g.tile_w, dcol = NT - drow * g.tile_w;
    const int slots = g.tile_w >> 3;                              // 8-pixel groups per tile row (vector path)
    const int vitems = g.tile_h * slots;
    const uint8_t* tbase = src + (long long)ty * g.tile_h * step + (long long)tx * g.tile_w * 2;
    auto vload = [&](int it) -> u32x4 {
        const int row = it / slots, slot = it - row * slots;
        return *reinterpret_cast<const u32x4*>(tbase + (long long)row * step + (slot << 4));
    };
    uint32_t lmin = 0xffffu, lmax = 0, lor = 0;
    auto vadd = [&](const u32x4& q, int half) { hist16_add_vec<kWinBits>(h16, q, half, lmin, lmax, lor); };
    __shared__ uint32_t s_wm;                                     // which windows beyond the first hold a value (vector path)
    if (t == 0) { s_lo = 0xffffu; s_hi = 0; s_or = 0; s_wm = 0; }
    uint32_t lo = 0, hi = 0;
    if (vec) {
        // ---- optimistic sweep: window 0 counted, the others noted (see hist16_fast_vec)
        for (int i = t; i < kWin / 4; i += NT) reinterpret_cast<u32x4*>(h16)[i] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        u16x2 pmin = {0xffff, 0xffff}, pmax = {0, 0};
        uint32_t wmask = 0;
        int row = t / slots, slot = t - row * slots;
        const int vdrow = NT / slots, vdslot = NT - vdrow * slots;
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int it = t; it < vitems; it += 4 * NT) {           // (row, slot) items walked incrementally, four predicated loads in flight
            u32x4 q[4]; bool qv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qv[k] = it + k * NT < vitems;
                const u32x4* ptr = reinterpret_cast<const u32x4*>(tbase + (long long)row * step + (slot << 4));
                q[k] = qv[k] ? *ptr : zero;
                row += vdrow; slot += vdslot;
                if (slot >= slots) { slot -= slots; ++row; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (qv[k]) hist16_fast_vec<kWinBits>(h16, q[k], pmin, pmax, lor, wmask);
        }
        lmin = min((uint32_t)pmin.x, (uint32_t)pmin.y); lmax = max((uint32_t)pmax.x, (uint32_t)pmax.y);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lmin = min(lmin, (uint32_t)__shfl_xor((int)lmin, d, 64)); lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
            lor |= (uint32_t)__shfl_xor((int)lor, d, 64); wmask |= (uint32_t)__shfl_xor((int)wmask, d, 64);
        }
        if ((t & 63) == 0) {
            __hip_atomic_fetch_min(&s_lo, lmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&s_hi, lmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_or(&s_or, lor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wmask) __hip_atomic_fetch_or(&s_wm, wmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        lo = s_lo; hi = s_hi;
        const uint32_t wm = s_wm;
        // window 0: the counters are its histogram (bins are stored from a 4-aligned start: the LUT kernel loads 16 bytes)
        if (lo < (uint32_t)kWin)
            for (uint32_t i = (lo & ~3u) + (uint32_t)t; i <= min(hi, (uint32_t)kWin - 1u); i += NT) out[i] = h16[i];
        // the windows above it: one sweep for each that holds something, zeros for the stretches of [lo, hi] in the others
        for (int half = 1; half < (kHist16 >> kWinBits); ++half) {
            const uint32_t base = (uint32_t)half * (uint32_t)kWin;
            if (hi < base) break;
            const uint32_t b0 = max(lo, base), b1 = min(hi, base + (uint32_t)kWin - 1);
            if (b0 > b1) continue;                                  // (the range starts above this window)
            if (!((wm >> half) & 1u)) {
                for (uint32_t i = (b0 & ~3u) + (uint32_t)t; i <= b1; i += NT) out[i] = 0u;
                continue;
            }
            __syncthreads();                                        // the previous window's counters have been stored
            for (int i = t; i < kWin / 4; i += NT) reinterpret_cast<u32x4*>(h16)[i] = zero;
            __syncthreads();
            int it = t;
            for (; it + 3 * NT < vitems; it += 4 * NT) {
                const u32x4 a = vload(it), b = vload(it + NT), c = vload(it + 2 * NT), d = vload(it + 3 * NT);
                vadd(a, half); vadd(b, half); vadd(c, half); vadd(d, half);
            }
            for (; it < vitems; it += NT) vadd(vload(it), half);
            __syncthreads();
            for (uint32_t i = (b0 & ~3u) + (uint32_t)t; i <= b1; i += NT) out[i] = h16[i & ((1u << kWinBits) - 1)];
        }
        if (t == 0) {
            constexpr uint32_t kPer = (uint32_t)kWin >> 12;             // 4096-value stretches per window
            uint32_t m = lo < (uint32_t)kWin ? (1u << kPer) - 1u : 0u;
            for (uint32_t w = 1; w < (uint32_t)(kHist16 >> kWinBits); ++w) if ((wm >> w) & 1u) m |= ((1u << kPer) - 1u) << (w * kPer);
            Range16 r; r.lo = lo | ((m & chunk_bits(lo, hi)) << 16); r.hi = hi | (shift_of_or(s_or) << 16); ranges[tile_id] = r;
        }
        return;
    }
    // ---- scalar path (REFLECT_101 padding, odd tile widths): one window of the value range per sweep
    for (int half = 0; half < (kHist16 >> kWinBits); ++half) {
        for (int i = t; i < kWin; i += NT) h16[i] = 0;
        __syncthreads();
        {
            int row = t / g.tile_w, col = t - row * g.tile_w;
            for (long long it = t; it < items; it += NT) {
                const int y = reflect101(ty * g.tile_h + row, g.height);
                const int x = reflect101(tx * g.tile_w + col, g.width);
                const uint32_t v = *reinterpret_cast<const uint16_t*>(src + (long long)y * step + 2 * (long long)x);
                lmin = min(lmin, v); lmax = max(lmax, v); lor |= v;
                if ((int)(v   >> kWinBits) == half) lds_inc(h16, v & ((1u << kWinBits) - 1));
                row += drow; col += dcol;
                if (col >= g.tile_w) { col -= g.tile_w; ++row; }
            }
        }
        if (half == 0) {                                          // the tile's range is known after the first sweep
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                lmin = min(lmin, (uint32_t)__shfl_xor((int)lmin, d, 64)); lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
                lor |= (uint32_t)__shfl_xor((int)lor, d, 64);
            }
            if ((t & 63) == 0) {
                __hip_atomic_fetch_min(&s_lo, lmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&s_hi, lmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(&s_or, lor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        lo = s_lo; hi = s_hi;
        // store the populated bins of this window only: [max(lo, base), min(hi, base + kWin - 1)]
        const uint32_t base = (uint32_t)half * (uint32_t)kWin;
        const uint32_t b0 = max(lo, base), b1 = min(hi, base + (uint32_t)kWin - 1);
        if (b0 <= b1)
            for (uint32_t i = (b0 & ~3u) + (uint32_t)t; i <= b1; i += NT) out[i] = h16[i & ((1u << kWinBits) - 1)];   // from a 4-aligned start: the LUT kernel loads 16 B
        if (hi < base + (uint32_t)kWin) break;                    // nothing above this window: done
        __syncthreads();
    }
    if (t == 0) { Range16 r; r.lo = lo | (chunk_bits(lo, hi) << 16); r.hi = hi | (shift_of_or(s_or) << 16); ranges[tile_id] = r; }
}

__global__ __launch_bounds__(1024) void tile_hist16_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                          ClaheGeom g, uint32_t* __restrict__ hist, Range16* __restrict__ ranges, int vec)
{
    extern __shared__ uint32_t h16[];                            // [32768]
    __shared__ uint32_t s_lo, s_hi, s_or;
    tile_hist16_careful<15>(h16, s_lo, s_hi, s_or, src_base, step, frame_stride, g, hist, ranges, vec);
}

// ---- 12-bit fast path -----------------------------------------------------------------------------------------------------
// What 16-bit video actually carries is 10 or 12 bits.  tile_hist12_kernel BETS on that: 4096 bins x COPIES copies (32 KiB of LDS with
// two: four workgroups of 512 threads per CU), a lane counts at (value & 4095) * COPIES + (lane & (COPIES - 1)), and the packed min / max it
// tracks anyway tells at the end whether the bet held (max < 4096).  Against the 32 768-counter sweep above:
//   * the sweep is bound by the loads a CU keeps in flight (a constant frame takes as long as noise): two workgroups per CU double
//     them -- the 32 768-counter kernel fills the LDS with one;
//   * equal values meeting in one ds_add serialise, and neighbouring pixels of a real image ARE equal or close: with four copies
//     at most the 16 lanes that share a copy can meet, not 64;
//   * the loads of the next four vectors are in flight while the current four are counted (two register sets): with one
//     workgroup of 16 waves per CU the ~2 us of HBM latency per iteration was otherwise exposed;
//   * the whole LUT stage is done right here, from the counters in LDS (clip, redistribute, prefix sum over bins 0..4095 -- the same
//     arithmetic as tile_lut16_kernel, which then returns at once for such a tile as long as the WHOLE frame stayed below 4096).
// A tile that loses the bet -- some value >= 4096, noticed after the first four vectors per lane or at the end -- is redone by
// tile_hist16_careful (as many counters per sweep as fit the same LDS: 8192 with two copies, up to eight sweeps) in the same workgroup.  ranges[tile].hi carries bit 31 when the tile's LUT was written here.
constexpr uint32_t kLutDone = 0x80000000u;
constexpr uint32_t kHistCompressed = 0x40000000u;   // Range16.hi of a tile: its histogram is stored at index value >> shift (tile_hist12_kernel with a shift)
constexpr int kBins12 = 4096;
// Shipped shape: 1024 threads, 4 copies = 64 KiB of LDS, two workgroups per CU.  512 threads x 2 copies (32 KiB, four workgroups per
// CU) measured the same on 12-bit content (16 4K frames: 59.6 us alone either way; the sweep on its own 45 us either way,
// tools/hist12_probe.hip) and leaves the careful path half the counters per sweep, so the larger shape stays.
constexpr int kHist12Threads = 1024;
constexpr int kCopies12 = 4;
constexpr int kHist12Words = kBins12 * kCopies12;

// sft / wl are wave-uniform: a pixel is counted at bits sft .. sft + wl - 1 of its value (wl = min(12, 16 - sft): v_bfe_u32 on the packed
// pair must not reach into the neighbour's bits).  por collects the OR of everything seen: it tells at the end whether every value
// really was a multiple of 1 << sft AND below 4096 << sft (some value has a bit set iff the OR has it), so the sweep tracks no
// minimum / maximum -- the tile's range is read off its histogram afterwards.
template <int COPIES>
__device__ __forceinline__ void hist12_dword(uint32_t* h, uint32_t w, uint32_t cp, uint32_t sft, uint32_t wl, uint32_t& por)
{
    por |= w;
    lds_inc(h, (__builtin_amdgcn_ubfe(w, sft, wl) * COPIES) | cp);
    lds_inc(h, (__builtin_amdgcn_ubfe(w, sft + 16u, wl) * COPIES) | cp);
}
template <int COPIES>
__device__ __forceinline__ void hist12_vec(uint32_t* h, const u32x4& q, uint32_t cp, uint32_t sft, uint32_t wl, uint32_t& por)
{
    const uint32_t v0 = q.x & 0xffffu;
    const bool flat = q.x == q.y && q.y == q.z && q.z == q.w && v0 == (q.x >> 16);
    if (__builtin_expect(flat, 0)) {                                // flat regions never reach the LDS pixel by pixel (as hist16_fast_vec)
        por |= q.x;
        const unsigned long long active = __ballot(1);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0);
        const uint32_t idx = (__builtin_amdgcn_ubfe(v0, sft, wl) * COPIES) | cp;
        if (__ballot(v0 == first) == active) {
            if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(active)) lds_add(h, idx, 8u * (uint32_t)__builtin_popcountll(active));
        } else {
            lds_add(h, idx, 8u);
        }
        return;
    }
    hist12_dword<COPIES>(h, q.x, cp, sft, wl, por); hist12_dword<COPIES>(h, q.y, cp, sft, wl, por);
    hist12_dword<COPIES>(h, q.z, cp, sft, wl, por); hist12_dword<COPIES>(h, q.w, cp, sft, wl, por);
}

// The context's shift hint ("which shift did the previous call's frames settle on"): TWO words.  Word 0 is only READ while the tile
// kernels of a call run -- every tile of every frame of a launch sees one value, so tiles of a frame cannot disagree because they
// read at different times -- word 1 collects what this call's frames settle on (any of them: a stream keeps its format), and the
// interpolation kernel, which runs when all of that is over, copies word 1 to word 0 for the next call.  Relaxed agent-scope
// accesses: the words live across launches and XCDs.  Speed only: bytes never depend on the hint (see tile_hist12_kernel).
__device__ __forceinline__ uint32_t hint_in(const uint32_t* h) { return __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hint_out(uint32_t* h, uint32_t v) { __hip_atomic_store(h + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hint_roll(uint32_t* h)
{
    if (h && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(h, __hip_atomic_load(h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid = (tiles, frames), NT threads, 4096 * COPIES * 4 bytes of dynamic LDS; vector geometry only (the host checks: no padding,
// tile_w % 8 == 0, 16-B aligned rows).  A thread owns 4096 / NT consecutive bins in the LUT stage.
template <int NT, int COPIES>
__global__ __launch_bounds__(NT, 8) void tile_hist12_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                        ClaheGeom g, uint32_t* __restrict__ hist, Range16* __restrict__ ranges,
                                                        float lut_scale16, int clip16, uint16_t* __restrict__ luts,
                                                        uint32_t* __restrict__ sync, Range16* __restrict__ frame_ranges, uint32_t* __restrict__ frame_done,
                                                        uint32_t* __restrict__ shift_hint)
{
    static_assert(COPIES == 2 || COPIES == 4, "copies");
    constexpr int NW = NT / 64, BPT = kBins12 / NT;                // waves; bins per thread
    static_assert(BPT % 4 == 0 && BPT >= 4, "a thread owns whole 16-byte groups of bins");
    constexpr int kCarefulBits = COPIES == 4 ? 14 : 13;            // the careful path's counters fill the same LDS
    extern __shared__ uint32_t h16[];                            // [4096][COPIES], or [1 << kCarefulBits] for the careful path
    __shared__ uint32_t s_lo, s_hi, s_sum, s_or;
    __shared__ uint32_t s_w[NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int tile = blockIdx.x, f = blockIdx.y;
    const size_t tile_id = (size_t)f * gridDim.x + tile;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const uint8_t* src = src_base + (long long)f * frame_stride;
    const int slots = g.tile_w >> 3;
    const int vitems = g.tile_h * slots;
    const uint8_t* tbase = src + (long long)ty * g.tile_h * step + (long long)tx * g.tile_w * 2;
    const uint32_t cp = (uint32_t)t & (uint32_t)(COPIES - 1);
    if (t == 0) { s_lo = 0xffffu; s_hi = 0; s_sum = 0; s_or = 0; }
    for (int i = t; i < kBins12 * COPIES / 4; i += NT) reinterpret_cast<u32x4*>(h16)[i] = u32x4{0u, 0u, 0u, 0u};
    // (row, slot) items walked incrementally, four predicated loads per set
    int row = t / slots, slot = t - row * slots;
    const int vdrow = NT / slots, vdslot = NT - vdrow * slots;
    const u32x4 zero = {0u, 0u, 0u, 0u};
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
    // The first set -- and, in wave 0, one vector per lane from 64 places spread over the tile (a letterbox bar at the top says nothing
    // about the picture below it) -- decides the bet: sft = the trailing zero bits their OR has (0 for ordinary content; 6 for P010,
    // 4 for MSB-aligned 12-bit), and the bet is that every value of the tile is a multiple of 1 << sft below 4096 << sft.  A frame that
    // cannot be so shows it here (almost) always.  Whatever is assumed here is verified at the end on all pixels.
    uint32_t m0 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) m0 |= cur[k].x | cur[k].y | cur[k].z | cur[k].w;      // OR of the packed pixels
    if (wv == 0) {
        const int it_s = (int)(((long long)lane * vitems) >> 6);
        const int row_s = it_s / slots, slot_s = it_s - row_s * slots;
        const u32x4 qs = *reinterpret_cast<const u32x4*>(tbase + (long long)row_s * step + (slot_s << 4));
        m0 |= qs.x | qs.y | qs.z | qs.w;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m0 |= (uint32_t)__shfl_xor((int)m0, d, 64);
    if (lane == 0) s_w[wv] = m0;                                    // one slot per wave: nothing to initialise, one barrier
    __syncthreads();                                                // (also orders the zeroing above)
    uint32_t o32 = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) o32 |= s_w[k];
    const uint32_t o16 = (o32 | (o32 >> 16)) & 0xffffu;
    // No shift while everything seen is below 4096: a flat tile of an ordinary 12-bit frame (a letterbox bar at black level 256) must
    // not pick a shift of its own -- the frame is only "done" when all its tiles used one.
    // With values >= 4096 every shift from bitlen(OR) - 12 up to the OR's trailing zeros would do, and the results do not depend on the
    // choice -- only whether the frame's tiles agree does.  The letterbox bars of a P010 frame (black = 64 << 6: trailing zeros 12) and
    // its picture (6) agree if the bars take the shift the context's previous frame ran with (*shift_hint, written when a frame is
    // settled): a video stream keeps its format.
    uint32_t sft_v = 0u;                                             // uniform (made scalar below)
    if (o16 >= (uint32_t)kBins12) {
        const uint32_t smax = (uint32_t)__builtin_ctz(o16), smin = 20u - (uint32_t)__builtin_clz(o16);     // bitlen(o16) - 12
        const uint32_t hint = hint_in(shift_hint);
        sft_v = (hint >= smin && hint <= smax) ? hint : smax;
    } else if (o16 == 0u) {
        sft_v = min(hint_in(shift_hint), 15u);                      // nothing but zeros seen: any shift will do, so go along with the frame before
    }
    const uint32_t sft = (uint32_t)__builtin_amdgcn_readfirstlane((int)sft_v);       // in SGPRs: the sweep has no VGPR to spare
    const uint32_t wl = min(12u, 16u - sft);
    bool lost = (o16 >> sft) >= (uint32_t)kBins12;                  // uniform
    if (!lost) {
        uint32_t por = 0;
        for (int it = t; it < vitems; it += 4 * NT) {
            const bool more = it + 4 * NT < vitems;                  // uniform per lane only; the loads are predicated anyway
            if (more) load_set(it + 4 * NT, nxt, nv);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (cv[k]) hist12_vec<COPIES>(h16, cur[k], cp, sft, wl, por);
            if (more) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { cur[k] = nxt[k]; cv[k] = nv[k]; }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) por |= (uint32_t)__shfl_xor((int)por, d, 64);
        if (lane == 0 && por) __hip_atomic_fetch_or(&s_or, por, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        // the bet held if every value is below 4096 << sft and (with a shift) a multiple of 1 << sft
        const uint32_t all16 = (s_or | (s_or >> 16)) & 0xffffu;
        lost = (all16 >> sft) >= (uint32_t)kBins12 || (sft != 0 && (all16 & ((1u << sft) - 1u)) != 0u);
    }
    // The frame's range and "every tile wrote its LUT here" are settled by the LAST workgroup of the frame to arrive, so that
    // tile_lut16_kernel can leave at once on one scalar load (launched only to leave, it still cost 26 us per 16 frames in the sequence:
    // every workgroup re-derived the frame's range from the tiles' ranges first).  ONE 64-bit word per frame, zero between launches:
    // bits 0..15 arrivals, 16..31 tiles whose bet held, 32..47 which 256-bin buckets hold a tile's lowest / highest bin, 48..63 which
    // shifts the tiles bet on.  The frame is done if every tile's bet held AND they all used one shift (their LUTs then share a
    // domain); a letterboxed P010 frame, whose bars have another shift than its picture, goes through tile_lut16_kernel.  A tile
    // ORs its bits in and then adds its arrival -- two relaxed agent-scope atomics on the SAME address, so every arrival
    // the last workgroup sees comes with its bits (performed at the L2: no write-back / invalidate of this XCD's L2, see
    // hist_lut_kernel).  Thread 0 issues them as soon as the tile knows its range (or that its bet is lost), and looks at the returned
    // value only at the very end: waiting for it held the whole workgroup at its next barrier (9 us per 16 frames), arriving after
    // the LUT made thread 0 sit on its own stores.  The frame range so reported is rounded out to buckets (the interpolation sizes
    // its tables by the tiles' own exact ranges).  A tile that lost its bet adds an arrival and nothing else: the frame is then not
    // "done" and tile_lut16_kernel derives the exact range itself.
    unsigned long long before64 = 0;
    uint32_t own_bits = 0, own_add = 1u;
    unsigned long long* const sy = reinterpret_cast<unsigned long long*>(sync) + 2 * (size_t)f;
    auto arrive = [&](bool done, uint32_t jlo, uint32_t jhi) {      // thread 0; jlo / jhi: lowest / highest populated bin
        if (done) {
            own_bits = (1u << (jlo >> 8)) | (1u << (jhi >> 8)) | (0x10000u << sft);
            own_add = 0x10001u;
            __hip_atomic_fetch_or(sy, (unsigned long long)own_bits << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        before64 = __hip_atomic_fetch_add(sy, (unsigned long long)own_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto settle_frame = [&]() {                                     // thread 0, last statement of either path
        if ((uint32_t)(before64 & 0xffffu) != gridDim.x - 1) return;
        const uint32_t nd = (uint32_t)((before64 >> 16) & 0xffffu) + (own_add >> 16);
        const uint32_t bits = (uint32_t)(before64 >> 32) | own_bits;
        __hip_atomic_store(sy, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t buckets = bits & 0xffffu, shifts = bits >> 16;
        const bool done = nd == gridDim.x && __builtin_popcount(shifts) == 1;
        Range16 r; r.lo = 0xffffu; r.hi = 0;
        if (done) {
            const uint32_t fs = (uint32_t)__builtin_ctz(shifts);
            r.lo = ((uint32_t)__builtin_ctz(buckets) << 8) << fs;
            r.hi = min(((((31u - (uint32_t)__builtin_clz(buckets)) << 8) | 255u) << fs), 0xffffu) | (fs << 16);
        }
        frame_ranges[f] = r;
        frame_done[f] = done ? 1u : 0u;
        if (done) hint_out(shift_hint, (uint32_t)__builtin_ctz(shifts));
    };
    if (lost) {                                                   // uniform over the workgroup: redo the tile the careful way
        __syncthreads();
        tile_hist16_careful<kCarefulBits, NT>(h16, s_lo, s_hi, s_or, src_base, step, frame_stride, g, hist, ranges, 1);   // its counters fill the same LDS
        if (t == 0) { arrive(false, 0u, 0u); settle_frame(); }      // (after the sweeps: nothing of the arrival is live across them)
        return;
    }
    // ---- the tile's 4096 counts: thread t owns bins BPT * t .. BPT * t + BPT - 1 (sum of the copies)
    // Read out conflict-free: consecutive lanes read the consecutive copies of consecutive bins, the 4096 sums are compacted to the head
    // of the LDS array, and each thread then picks up its consecutive bins.
    int v[BPT];
    {
        uint32_t part[BPT];
#pragma unroll
        for (int j = 0; j < BPT; ++j) {                              // bin t + NT * j
            if (COPIES == 4) { const u32x4 a = reinterpret_cast<const u32x4*>(h16)[t + NT * j]; part[j] = a.x + a.y + a.z + a.w; }
            else { const uint2 a = reinterpret_cast<const uint2*>(h16)[t + NT * j]; part[j] = a.x + a.y; }
        }
        __syncthreads();                                            // every copy has been read: the head of the array may be overwritten
#pragma unroll
        for (int j = 0; j < BPT; ++j) h16[t + NT * j] = part[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BPT / 4; ++j) {
            const u32x4 q = reinterpret_cast<const u32x4*>(h16)[t * (BPT / 4) + j];
            v[4 * j] = (int)q.x; v[4 * j + 1] = (int)q.y; v[4 * j + 2] = (int)q.z; v[4 * j + 3] = (int)q.w;
        }
    }
    const uint32_t b0 = (uint32_t)t * BPT;
    // the tile's range, read off the histogram: lowest / highest populated bin (exact: every value is a multiple of 1 << sft)
    {
        uint32_t first = 0xffffu, last = 0u;
#pragma unroll
        for (int k = BPT - 1; k >= 0; --k) if (v[k]) first = b0 + (uint32_t)k;
#pragma unroll
        for (int k = 0; k < BPT; ++k) if (v[k]) last = b0 + (uint32_t)k;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { first = min(first, (uint32_t)__shfl_xor((int)first, d, 64)); last = max(last, (uint32_t)__shfl_xor((int)last, d, 64)); }
        if (lane == 0) {
            __hip_atomic_fetch_min(&s_lo, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&s_hi, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
    }
    const uint32_t lo = s_lo << sft, hi = s_hi << sft;
    if (t == 0) arrive(true, s_lo, s_hi);
    // the histogram itself (bin j = count of value j << sft), for tile_lut16_kernel: always needed with a shift, and without one should
    // the FRAME turn out wider than 4096 values (another tile lost its bet)
    {
        const uint32_t lo_c = lo >> sft, hi_c = hi >> sft;
#pragma unroll
        for (int j = 0; j < BPT / 4; ++j) {
            const uint32_t bj = b0 + 4 * j;
            if (bj + 3 >= (lo_c & ~3u) && bj <= hi_c)
                *reinterpret_cast<u32x4*>(hist + tile_id * kHist16 + bj) = u32x4{(uint32_t)v[4 * j], (uint32_t)v[4 * j + 1], (uint32_t)v[4 * j + 2], (uint32_t)v[4 * j + 3]};
        }
    }
    auto block_scan = [&](uint32_t x, uint32_t& total) -> uint32_t {  // inclusive prefix of x over the NT threads
        const uint32_t incl = wave_incl_scan(x);
        __syncthreads();
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int k = 0; k < NW; ++k) { const uint32_t y = s_w[k]; if (k < wv) off += y; tot += y; }
        total = tot;
        return off + incl;
    };
    // ---- clip, redistribute, prefix sum, scale: clahe.cpp for histSize 65536, exactly as tile_lut16_kernel does it, in this tile's
    // own domain (bin j stands for value j << sft; the bins in between in closed form).  The LUT is stored at index j: it is THE
    // LUT if the frame's shift turns out to be this tile's, else tile_lut16_kernel writes it again
    int batch = 0, residual = 0, rstep = 1;
    if (clip16 > 0) {
        uint32_t excess = 0;
#pragma unroll
        for (int k = 0; k < BPT; ++k) if (v[k] > clip16) excess += (uint32_t)(v[k] - clip16);
        // a sum, not a scan: one LDS add per wave and one barrier
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) excess += (uint32_t)__shfl_xor((int)excess, d, 64);
        if (lane == 0 && excess) __hip_atomic_fetch_add(&s_sum, excess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const uint32_t clipped = s_sum;
        batch = (int)clipped / kHist16;
        residual = (int)clipped - batch * kHist16;
        if (residual != 0) { rstep = kHist16 / residual; if (rstep < 1) rstep = 1; }
    }
    uint32_t local = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        if (clip16 > 0 && v[k] > clip16) v[k] = clip16;
        local += (uint32_t)v[k];
        v[k] = (int)local;                                           // inclusive prefix of the clipped counts within the thread's bins
    }
    uint32_t total;
    const uint32_t before = block_scan(local, total) - local;
    uint32_t packed[BPT / 2];
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        uint32_t sum = before + (uint32_t)v[k];
        if (clip16 > 0) {
            const uint32_t b = (b0 + (uint32_t)k) << sft;            // the value this bin stands for
            sum += (uint32_t)batch * (b + 1u);
            if (residual != 0) sum += min((uint32_t)residual, b / (uint32_t)rstep + 1u);
        }
        int r = __float2int_rn(__fmul_rn((float)(int)sum, lut_scale16));
        r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
        if (k & 1) packed[k >> 1] |= (uint32_t)r << 16; else packed[k >> 1] = (uint32_t)r;
    }
#pragma unroll
    for (int j = 0; j < BPT / 4; ++j)
        *reinterpret_cast<uint2*>(luts + tile_id * kHist16 + b0 + 4 * j) = make_uint2(packed[2 * j], packed[2 * j + 1]);
    if (t == 0) { Range16 r; r.lo = lo | (chunk_bits(lo, hi) << 16); r.hi = hi | (sft << 16) | (sft ? kHistCompressed : 0u) | kLutDone; ranges[tile_id] = r; settle_frame(); }
}

// grid = (tiles, frames), 1024 threads.  Works in the frame's COMPRESSED domain j = value >> shift (shift = the smallest of its tiles'
// shifts; 0 for ordinary content, where j is the value itself): bins are walked in chunks of 4096, four consecutive j per thread, over
// the range of the tile and its eight neighbours only (nobody else's pixels read this LUT), and the LUT is stored at index j.  Semantics of clahe.cpp for histSize 65536: clip at
// clip16, excess / 65536 added to every one of the 65 536 bins, the residual spread with stride max(65536 / residual, 1); then the
// prefix sum scaled by lut_scale16.  Bins that cannot be populated (below the range, between multiples of 1 << shift) still receive
// `batch` and their share of the residual increments: in closed form,
//     sum over bins <= b  =  (clipped counts of the populated bins <= b)  +  batch * (b + 1)  +  min(residual, b / rstep + 1).
__global__ __launch_bounds__(1024) void tile_lut16_kernel(const uint32_t* __restrict__ hist, const Range16* __restrict__ ranges, ClaheGeom g,
                                                         float lut_scale16, int clip16, uint16_t* __restrict__ luts, Range16* __restrict__ frame_ranges,
                                                         const uint32_t* __restrict__ frame_done, uint32_t* __restrict__ shift_hint, WideHint wide_hint)
{
    if (wide_hint.host && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)       // "this call has been executed": see WideHint
        __hip_atomic_store(wide_hint.host + 1, wide_hint.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // tile_hist12_kernel has written every LUT of this frame (bins 0..4095: all anybody reads) and the frame's range: one scalar load
    if (frame_done && frame_done[blockIdx.y]) return;
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_flo, s_fhi, s_fs;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int tiles = gridDim.x;
    const size_t tile_id = (size_t)blockIdx.y * tiles + blockIdx.x;
    const uint32_t* h = hist + tile_id * kHist16;
    uint16_t* lut = luts + tile_id * kHist16;
    auto block_scan = [&](uint32_t v, uint32_t& total) -> uint32_t {     // inclusive prefix of v over the 1024 threads
        const uint32_t incl = wave_incl_scan(v);
        __syncthreads();
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { const uint32_t x = s_w[k]; if (k < w) off += x; tot += x; }
        total = tot;
        return off + incl;
    };
    // frame range and shift: union / minimum over the tiles (every workgroup of the frame computes the same three numbers)
    if (t == 0) { s_flo = 0xffffu; s_fhi = 0; s_fs = 15u; }
    __syncthreads();
    {
        uint32_t l = 0xffffu, u = 0, sh = 15u;
        const Range16* fr = ranges + (size_t)blockIdx.y * tiles;
        for (int i = t; i < tiles; i += 1024) { const Range16 r = fr[i]; l = min(l, range_lo(r.lo)); u = max(u, range_hi(r.hi)); sh = min(sh, range_shift(r.hi)); }
        if (t < tiles) {
            __hip_atomic_fetch_min(&s_flo, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&s_fhi, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(&s_fs, sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const uint32_t flo = s_flo, fhi = s_fhi, sft = s_fs;
    const Range16 own_r = ranges[tile_id];
    if (blockIdx.x == 0 && t == 0) {
        Range16 r; r.lo = flo; r.hi = fhi | (sft << 16); frame_ranges[blockIdx.y] = r;
        if (shift_hint) hint_out(shift_hint, sft);                  // what the next frame's flat tiles should go along with
    }
    // Who reads this tile's LUT, and where?  Pixels of the tile itself and of its eight neighbours, at THEIR values.  So the LUT is
    // needed over the union of those nine tiles' ranges only, not over the frame's: one hot pixel at 65535 in a 12-bit frame then
    // costs the nine tiles around it a long LUT, not all 64 (the interpolation stages by the same rule, see clahe_interp16_kernel).
    uint32_t need_lo = 0xffffu, need_hi = 0u, need_mask = 0u;
    {
        const int tx0 = (int)blockIdx.x % g.tiles_x, ty0 = (int)blockIdx.x / g.tiles_x;
        const Range16* fr = ranges + (size_t)blockIdx.y * tiles;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = tx0 + dx, ny = ty0 + dy;
                if (nx < 0 || ny < 0 || nx >= g.tiles_x || ny >= g.tiles_y) continue;
                const Range16 r = fr[ny * g.tiles_x + nx];
                need_lo = min(need_lo, range_lo(r.lo)); need_hi = max(need_hi, range_hi(r.hi)); need_mask |= range_mask(r.lo);
            }
    }
    // tile_hist12_kernel has already written this tile's LUT for bins 0..4095 of ITS domain: that is all anybody reads if that domain
    // is the frame's and the neighbourhood stayed inside it
    if ((own_r.hi & kLutDone) && ((own_r.hi & kHistCompressed) ? range_shift(own_r.hi) : 0u) == sft && (need_hi >> sft) < (uint32_t)kBins12) return;
    const uint32_t own_lo = range_lo(own_r.lo), own_hi = range_hi(own_r.hi);
    // where this tile's counts are: at index value (careful sweeps, unshifted bets) or at index value >> own shift (shifted bets)
    const uint32_t own_store = (own_r.hi & kHistCompressed) ? range_shift(own_r.hi) : 0u;      // >= sft: sft is the minimum over the tiles
    const uint32_t dsh = own_store - sft;
    // counts of the compressed bins j0 .. j0 + 3 of this tile (bin j holds value j << shift), zero outside the tile's stored range
    auto load4 = [&](uint32_t j0, int* v) {
        if (own_store == sft) {                                     // stored in the frame's own domain: one 16-byte load
            const uint32_t jl = own_lo >> sft, jh = own_hi >> sft;
            if (j0 + 3 < jl || j0 > jh) { v[0] = v[1] = v[2] = v[3] = 0; return; }
            const u32x4 q = *reinterpret_cast<const u32x4*>(h + j0);
            const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (j0 + k >= jl && j0 + k <= jh) ? (int)x[k] : 0;
            return;
        }
        if (own_store != 0) {                                       // stored at a coarser shift than the frame's: every (1 << dsh)-th bin
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t j = j0 + k, b = j << sft;
                v[k] = ((j & ((1u << dsh) - 1u)) == 0u && b >= own_lo && b <= own_hi) ? (int)h[j >> dsh] : 0;
            }
            return;
        }
        if (sft == 0) {
            if (j0 + 3 < own_lo || j0 > own_hi) { v[0] = v[1] = v[2] = v[3] = 0; return; }
            const u32x4 q = *reinterpret_cast<const u32x4*>(h + j0);
            const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (j0 + k >= own_lo && j0 + k <= own_hi) ? (int)x[k] : 0;
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t b = (j0 + k) << sft;
            v[k] = (b >= own_lo && b <= own_hi) ? (int)h[b] : 0;
        }
    };
    const uint32_t jhi = need_hi >> sft, own_jlo = own_lo >> sft, own_jhi = own_hi >> sft;
    int batch = 0, residual = 0, rstep = 1;
    if (clip16 > 0) {
        uint32_t excess = 0;
        for (uint32_t j0 = (own_jlo & ~3u) + (uint32_t)t * 4; j0 <= own_jhi; j0 += 4096) {
            int v[4];
            load4(j0, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (v[k] > clip16) excess += (uint32_t)(v[k] - clip16);
        }
        uint32_t clipped;
        (void)block_scan(excess, clipped);
        batch = (int)clipped / kHist16;
        residual = (int)clipped - batch * kHist16;
        if (residual != 0) { rstep = kHist16 / residual; if (rstep < 1) rstep = 1; }
    }
    const uint32_t start = (need_lo >> sft) & ~3u;
    uint32_t running = 0;                                           // clipped counts of the populated bins before the chunk
    for (uint32_t c0 = start; c0 <= jhi; c0 += 4096) {
        // nobody in the neighbourhood has a value in this chunk's stretch of the value range: nothing will ever be looked up here
        if (!(need_mask & chunk_bits(c0 << sft, ((c0 + 4095u) << sft) | ((1u << sft) - 1u)))) continue;      // uniform over the workgroup
        const uint32_t j0 = c0 + (uint32_t)t * 4;
        const bool active = j0 <= jhi;
        int v[4] = {0, 0, 0, 0};
        uint32_t local = 0;
        if (active) {
            load4(j0, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (clip16 > 0 && v[k] > clip16) v[k] = clip16;
                local += (uint32_t)v[k];
                v[k] = (int)local;                                   // inclusive prefix within the thread's four bins
            }
        }
        uint32_t total;
        const uint32_t before = running + block_scan(local, total) - local;     // everything before this thread's first bin
        running += total;
        if (active) {
            uint32_t packed[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t sum = before + (uint32_t)v[k];
                if (clip16 > 0) {
                    const uint32_t b = (j0 + (uint32_t)k) << sft;        // the value this bin stands for
                    sum += (uint32_t)batch * (b + 1u);
                    if (residual != 0) sum += min((uint32_t)residual, b / (uint32_t)rstep + 1u);
                }
                int r = __float2int_rn(__fmul_rn((float)(int)sum, lut_scale16));
                r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
                if (k & 1) packed[k >> 1] |= (uint32_t)r << 16; else packed[k >> 1] = (uint32_t)r;
            }
            *reinterpret_cast<uint2*>(lut + j0) = make_uint2(packed[0], packed[1]);
        }
    }
}

// grid = (pairs * bands * subs, frames), 512 threads, 64 KiB dynamic LDS.  A workgroup owns the pixels whose horizontal tile pair
// is p (unclamped tx1 = p - 1) and whose unclamped ty1 is band - 1, so its four LUTs are fixed; a lane owns one 8-pixel group (16 B)
// of fixed columns -- column weights and ownership are lane constants -- and walks down the band's rows.  Ownership is decided by
// the reference's own float expressions on ranges widened by a few pixels, so a pair / band edge can never be mis-assigned;
// an 8-pixel group cut by a pair edge is visited by both neighbours, each storing only its own pixels.
// ENTRIES / THREADS: the table and the workgroup (8192 / 512: clahe_interp16_kernel; 16384 / 1024: clahe_interp16_mid_kernel).
// `id`: the work item (clahe_interp16_kernel: the workgroup's index).  mid_runs: clahe_interp16_mid_kernel is part of this call and takes
// the rectangles rect_is_mid() names; everybody else's are the small table's.
template <bool FMA, int ENTRIES, int THREADS>
__device__ __forceinline__ void interp16_item(long long id, const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                              uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                              const ClaheGeom& g, const uint16_t* __restrict__ luts,
                                              const Range16* __restrict__ frame_ranges, int subs, int n_frames,
                                              const Range16* __restrict__ tile_ranges, int mid_runs, const WideHint& wide_hint)
{
    constexpr bool MID = ENTRIES == kInterp16MidEntries;
    extern __shared__ __attribute__((aligned(16))) uint2 tab[];      // [ENTRIES] {a | b << 16, c | d << 16}
    const int t = threadIdx.x;
    const int npairs = g.tiles_x + 1, bands = g.tiles_y + 1;
    // Workgroups reach the 8 XCDs round-robin in launch order.  A ROW of workgroups -- the tiles_x + 1 pairs of one (frame, band,
    // sub-band) -- is given to ONE XCD, its pairs one after the other: the rectangles of neighbouring pairs meet in the middle of a
    // 128-byte line (a 4K pair is 960 bytes wide and starts at byte 480), and only an L2 that sees both halves writes whole lines
    // back.  grid = 8 * ceil(rows / 8) * (tiles_x + 1) workgroups, one dimension; rows beyond the last return at once.
    const int xcd = (int)(id & 7);
    const long long k = id >> 3;
    const int pr = (int)(k % npairs);
    const long long row = (k / npairs) * 8 + xcd;
    if (row >= (long long)bands * subs * n_frames) return;
    // frames are walked last-to-first: the histogram pass has just streamed the batch first-to-last, so its tail is what the memory-side
    // Infinity Cache still holds
    const int sub = (int)(row % subs), band = (int)((row / subs) % bands), f = n_frames - 1 - (int)(row / ((long long)subs * bands));
    const int ty1u = band - 1;
    const int ty1 = max(ty1u, 0), ty2 = min(ty1u + 1, g.tiles_y - 1);
    const int tx1 = max(pr - 1, 0), tx2 = min(pr, g.tiles_x - 1);
    const uint16_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * kHist16;
    const uint16_t* la = lf + ((size_t)ty1 * g.tiles_x + tx1) * kHist16;
    const uint16_t* lb = lf + ((size_t)ty1 * g.tiles_x + tx2) * kHist16;
    const uint16_t* lc = lf + ((size_t)ty2 * g.tiles_x + tx1) * kHist16;
    const uint16_t* ld = lf + ((size_t)ty2 * g.tiles_x + tx2) * kHist16;
    // everything below works in the frame's COMPRESSED domain j = value >> sft (see Range16; sft = 0 for ordinary content)
    const Range16 fr_raw = frame_ranges[f];
    const uint32_t sft = range_shift(fr_raw.hi);
    Range16 fr; fr.lo = fr_raw.lo >> sft; fr.hi = range_hi(fr_raw.hi) >> sft;
    // A range wider than the table is walked in WINDOWS of ENTRIES values: the table is staged once per window and a pixel is
    // finished in the window its value falls into (2-byte stores).  That re-reads the workgroup's pixels once per window, so it
    // cannot be done in place: in-place RECTANGLES that need several windows are left to clahe_interp16_wide_kernel (below).
    // The table only has to cover the values this workgroup's pixels can have: they lie in (at most) the four tiles whose LUTs it blends,
    // so the union of THOSE tiles' ranges replaces the frame's (a hot pixel, a bright corner widen the table of their own rectangles
    // only; tile_lut16_kernel writes every LUT over its tile's 3 x 3 neighbourhood, which contains these four).
    uint32_t presence;
    {
        const Range16* tr = tile_ranges + (size_t)f * g.tiles_x * g.tiles_y;
        const Range16 r00 = tr[ty1 * g.tiles_x + tx1], r01 = tr[ty1 * g.tiles_x + tx2], r10 = tr[ty2 * g.tiles_x + tx1], r11 = tr[ty2 * g.tiles_x + tx2];
        fr.lo = min(min(range_lo(r00.lo), range_lo(r01.lo)), min(range_lo(r10.lo), range_lo(r11.lo))) >> sft;
        fr.hi = max(max(range_hi(r00.hi), range_hi(r01.hi)), max(range_hi(r10.hi), range_hi(r11.hi))) >> sft;
        presence = range_mask(r00.lo) | range_mask(r01.lo) | range_mask(r10.lo) | range_mask(r11.lo);
    }
    const uint32_t start = fr.lo & ~3u;
    if (src_base == dst_base) {                                     // in place: by rectangles (rect_owner_in_place), uniform
        const int owner = rect_owner_in_place(fr.hi - start, mid_runs);
        if (!MID && owner != 0) { if (t == 0 && fr.hi - start < (uint32_t)kInterp16MidEntries) wide_seen(wide_hint); return; }
        if (MID && owner != 1) return;
    }
    if (MID) {
        if (!rect_is_mid(fr.hi - start, presence)) return;                    // uniform: the small table's rectangle
    } else {
        if (rect_is_mid(fr.hi - start, presence)) {                           // uniform
            if (t == 0) wide_seen(wide_hint);
            if (mid_runs) return;                                   // clahe_interp16_mid_kernel's rectangle
        }
    }
    const bool multi = fr.hi - start >= (uint32_t)ENTRIES;
    // A range of at most kInterp16F32Entries values (every 12-bit source) gets the table as FLOATS, {a, c, b, d} in 16 bytes: one
    // ds_read_b128 per pixel feeds v_pk_mul / v_pk_add directly and the four ushort -> float conversions per pixel are gone
    // (the blend was VALU-bound: ~25 instructions per pixel, now ~12).  Same 64 KiB of LDS either way.
    const bool f32tab = fr.hi - start < (uint32_t)kInterp16F32Entries;
    f32x4* const tabf = reinterpret_cast<f32x4*>(tab);

    // rows of the band (as clahe_interp_kernel): ideal range widened, then trimmed with the float expression
    const int y_lo_band = (int)max(0LL, ((long long)(2 * band - 1) * g.tile_h) / 2 - kBandMargin);
    const int y_hi_band = (int)min((long long)g.height, ((long long)(2 * band + 1) * g.tile_h + 1) / 2 + kBandMargin);
    const int nrows = max(0, y_hi_band - y_lo_band);
    int y_lo = y_lo_band + (int)((long long)nrows * sub / subs);
    int y_hi = y_lo_band + (int)((long long)nrows * (sub + 1) / subs);
    auto ty1_of = [&](int y) { return floor_f32_to_int(tile_coord<FMA>(y, g.inv_th)); };
    while (y_lo < y_hi && ty1_of(y_lo) != ty1u) ++y_lo;
    while (y_hi > y_lo && ty1_of(y_hi - 1) != ty1u) --y_hi;
    // columns of the pair, in 8-pixel groups
    const int x_lo = (int)max(0LL, ((long long)(2 * pr - 1) * g.tile_w) / 2 - kBandMargin);
    const int x_hi = (int)min((long long)g.width, ((long long)(2 * pr + 1) * g.tile_w + 1) / 2 + kBandMargin);
    if (x_lo >= x_hi || y_lo >= y_hi) return;                       // uniform over the workgroup
    const int g_lo = x_lo >> 3, ngroups = ((x_hi + 7) >> 3) - g_lo;
    const int phases = max(1, THREADS / ngroups);
    const int passes = (ngroups + THREADS - 1) / THREADS;          // > 1 only for tiles wider than 8 * THREADS pixels
    const uint8_t* src = src_base + (long long)f * src_frame;
    uint8_t* dst = dst_base + (long long)f * dst_frame;

    // Which windows does this workgroup's rectangle populate at all?  (One extra read of its pixels: real images are locally much
    // narrower than their frame.)  Bit w of s_windows: some owned pixel has (value - start) / ENTRIES == w; at most 8 (mid kernel: 4) windows.
    __shared__ uint32_t s_windows;
    if (multi) {
        if (t == 0) s_windows = 0;
        __syncthreads();
        uint32_t seen = 0;
        const bool al16 = ((((uintptr_t)src | (unsigned long long)src_step) & 15) == 0);
        for (int pass = 0; pass < passes; ++pass) {
            const int gi = pass * THREADS + (passes > 1 ? t : t % ngroups);
            const int phase = passes > 1 ? 0 : t / ngroups;
            if (gi >= ngroups || phase >= phases) continue;
            const int x0 = (g_lo + gi) << 3;
            auto note = [&](uint32_t v) { seen |= 1u << (((v >> sft) - start) / (uint32_t)ENTRIES); };
            if (al16 && x0 + 8 <= g.width) {                        // a superset of the owned pixels is fine here
                for (int y = y_lo + phase; y < y_hi; y += phases) {
                    const u32x4 q = *reinterpret_cast<const u32x4*>(src + (long long)y * src_step + 2 * (long long)x0);
                    note(q.x & 0xffffu); note(q.x >> 16); note(q.y & 0xffffu); note(q.y >> 16);
                    note(q.z & 0xffffu); note(q.z >> 16); note(q.w & 0xffffu); note(q.w >> 16);
                }
            } else {
                for (int y = y_lo + phase; y < y_hi; y += phases)
                    for (int jx = 0; jx < 8 && x0 + jx < g.width; ++jx)
                        note(*reinterpret_cast<const uint16_t*>(src + (long long)y * src_step + 2 * (long long)(x0 + jx)));
            }
        }
        if (seen) __hip_atomic_fetch_or(&s_windows, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
    }
    const uint32_t windows = multi ? s_windows : 1u;

    // One window and one pass -- every call on 10/12/13-bit content: a lane's first four rows are requested BEFORE the table is staged
    // (the staging is ~4 us of L2 reads and LDS writes during which the workgroup had nothing in flight), and inside the row loop the
    // next four are requested before the current four are blended.  The kernel runs at 4 waves per SIMD (64 KiB of LDS per workgroup):
    // without the second register set every iteration exposed the full HBM latency (measured alone, 16 4K frames: 142 us, see
    // profiles/r03_n_*).
    constexpr int kRows = 4;
    const bool aligned = ((((uintptr_t)src | (uintptr_t)dst | (unsigned long long)src_step | (unsigned long long)dst_step) & 15) == 0);
    u32x4 q[kRows];
    bool pre_valid = false;
    if (!multi && passes == 1 && aligned) {
        const int gi = t % ngroups, phase = t / ngroups;
        const int x0 = (g_lo + gi) << 3, y = y_lo + phase;
        if (phase < phases && x0 + 8 <= g.width && y + (kRows - 1) * phases < y_hi) {
#pragma unroll
            for (int k = 0; k < kRows; ++k) q[k] = *reinterpret_cast<const u32x4*>(src + (long long)(y + k * phases) * src_step + 2 * (long long)x0);
            pre_valid = true;
        }
    }

    for (uint32_t w0 = start, wi = 0; w0 <= fr.hi; w0 += (uint32_t)ENTRIES, ++wi) {
        if (!((windows >> wi) & 1u)) continue;                      // uniform: nothing of this rectangle lives in that window
        __syncthreads();                                            // the previous window's table is no longer read
        {
            const uint32_t w1 = min(fr.hi, w0 + (uint32_t)ENTRIES - 1);
            // One table entry per lane and step, consecutive lanes -> consecutive entries: conflict-free LDS writes; the sixteen 2-byte
            // loads of four steps are in flight together.
            const uint32_t n = w1 - w0 + 1;
            if (MID) {
                // four entries per lane and step from four 8-byte loads, written as two 16-byte stores, two steps in flight: 16384
                // entries are two trips to the LUTs (another XCD's tile kernels wrote them: ~2 us each) instead of the four that the
                // 2-byte loads below would make.  w0 is a multiple of four; the last step may read up to three entries past the range
                // (inside the 65536-entry LUT: the range ends at 65535 at most; never looked up).
                const uint32_t n4 = (n + 3u) & ~3u;
#pragma unroll 2
                for (uint32_t i4 = (uint32_t)t * 4u; i4 < n4; i4 += (uint32_t)THREADS * 4u) {
                    const uint32_t v = w0 + i4;
                    const uint2 A = *reinterpret_cast<const uint2*>(la + v), B = *reinterpret_cast<const uint2*>(lb + v);
                    const uint2 C = *reinterpret_cast<const uint2*>(lc + v), D = *reinterpret_cast<const uint2*>(ld + v);
                    u32x4 e0, e1;
                    e0.x = (A.x & 0xffffu) | (B.x << 16);         e0.y = (C.x & 0xffffu) | (D.x << 16);
                    e0.z = (A.x >> 16) | (B.x & 0xffff0000u);     e0.w = (C.x >> 16) | (D.x & 0xffff0000u);
                    e1.x = (A.y & 0xffffu) | (B.y << 16);         e1.y = (C.y & 0xffffu) | (D.y << 16);
                    e1.z = (A.y >> 16) | (B.y & 0xffff0000u);     e1.w = (C.y >> 16) | (D.y & 0xffff0000u);
                    u32x4* o = reinterpret_cast<u32x4*>(tab + i4);
                    o[0] = e0; o[1] = e1;
                }
            } else
            for (uint32_t i0 = 0; i0 < n; i0 += 4 * THREADS) {
                uint32_t va[4], vb[4], vc[4], vd[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t v = min(w0 + i0 + (uint32_t)(k * THREADS + t), w1);
                    va[k] = la[v]; vb[k] = lb[v]; vc[k] = lc[v]; vd[k] = ld[v];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t i = i0 + (uint32_t)(k * THREADS + t);
                    if (i >= n) continue;
                    if (f32tab) {
                        const f32x4 e = {(float)va[k], (float)vc[k], (float)vb[k], (float)vd[k]};      // {a, c, b, d}
                        tabf[i] = e;
                    } else {
                        uint2 e; e.x = va[k] | (vb[k] << 16); e.y = vc[k] | (vd[k] << 16);
                        tab[i] = e;
                    }
                }
            }
        }
        __syncthreads();
        for (int pass = 0; pass < passes; ++pass) {
            const int gi = pass * THREADS + (passes > 1 ? t : t % ngroups);
            const int phase = passes > 1 ? 0 : t / ngroups;
            if (gi >= ngroups || phase >= phases) continue;
            const int x0 = (g_lo + gi) << 3;
            float xa[8], xa1[8];
            uint32_t own = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float txf = tile_coord<FMA>(x0 + j, g.inv_tw);
                const int txu = floor_f32_to_int(txf);
                xa[j] = __fsub_rn(txf, (float)txu);
                xa1[j] = __fsub_rn(1.0f, xa[j]);
                int q = txu + 1; q = q < 0 ? 0 : (q > g.tiles_x ? g.tiles_x : q);
                if (q == pr && x0 + j < g.width) own |= 1u << j;
            }
            if (!own) continue;
            const bool vec_ok = own == 0xffu && aligned && !multi;
            auto blend_row = [&](int y, const uint32_t* px, uint32_t* res) {
                const float tyf = tile_coord<FMA>(y, g.inv_th);
                const float ya = __fsub_rn(tyf, (float)ty1u), ya1 = __fsub_rn(1.0f, ya);
                if (f32tab) {
                    const f32x2 yv = {ya1, ya};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t idx = min((px[j] >> sft) - w0, (uint32_t)kInterp16F32Entries - 1);
                        const f32x4 e = tabf[idx];
                        const f32x2 ac = {e.x, e.y}, bd = {e.z, e.w}, xw = {xa1[j], xa[j]};
                        float r;
                        if (FMA) {
                            const f32x2 tb = pk_fma_bcast_lo(ac, xw, pk_mul_bcast_hi(bd, xw));       // {fma(a,xa1,b*xa), fma(c,xa1,d*xa)}
                            r = __fmaf_rn(tb.x, ya1, __fmul_rn(tb.y, ya));
                        } else {
                            const f32x2 tb = (pk_mul_bcast_lo(ac, xw) + pk_mul_bcast_hi(bd, xw)) * yv; // nine individually rounded operations
                            r = __fadd_rn(tb.x, tb.y);
                        }
                        const int ri = __float2int_rn(r);
                        res[j] = (uint32_t)(ri < 0 ? 0 : (ri > 65535 ? 65535 : ri));
                    }
                    return;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // owned pixels of this window index the table directly; anything else is masked out later: clamp its index
                    const uint32_t idx = min((px[j] >> sft) - w0, (uint32_t)ENTRIES - 1);
                    const uint2 e = tab[idx];
                    const float a = (float)(e.x & 0xffffu), b = (float)(e.x >> 16), c = (float)(e.y & 0xffffu), d = (float)(e.y >> 16);
                    int r = __float2int_rn(clahe_blend_f<FMA>(a, b, c, d, xa[j], xa1[j], ya, ya1));
                    res[j] = (uint32_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
                }
            };
            int y = y_lo + phase;
            if (vec_ok) {
                // four rows per set, two sets: the next set is in flight while the current one is blended (128 B per lane)
                auto do_vec_row = [&](int yy, const u32x4& qq) {
                    const uint32_t px[8] = {qq.x & 0xffffu, qq.x >> 16, qq.y & 0xffffu, qq.y >> 16, qq.z & 0xffffu, qq.z >> 16, qq.w & 0xffffu, qq.w >> 16};
                    uint32_t res[8];
                    blend_row(yy, px, res);
                    u32x4 o;
                    o.x = res[0] | (res[1] << 16); o.y = res[2] | (res[3] << 16); o.z = res[4] | (res[5] << 16); o.w = res[6] | (res[7] << 16);
                    *reinterpret_cast<u32x4*>(dst + (long long)yy * dst_step + 2 * (long long)x0) = o;
                };
                auto ld_row = [&](int yy) { return *reinterpret_cast<const u32x4*>(src + (long long)yy * src_step + 2 * (long long)x0); };
                bool have = y + (kRows - 1) * phases < y_hi;
                if (have && !pre_valid) {
#pragma unroll
                    for (int k = 0; k < kRows; ++k) q[k] = ld_row(y + k * phases);
                }
                pre_valid = false;
                while (have) {
                    const int y2 = y + kRows * phases;
                    const bool more = y2 + (kRows - 1) * phases < y_hi;
                    u32x4 nq[kRows];
                    if (more) {
#pragma unroll
                        for (int k = 0; k < kRows; ++k) nq[k] = ld_row(y2 + k * phases);
                    }
#pragma unroll
                    for (int k = 0; k < kRows; ++k) { do_vec_row(y + k * phases, q[k]); __builtin_amdgcn_sched_barrier(0); }
                    if (more) {
#pragma unroll
                        for (int k = 0; k < kRows; ++k) q[k] = nq[k];
                    }
                    y = y2; have = more;
                }
                for (; y < y_hi; y += phases) do_vec_row(y, *reinterpret_cast<const u32x4*>(src + (long long)y * src_step + 2 * (long long)x0));
            } else {
                for (; y < y_hi; y += phases) {
                    const uint8_t* sp = src + (long long)y * src_step + 2 * (long long)x0;
                    uint8_t* dp = dst + (long long)y * dst_step + 2 * (long long)x0;
                    uint32_t px[8], res[8], todo = 0;
                    if (own == 0xffu && aligned) {
                        const u32x4 q = *reinterpret_cast<const u32x4*>(sp);
                        px[0] = q.x & 0xffffu; px[1] = q.x >> 16; px[2] = q.y & 0xffffu; px[3] = q.y >> 16;
                        px[4] = q.z & 0xffffu; px[5] = q.z >> 16; px[6] = q.w & 0xffffu; px[7] = q.w >> 16;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) px[j] = (own >> j) & 1u ? *reinterpret_cast<const uint16_t*>(sp + 2 * j) : 0xffffffffu;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (((own >> j) & 1u) && (px[j] >> sft) - w0 < (uint32_t)ENTRIES) todo |= 1u << j;   // this window's pixels
                    if (!todo) continue;
                    blend_row(y, px, res);
#pragma unroll
                    for (int j = 0; j < 8; ++j) if ((todo >> j) & 1u) *reinterpret_cast<uint16_t*>(dp + 2 * j) = (uint16_t)res[j];
                }
            }
        }
    }
}


template <bool FMA>
__global__ __launch_bounds__(kInterp16Threads) void clahe_interp16_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                         uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                         ClaheGeom g, const uint16_t* __restrict__ luts,
                                                                         const Range16* __restrict__ frame_ranges, int subs, int n_frames,
                                                                         const Range16* __restrict__ tile_ranges, uint32_t* shift_hint,
                                                                         int mid_runs, WideHint wide_hint)
{
    hint_roll(shift_hint);
    interp16_item<FMA, kInterp16Entries, kInterp16Threads>(blockIdx.x, src_base, src_step, src_frame, dst_base, dst_step, dst_frame, g, luts,
                                                           frame_ranges, subs, n_frames, tile_ranges, mid_runs, wide_hint);
}

// PERSISTENT: grid = min(work items, CUs rounded to a multiple of 8) workgroups of 1024 threads with 128 KiB of dynamic LDS (one per CU
// is all that fits; launched one per item, the thousands that only return cost 23 us).  The work items are clahe_interp16_kernel's,
// item for item (same `subs`, same XCD dealing: item & 7 -- the grid is a multiple of 8, so a workgroup stays on its XCD).  Frames whose
// whole range fits the small table are known from a bit mask built once per workgroup.  In place it keeps to frames of at most 16384
// values (one window per rectangle); wider ones are the gathering kernel's, whole.
template <bool FMA>
__global__ __launch_bounds__(kInterp16MidThreads) void clahe_interp16_mid_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                                uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                                ClaheGeom g, const uint16_t* __restrict__ luts,
                                                                                const Range16* __restrict__ frame_ranges, int subs, int n_frames,
                                                                                const Range16* __restrict__ tile_ranges)
{
    __shared__ unsigned long long s_wide[16];                        // bit f % 64 of word f / 64: frame f may hold such a rectangle (<= 1024 frames per launch)
    const int t = threadIdx.x;
    for (int f0 = 0; f0 < n_frames; f0 += kInterp16MidThreads) {
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
    const int npairs = g.tiles_x + 1, bands = g.tiles_y + 1;
    const long long rows_total = (long long)bands * subs * n_frames;
    const long long items = (rows_total + 7) / 8 * 8 * npairs;
    const WideHint none{nullptr, nullptr, 0u};
    for (long long id = blockIdx.x; id < items; id += gridDim.x) {
        const long long row = ((id >> 3) / npairs) * 8 + (id & 7);
        if (row >= rows_total) continue;
        const int f = n_frames - 1 - (int)(row / ((long long)subs * bands));
        if (!((s_wide[(f >> 6) & 15] >> (f & 63)) & 1ull)) continue;  // uniform: no such rectangle in this frame
        __syncthreads();                                            // the previous item's table is no longer read
        interp16_item<FMA, kInterp16MidEntries, kInterp16MidThreads>(id, src_base, src_step, src_frame, dst_base, dst_step, dst_frame, g, luts,
                                                                     frame_ranges, subs, n_frames, tile_ranges, 1, none);
    }
}

// IN-PLACE calls, the RECTANGLES whose populated range does not fit one window of a table (rect_owner_in_place; full-range 16-bit
// sources, the neighbourhood of a hot pixel; out of place such rectangles go through the table in several windows): one pixel per lane, four ushort gathers from the per-tile LUTs in L2 -- bound by the divergent gathers themselves (up to 64 cache lines per wave
// instruction).  Launched after clahe_interp16_kernel on every call; a workgroup whose frame was handled there returns at once, so
// the grid is kept small: grid = (min(items, max(512, 2048 / frames)), 1, frames) workgroups walking (row, 256-pixel block) items in row-major order
// with stride gridDim.x -- the rows in flight at any moment are neighbours, so the LUTs they gather from (two tile rows) stay in L2
// (rows strided over the whole image measured 2x slower: all 64 tiles' LUTs in use at once).
__global__ __launch_bounds__(kThreads) void clahe_interp16_wide_kernel(const uint8_t* __restrict__ src_base, long long src_step, long long src_frame,
                                                                      uint8_t* __restrict__ dst_base, long long dst_step, long long dst_frame,
                                                                      ClaheGeom g, const uint16_t* __restrict__ luts,
                                                                      const Range16* __restrict__ frame_ranges, int mid_runs,
                                                                      const Range16* __restrict__ tile_ranges)
{
    const int f = blockIdx.z;
    const Range16 fr = frame_ranges[f];
    const uint32_t sft = range_shift(fr.hi);                        // the LUTs are stored at index value >> sft
    // a frame whose whole range is one window of a table that runs has no rectangle for this kernel; nor has a call that is not in place
    if ((range_hi(fr.hi) >> sft) - ((fr.lo >> sft) & ~3u) < (uint32_t)(mid_runs ? kInterp16MidEntries : kInterp16Entries) || src_base != dst_base) return;
    const Range16* tr = tile_ranges + (size_t)f * g.tiles_x * g.tiles_y;
    // Which rectangles -- (tile pair, band) -- are this kernel's?  (rect_owner_in_place on the range of the rectangle's four tiles, as the
    // table kernels compute it.)  Worked out once per workgroup into LDS, one byte per rectangle, so that a pixel costs one LDS read and
    // a 256-pixel block whose ends are both somebody else's (and which is no wider than a tile: at most two rectangles) nothing more.
    constexpr int kMaxRects = 4096;
    __shared__ uint8_t s_mine[kMaxRects];
    const int npairs = g.tiles_x + 1, nbands = g.tiles_y + 1;
    const bool tabled = npairs * nbands <= kMaxRects;               // (more rectangles than that: every pixel asks the tiles' ranges itself)
    auto rect_mine = [&](int pr, int band) {
        const int tx1 = max(pr - 1, 0), tx2 = min(pr, g.tiles_x - 1), ty1 = max(band - 1, 0), ty2 = min(band, g.tiles_y - 1);
        const Range16 r00 = tr[ty1 * g.tiles_x + tx1], r01 = tr[ty1 * g.tiles_x + tx2], r10 = tr[ty2 * g.tiles_x + tx1], r11 = tr[ty2 * g.tiles_x + tx2];
        const uint32_t rlo = min(min(range_lo(r00.lo), range_lo(r01.lo)), min(range_lo(r10.lo), range_lo(r11.lo))) >> sft;
        const uint32_t rhi = max(max(range_hi(r00.hi), range_hi(r01.hi)), max(range_hi(r10.hi), range_hi(r11.hi))) >> sft;
        return rect_owner_in_place(rhi - (rlo & ~3u), mid_runs) == 2;
    };
    if (tabled) {
        for (int i = threadIdx.x; i < npairs * nbands; i += kThreads) s_mine[i] = rect_mine(i % npairs, i / npairs) ? 1 : 0;
        __syncthreads();
    }
    auto pair_of = [&](int x) { const int q = floor_f32_to_int(tile_coord(x, g.inv_tw, g.contract)) + 1; return q < 0 ? 0 : (q > g.tiles_x ? g.tiles_x : q); };
    auto band_of = [&](int y) { const int q = floor_f32_to_int(tile_coord(y, g.inv_th, g.contract)) + 1; return q < 0 ? 0 : (q > g.tiles_y ? g.tiles_y : q); };
    auto mine = [&](int pr, int band) { return tabled ? s_mine[band * npairs + pr] != 0 : rect_mine(pr, band); };
    const bool by_block = tabled && g.tile_w >= kThreads;
    const uint16_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * kHist16;
    const int bx = (g.width + kThreads - 1) / kThreads;
    const long long items = (long long)bx * g.height;
    const uint8_t* src = src_base + (long long)f * src_frame;
    uint8_t* dst = dst_base + (long long)f * dst_frame;
    constexpr int kChains = 4;                                       // independent pixel -> gather -> store chains per lane and iteration:
    for (long long it0 = blockIdx.x; it0 < items; it0 += (long long)kChains * gridDim.x) {   // the kernel is bound by memory latency
        int xs[kChains], ys[kChains];
        uint32_t v[kChains];
        bool on[kChains];
#pragma unroll
        for (int k = 0; k < kChains; ++k) {
            const long long it = it0 + (long long)k * gridDim.x;
            ys[k] = (int)(it / bx);
            xs[k] = (int)(it - (long long)ys[k] * bx) * kThreads + threadIdx.x;
            on[k] = it < items && xs[k] < g.width;
            if (by_block && it < items) {                            // uniform over the workgroup
                const int xb = (int)(it - (long long)ys[k] * bx) * kThreads, band = band_of(ys[k]);
                if (!mine(pair_of(xb), band) && !mine(pair_of(min(xb + kThreads - 1, g.width - 1)), band)) on[k] = false;
            }
            v[k] = on[k] ? (uint32_t)*reinterpret_cast<const uint16_t*>(src + (long long)ys[k] * src_step + 2 * (long long)xs[k]) >> sft : 0u;
        }
        float a[kChains], b[kChains], c[kChains], d[kChains], xa[kChains], ya[kChains];
#pragma unroll
        for (int k = 0; k < kChains; ++k) {
            const float txf = tile_coord(xs[k], g.inv_tw, g.contract);
            int tx1 = floor_f32_to_int(txf);
            xa[k] = __fsub_rn(txf, (float)tx1);
            int tx2 = tx1 + 1; tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
            const float tyf = tile_coord(ys[k], g.inv_th, g.contract);
            int ty1 = floor_f32_to_int(tyf);
            ya[k] = __fsub_rn(tyf, (float)ty1);
            int ty2 = ty1 + 1; ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
            if (on[k]) on[k] = mine(min(max(floor_f32_to_int(txf) + 1, 0), g.tiles_x), min(max(floor_f32_to_int(tyf) + 1, 0), g.tiles_y));
            if (on[k]) {
                a[k] = (float)lf[((size_t)ty1 * g.tiles_x + tx1) * kHist16 + v[k]]; b[k] = (float)lf[((size_t)ty1 * g.tiles_x + tx2) * kHist16 + v[k]];
                c[k] = (float)lf[((size_t)ty2 * g.tiles_x + tx1) * kHist16 + v[k]]; d[k] = (float)lf[((size_t)ty2 * g.tiles_x + tx2) * kHist16 + v[k]];
            } else {
                a[k] = b[k] = c[k] = d[k] = 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < kChains; ++k) {
            if (!on[k]) continue;
            const float xa1 = __fsub_rn(1.0f, xa[k]), ya1 = __fsub_rn(1.0f, ya[k]);
            int r = __float2int_rn(g.contract ? clahe_blend_f<true>(a[k], b[k], c[k], d[k], xa[k], xa1, ya[k], ya1)
                                              : clahe_blend_f<false>(a[k], b[k], c[k], d[k], xa[k], xa1, ya[k], ya1));
            r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
            *reinterpret_cast<uint16_t*>(dst + (long long)ys[k] * dst_step + 2 * (long long)xs[k]) = (uint16_t)r;
        }
    }
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
                                                                  ClaheGeom g, const uint16_t* __restrict__ lutT,
                                                                  const Range16* __restrict__ frame_ranges, uint32_t* shift_hint)
{
    hint_roll(shift_hint);
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
    const uint32_t v = (uint32_t)*reinterpret_cast<const uint16_t*>(src_base + (long long)f * src_frame + (long long)y * src_step + 2 * (long long)x)
                       >> range_shift(frame_ranges[f].hi);       // the LUTs are stored at index value >> shift
    const uint16_t* e = lutT + ((size_t)f * kHist16 + v) * tiles;
    const float a = (float)e[ty1 * g.tiles_x + tx1], b = (float)e[ty1 * g.tiles_x + tx2];
    const float c = (float)e[ty2 * g.tiles_x + tx1], d = (float)e[ty2 * g.tiles_x + tx2];
    int r = __float2int_rn(g.contract ? clahe_blend_f<true>(a, b, c, d, xa, xa1, ya, ya1) : clahe_blend_f<false>(a, b, c, d, xa, xa1, ya, ya1));
    r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
    *reinterpret_cast<uint16_t*>(dst_base + (long long)f * dst_frame + (long long)y * dst_step + 2 * (long long)x) = (uint16_t)r;
}

}  // namespace mi
