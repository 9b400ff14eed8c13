// equalize.hip.h -- K1 histogram partials, K2 CDF->LUT, K3 LUT apply (+UV): the three-kernel equalizeHist path
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
#pragma once
#include "common.hip.h"

namespace mi {
// ---------------------------------------------------------------------------------------------
// K1  histogram partials.  grid = (B, n_frames); partial[f][b][256].
// Reads W*H bytes per frame once; writes B KiB per frame.  Bound: HBM read.
// ---------------------------------------------------------------------------------------------
// 256 threads: 512-thread workgroups (the change that took the CLAHE tile histograms from 132 to 124 us) bought nothing here
// (137.6 vs 139 us per 64 4K frames behind the previous launch's write drain) and cost the strided-ROI path 12 %, whose rows are
// shorter than 512 x 16 bytes.
constexpr int kHistThreads = kThreads;

// Strided rows (ROI views) whose pitch is a multiple of 16: every row then has the SAME alignment phase, so the rows of a workgroup
// (a contiguous band) are walked as (row, 16-byte slot) items NT apart -- four predicated vector loads in flight per lane, as in the
// tile histograms -- instead of row by row with one vector per lane in flight (a 3840-byte row is 240 vectors for 256 lanes).
// The head / tail bytes of the rows (at most 15 + 15 per row) follow as byte items.  `row0` points at the first row of the band.
struct RowBand {
    int head, slots, tail;              // bytes before the first aligned vector, whole vectors, bytes after the last
    __device__ __forceinline__ RowBand(const void* aligned_on, long long row_bytes)
    {
        long long hd = (16 - (long long)((uintptr_t)aligned_on & 15)) & 15;
        if (hd > row_bytes) hd = row_bytes;
        head = (int)hd; slots = (int)((row_bytes - hd) >> 4); tail = (int)(row_bytes - hd - ((long long)slots << 4));
    }
};

template <int NT>
__device__ __forceinline__ void hist_rows(uint32_t* h, const uint8_t* row0, long long step, long long row_bytes, int nrows)
{
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const RowBand rb(row0, row_bytes);
    if (rb.slots > 0) {
        const long long items = (long long)nrows * rb.slots;
        int row = t / rb.slots, slot = t - row * rb.slots;
        const int drow = NT / rb.slots, dslot = NT - drow * rb.slots;
        const uint8_t* vb = row0 + rb.head;
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (long long it = t; it < items; it += 4 * NT) {
            u32x4 q[4]; bool qv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qv[k] = it + (long long)k * NT < items;
                const u32x4* ptr = reinterpret_cast<const u32x4*>(vb + (long long)row * step + (slot << 4));
                q[k] = qv[k] ? *ptr : zero;
                row += drow; slot += dslot;
                if (slot >= rb.slots) { slot -= rb.slots; ++row; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (qv[k]) hist_add_vec(h, q[k], copy);
        }
    }
    const int e = rb.head + rb.tail;
    if (e > 0) {
        const long long items = (long long)nrows * e;
        for (long long it = t; it < items; it += NT) {
            const int row = (int)(it / e), c = (int)(it - (long long)row * e);
            const long long off = c < rb.head ? c : row_bytes - rb.tail + (c - rb.head);
            lds_inc(h, ((uint32_t)row0[(long long)row * step + off] << kCopyShift) + copy);
        }
    }
}

// this workgroup's share (blockIdx.x of gridDim.x) of frame blockIdx.y, counted into the LDS histogram h (zeroed here)
__device__ __forceinline__ void hist_block(uint32_t* h, const PlaneBatch& p)
{
    const int t = threadIdx.x;
    for (int i = t; i < 256 * kCopies; i += kHistThreads) h[i] = 0;
    __syncthreads();
    const uint8_t* base = p.src + (long long)blockIdx.y * p.src_frame;
    if (p.rows == 1) {
        hist_flat<kHistThreads>(h, base, p.row_bytes, blockIdx.x, gridDim.x);
    } else if ((p.src_step & 15) == 0) {
        const int r0 = (int)((long long)p.rows * blockIdx.x / gridDim.x), r1 = (int)((long long)p.rows * (blockIdx.x + 1) / gridDim.x);
        hist_rows<kHistThreads>(h, base + (long long)r0 * p.src_step, p.src_step, p.row_bytes, r1 - r0);
    } else {
        for (int r = blockIdx.x; r < p.rows; r += gridDim.x) hist_flat<kHistThreads>(h, base + (long long)r * p.src_step, p.row_bytes, 0, 1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(kHistThreads) void hist_partial_kernel(PlaneBatch p, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[256 * kCopies];
    hist_block(h, p);
    const int t = threadIdx.x;
    if (t < 256) partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = lds_hist_bin(h, t);
}

// ---------------------------------------------------------------------------------------------
// K2  CDF -> LUT (oracle: orc_equalize_lut).  grid = n_frames, 256 threads = bins.
// partial[f][b][256] summed over b (b = 1 turns it into "LUT from a finished histogram").
// ---------------------------------------------------------------------------------------------
// The CDF -> LUT arithmetic for bin t = threadIdx.x given this bin's count c (all 256 threads call it).
// histogram.cpp cv::equalizeHist after the histogram: first non-zero bin i, constant-image shortcut,
// scale = 255.f/(total - hist[i]), lut[j] = saturate_cast<uchar>(sum_j * scale) with cvRound.
struct EqLutShared { uint32_t wave[4]; int first[4]; uint32_t hfirst; };

__device__ __forceinline__ uint8_t equalize_lut_value(uint32_t c, int total, EqLutShared* sh)
{
    const int t = threadIdx.x;
    const unsigned long long nz = __ballot(c != 0);
    __syncthreads();                                               // sh may still be read from a previous use
    if ((t & 63) == 0) sh->first[t >> 6] = nz ? (t + __builtin_ctzll(nz)) : 256;
    const uint32_t cdf = block_incl_scan(c, sh->wave, nullptr);    // contains the barriers that publish first[]
    const int first = min(min(sh->first[0], sh->first[1]), min(sh->first[2], sh->first[3]));
    if (t == first) sh->hfirst = c;
    __syncthreads();
    const uint32_t hfirst = sh->hfirst;
    if ((int)hfirst == total) return (uint8_t)first;                // dst.setTo(i)
    if (t <= first) return 0;
    const float scale = __fdiv_rn(255.0f, (float)(total - (int)hfirst));
    const int sum = (int)(cdf - hfirst);                            // bins first+1 .. t
    int r = __float2int_rn(__fmul_rn((float)sum, scale));           // cvRound: nearest, ties to even
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    return (uint8_t)r;
}

__global__ __launch_bounds__(kThreads) void equalize_lut_kernel(const uint32_t* __restrict__ partial, int nparts, int total,
                                                               uint8_t* __restrict__ lut_out, int32_t* __restrict__ hist_out)
{
    __shared__ EqLutShared sh;
    const int t = threadIdx.x, f = blockIdx.x;
    const uint32_t* pp = partial + (size_t)f * nparts * 256 + t;
    uint32_t c = 0;
    int b = 0;
    for (; b + 4 <= nparts; b += 4) {
        const uint32_t c0 = pp[(size_t)b * 256], c1 = pp[(size_t)(b + 1) * 256], c2 = pp[(size_t)(b + 2) * 256], c3 = pp[(size_t)(b + 3) * 256];
        c += c0 + c1 + c2 + c3;
    }
    for (; b < nparts; ++b) c += pp[(size_t)b * 256];
    if (hist_out) hist_out[(size_t)f * 256 + t] = (int32_t)c;
    if (!lut_out) return;
    lut_out[(size_t)f * 256 + t] = equalize_lut_value(c, total, &sh);
}

// ---------------------------------------------------------------------------------------------
// K1+K2 in one launch for FEW frames (a single frame of a stream, a cv::Mat call): every workgroup adds its non-zero bins to the
// frame's global histogram with agent-scope atomics and takes an arrival number; the LAST one to arrive exchanges the 256 counts
// out (leaving the scratch zeroed for the next launch) and writes the LUT.  Nobody waits for anybody -- the "last block" pattern
// has no inter-workgroup dependency, so unlike the fused kernel it needs neither bounded waits nor a finish kernel -- and one
// launch (plus its gap) is gone against K1 -> K2.  With lut_apply_kernel behind it a single 4K frame costs two launches.
// grid = (B, n_frames), 256 threads.  ghist[f][256] and cnt[f] must be zero on entry (they are zero again on exit).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kHistThreads) void hist_lut_kernel(PlaneBatch p, uint32_t* __restrict__ ghist, uint32_t* __restrict__ cnt,
                                                                int total, uint8_t* __restrict__ lut_out)
{
    __shared__ uint32_t h[256 * kCopies];
    __shared__ EqLutShared sh;
    __shared__ int s_last;
    hist_block(h, p);
    const int t = threadIdx.x, f = blockIdx.y;
    uint32_t* gh = ghist + (size_t)f * 256;
    const uint32_t mine = lds_hist_bin(h, t);
    if (mine) __hip_atomic_fetch_add(gh + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's atomics have been performed at the L2
    __syncthreads();
    // relaxed on purpose: everything the last workgroup reads arrives through atomics performed at the device's point of coherence
    // (drained above by vmcnt(0)); an agent-scope acquire / release here would write back and invalidate this XCD's L2 once per
    // workgroup (measured: 8 x 4K frames 88 us against 53 us for the three-kernel path)
    if (t == 0) s_last = __hip_atomic_fetch_add(cnt + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;                                            // uniform over the workgroup
    const uint32_t c = __hip_atomic_exchange(gh + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // RMW at the L2: coherent by construction
    if (t == 0) __hip_atomic_store(cnt + f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lut_out[(size_t)f * 256 + t] = equalize_lut_value(c, total, &sh);
}

// ---------------------------------------------------------------------------------------------
// K3  LUT apply (+ fused NV12 UV fill/copy)  (SURVEY 8a rows A4, A7).  grid = (B, n_frames).
// Reads W*H, writes W*H (plus UV: writes W*H/2, reads W*H/2 when copying).  Bound: HBM.
// LDS: lut[value][32] replicated -> conflict-free ds_read per pixel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lut_dword(const uint32_t* lut, uint32_t w, uint32_t copy)
{
    const uint32_t a = lut[((w & 0xffu) << kCopyShift) + copy];
    const uint32_t b = lut[(((w >> 8) & 0xffu) << kCopyShift) + copy];
    const uint32_t c = lut[(((w >> 16) & 0xffu) << kCopyShift) + copy];
    const uint32_t d = lut[((w >> 24) << kCopyShift) + copy];
    return a | (b << 8) | (c << 16) | (d << 24);
}

__device__ __forceinline__ u32x4 lut_vec(const uint32_t* lut, u32x4 q, uint32_t copy)
{
    u32x4 r;
    r.x = lut_dword(lut, q.x, copy); r.y = lut_dword(lut, q.y, copy);
    r.z = lut_dword(lut, q.z, copy); r.w = lut_dword(lut, q.w, copy);
    return r;
}

// dst[i] = lut[src[i]] for i in [0,n), vector body aligned on dst (src loads may be unaligned).
__device__ __forceinline__ void lut_flat(const uint32_t* lut, const uint8_t* src, uint8_t* dst, long long n, int part, int nparts)
{
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const Split16 s = split16(dst, n);
    if (part == 0 && t < s.head) dst[t] = (uint8_t)lut[((uint32_t)src[t] << kCopyShift) + copy];
    if (part == nparts - 1 && t < s.tail) {
        const long long o = s.head + (s.nvec << 4) + t;
        dst[o] = (uint8_t)lut[((uint32_t)src[o] << kCopyShift) + copy];
    }
    const long long v0 = s.nvec * part / nparts, v1 = s.nvec * (part + 1) / nparts;
    const u32x4_u* sp = reinterpret_cast<const u32x4_u*>(src + s.head);
    u32x4* dp = reinterpret_cast<u32x4*>(dst + s.head);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (long long i = v0 + t; i < v1; i += 4 * kThreads) {       // 4 x 16 B in flight per lane, each predicated on its own bound
        u32x4 q[4]; bool qv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { qv[k] = i + (long long)k * kThreads < v1; q[k] = qv[k] ? sp[i + (long long)k * kThreads] : zero; }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (qv[k]) dp[i + (long long)k * kThreads] = lut_vec(lut, q[k], copy);
    }
}

// dst = lut[src] over a band of strided rows whose DESTINATION pitch is a multiple of 16 (see hist_rows): aligned vector stores,
// source loads at whatever alignment the source has.
__device__ __forceinline__ void lut_rows(const uint32_t* lut, const uint8_t* src0, long long src_step, uint8_t* dst0, long long dst_step,
                                         long long row_bytes, int nrows)
{
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const RowBand rb(dst0, row_bytes);
    if (rb.slots > 0) {
        const long long items = (long long)nrows * rb.slots;
        int row = t / rb.slots, slot = t - row * rb.slots;
        const int drow = kThreads / rb.slots, dslot = kThreads - drow * rb.slots;
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (long long it = t; it < items; it += 4 * kThreads) {
            u32x4 q[4]; bool qv[4]; long long so[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qv[k] = it + (long long)k * kThreads < items;
                so[k] = (long long)row * dst_step + rb.head + (slot << 4);
                const u32x4_u* ptr = reinterpret_cast<const u32x4_u*>(src0 + (long long)row * src_step + rb.head + (slot << 4));
                q[k] = qv[k] ? *ptr : zero;
                row += drow; slot += dslot;
                if (slot >= rb.slots) { slot -= rb.slots; ++row; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (qv[k]) *reinterpret_cast<u32x4*>(dst0 + so[k]) = lut_vec(lut, q[k], copy);
        }
    }
    const int e = rb.head + rb.tail;
    if (e > 0) {
        const long long items = (long long)nrows * e;
        for (long long it = t; it < items; it += kThreads) {
            const int row = (int)(it / e), c = (int)(it - (long long)row * e);
            const long long off = c < rb.head ? c : row_bytes - rb.tail + (c - rb.head);
            dst0[(long long)row * dst_step + off] = (uint8_t)lut[((uint32_t)src0[(long long)row * src_step + off] << kCopyShift) + copy];
        }
    }
}

// UV plane: fill with 128 or copy, dst aligned stores.
__device__ __forceinline__ void uv_flat(const uint8_t* src, uint8_t* dst, long long n, int mode, int part, int nparts)
{
    const int t = threadIdx.x;
    const Split16 s = split16(dst, n);
    if (part == 0 && t < s.head) dst[t] = mode ? src[t] : (uint8_t)128;
    if (part == nparts - 1 && t < s.tail) {
        const long long o = s.head + (s.nvec << 4) + t;
        dst[o] = mode ? src[o] : (uint8_t)128;
    }
    const long long v0 = s.nvec * part / nparts, v1 = s.nvec * (part + 1) / nparts;
    u32x4* dp = reinterpret_cast<u32x4*>(dst + s.head);
    if (mode == 0) {
        const u32x4 g = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
        for (long long i = v0 + t; i < v1; i += kThreads) dp[i] = g;
    } else {
        const u32x4_u* sp = reinterpret_cast<const u32x4_u*>(src + s.head);
        long long i = v0 + t;
        for (; i + 3 * kThreads < v1; i += 4 * kThreads) {
            const u32x4 a = sp[i], b = sp[i + kThreads], c = sp[i + 2 * kThreads], d = sp[i + 3 * kThreads];
            dp[i] = a; dp[i + kThreads] = b; dp[i + 2 * kThreads] = c; dp[i + 3 * kThreads] = d;
        }
        for (; i < v1; i += kThreads) dp[i] = sp[i];
    }
}

__global__ __launch_bounds__(kThreads) void lut_apply_kernel(PlaneBatch p, const uint8_t* __restrict__ luts, UVJob uv)
{
    __shared__ uint32_t lut[256 * kCopies];
    // frames last-to-first: the histogram pass streamed the batch first-to-last, its tail is still in the Infinity Cache
    const int t = threadIdx.x, f = (int)gridDim.y - 1 - (int)blockIdx.y;
    {
        const uint32_t v = luts[(size_t)f * 256 + t];
#pragma unroll
        for (int k = 0; k < kCopies; ++k) lut[(t << kCopyShift) + ((k + t) & (kCopies - 1))] = v;
    }
    __syncthreads();
    const uint8_t* src = p.src + (long long)f * p.src_frame;
    uint8_t* dst = p.dst + (long long)f * p.dst_frame;
    if (p.rows == 1) {
        lut_flat(lut, src, dst, p.row_bytes, blockIdx.x, gridDim.x);
    } else if ((p.dst_step & 15) == 0) {
        const int r0 = (int)((long long)p.rows * blockIdx.x / gridDim.x), r1 = (int)((long long)p.rows * (blockIdx.x + 1) / gridDim.x);
        lut_rows(lut, src + (long long)r0 * p.src_step, p.src_step, dst + (long long)r0 * p.dst_step, p.dst_step, p.row_bytes, r1 - r0);
    } else {
        for (int r = blockIdx.x; r < p.rows; r += gridDim.x)
            lut_flat(lut, src + (long long)r * p.src_step, dst + (long long)r * p.dst_step, p.row_bytes, 0, 1);
    }
    if (uv.bytes > 0)
        uv_flat(uv.src + (long long)f * uv.src_frame, uv.dst + (long long)f * uv.dst_frame, uv.bytes, uv.mode, blockIdx.x, gridDim.x);
}


}  // namespace mi
