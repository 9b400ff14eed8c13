// diff.hip.h -- the reference's own device-vs-CPU check as a device operator: cv::absdiff + xf::cv::analyzeDiff
// Part of the gfx950 kernel set of libmi_lumaeq (see ../lumaeq_kernels.hip.h for the design notes).
//
// 1frameMeasure.cpp:91-100 compares the accelerator's plane with cv::equalizeHist's:
//     cv::absdiff(y_ocv, y_fpga, diff);  xf::cv::analyzeDiff(diff, 1, err_per);  pass iff err_per == 0
// analyzeDiff (Vitis Vision, common/xf_sw_utils.hpp) walks the difference image, tracks the smallest and the largest
// difference and counts the pixels whose difference EXCEEDS the threshold; err_per = 100 * count / (rows * cols).
// Here both steps are one pass over the two planes where they already are (device memory), so a full-size batch can be
// checked without a download: grid = (B, n_frames), 256 threads, 16 bytes per lane per step, HBM-bound (2 reads [+ 1 write]).
#pragma once
#include "common.hip.h"

namespace mi {

struct DiffJob {
    const uint8_t* a; const uint8_t* b; uint8_t* diff;       // b == nullptr: `a` already is a difference image; diff optional
    long long a_step, b_step, d_step;                         // bytes between rows
    long long a_frame, b_frame, d_frame;                      // bytes between frames
    long long row_bytes; int rows;                            // rows == 1: contiguous plane, row_bytes = W*H
    int threshold;
};

struct DiffAcc { uint32_t above, mx, mn; };

__device__ __forceinline__ void diff_byte(uint32_t x, uint32_t y, int thr, DiffAcc& acc, uint32_t& d)
{
    d = x > y ? x - y : y - x;
    acc.above += (int)d > thr ? 1u : 0u;
    acc.mx = max(acc.mx, d); acc.mn = min(acc.mn, d);
}

__device__ __forceinline__ uint32_t diff_dword(uint32_t x, uint32_t y, int thr, DiffAcc& acc)
{
    uint32_t d0, d1, d2, d3;
    diff_byte(x & 0xffu, y & 0xffu, thr, acc, d0);
    diff_byte((x >> 8) & 0xffu, (y >> 8) & 0xffu, thr, acc, d1);
    diff_byte((x >> 16) & 0xffu, (y >> 16) & 0xffu, thr, acc, d2);
    diff_byte(x >> 24, y >> 24, thr, acc, d3);
    return d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
}

// bytes [0, n) of one row (or of the whole contiguous plane), shared between `nparts` workgroups
__device__ __forceinline__ void diff_flat(const uint8_t* a, const uint8_t* b, uint8_t* d, long long n, int part, int nparts, int thr, DiffAcc& acc)
{
    const int t = threadIdx.x;
    const Split16 s = split16(a, n);                          // vector body aligned on `a`; b and diff may be unaligned
    auto one = [&](long long o) {
        uint32_t dv;
        diff_byte(a[o], b ? b[o] : 0u, thr, acc, dv);
        if (d) d[o] = (uint8_t)dv;
    };
    if (part == 0 && t < s.head) one(t);
    if (part == nparts - 1 && t < s.tail) one(s.head + (s.nvec << 4) + t);
    const long long v0 = s.nvec * part / nparts, v1 = s.nvec * (part + 1) / nparts;
    const u32x4* ap = reinterpret_cast<const u32x4*>(a + s.head);
    const u32x4_u* bp = reinterpret_cast<const u32x4_u*>(b ? b + s.head : nullptr);
    u32x4_u* dp = reinterpret_cast<u32x4_u*>(d ? d + s.head : nullptr);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (long long i = v0 + t; i < v1; i += 2 * kThreads) {   // two vectors of each plane in flight per lane
        const bool two = i + kThreads < v1;
        const u32x4 x0 = ap[i], y0 = b ? bp[i] : zero;
        const u32x4 x1 = two ? ap[i + kThreads] : zero, y1 = (two && b) ? bp[i + kThreads] : zero;
        u32x4 r;
        r.x = diff_dword(x0.x, y0.x, thr, acc); r.y = diff_dword(x0.y, y0.y, thr, acc);
        r.z = diff_dword(x0.z, y0.z, thr, acc); r.w = diff_dword(x0.w, y0.w, thr, acc);
        if (d) dp[i] = r;
        if (two) {
            r.x = diff_dword(x1.x, y1.x, thr, acc); r.y = diff_dword(x1.y, y1.y, thr, acc);
            r.z = diff_dword(x1.z, y1.z, thr, acc); r.w = diff_dword(x1.w, y1.w, thr, acc);
            if (d) dp[i + kThreads] = r;
        }
    }
}

// stats[f] = {above, max, min, total}; `total` and the identities of the three reductions are written by diff_init_kernel
__global__ __launch_bounds__(kThreads) void diff_init_kernel(uint32_t* __restrict__ stats, int n_frames, uint32_t total)
{
    const int f = blockIdx.x * kThreads + threadIdx.x;
    if (f < n_frames) { stats[4 * f + 0] = 0; stats[4 * f + 1] = 0; stats[4 * f + 2] = total ? 255u : 0u; stats[4 * f + 3] = total; }
}

__global__ __launch_bounds__(kThreads) void analyze_diff_kernel(DiffJob j, uint32_t* __restrict__ stats)
{
    __shared__ uint32_t s_acc[3][kThreads / 64];
    const int f = blockIdx.y, t = threadIdx.x;
    const uint8_t* a = j.a + (long long)f * j.a_frame;
    const uint8_t* b = j.b ? j.b + (long long)f * j.b_frame : nullptr;
    uint8_t* d = j.diff ? j.diff + (long long)f * j.d_frame : nullptr;
    DiffAcc acc{0u, 0u, 255u};
    if (j.rows == 1) {
        diff_flat(a, b, d, j.row_bytes, blockIdx.x, gridDim.x, j.threshold, acc);
    } else {
        for (int r = blockIdx.x; r < j.rows; r += gridDim.x)
            diff_flat(a + (long long)r * j.a_step, b ? b + (long long)r * j.b_step : nullptr, d ? d + (long long)r * j.d_step : nullptr,
                      j.row_bytes, 0, 1, j.threshold, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        acc.above += __shfl_xor(acc.above, o, 64);
        acc.mx = max(acc.mx, (uint32_t)__shfl_xor(acc.mx, o, 64));
        acc.mn = min(acc.mn, (uint32_t)__shfl_xor(acc.mn, o, 64));
    }
    if ((t & 63) == 0) { s_acc[0][t >> 6] = acc.above; s_acc[1][t >> 6] = acc.mx; s_acc[2][t >> 6] = acc.mn; }
    __syncthreads();
    if (t == 0) {
        uint32_t ab = 0, mx = 0, mn = 255u;
        for (int w = 0; w < kThreads / 64; ++w) { ab += s_acc[0][w]; mx = max(mx, s_acc[1][w]); mn = min(mn, s_acc[2][w]); }
        uint32_t* st = stats + 4 * (size_t)f;
        if (ab) atomicAdd(st + 0, ab);
        atomicMax(st + 1, mx);
        atomicMin(st + 2, mn);
    }
}

}  // namespace mi
