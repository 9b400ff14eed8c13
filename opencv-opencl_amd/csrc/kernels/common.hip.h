// common.hip.h -- shared types and device helpers (vector types, batch descriptors, bank-replicated LDS
// histogram primitives, wave/block scans).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// gfx950 ONLY, and not merely as a tuning choice.  The hand-offs between workgroups (hist_lut_kernel, tile_hist12_kernel, the fused
// equalize kernel) order their data with RELAXED agent-scope atomics drained by `s_waitcnt vmcnt(0)`: that is sound where a
// no-return atomic is counted by vmcnt and performed at the L2, the device's point of coherence -- the gfx9 family.  The HIP memory
// model promises neither; gfx10 and later count stores with vscnt, and there a LUT would silently be computed from incomplete
// histograms.  Building the device code for anything else is therefore an error, not a slower build (csrc/Makefile's ARCH is
// overridable).  tests/test_gpu_parity.py::test_equalize_paths_agree runs the last-workgroup path (two_kernel_max_frames=64), the fused and
// the three-kernel path over the same batches against the oracle.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmi_lumaeq's kernels rely on gfx950 (gfx9-family) memory semantics: relaxed agent-scope atomics + s_waitcnt vmcnt(0); build with --offload-arch=gfx950"
#endif

namespace mi {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));   // gfx950 global_load/store_dwordx4 accept any alignment

constexpr int kThreads = 256;        // 4 waves of 64
constexpr int kCopies = 32;          // LDS replication = number of ds_*_b32 banks
constexpr int kCopyShift = 5;

// A batch of strided 8-bit planes. "rows == 1" means the plane is contiguous and row_bytes = W*H.
struct PlaneBatch {
    const uint8_t* src;
    uint8_t* dst;
    long long src_step, dst_step;     // bytes between rows
    long long src_frame, dst_frame;   // bytes between frames
    long long row_bytes;              // bytes per row
    int rows;
};

// Trailing UV job of an NV12 frame fused into the apply launch.
struct UVJob {
    const uint8_t* src;
    uint8_t* dst;
    long long src_frame, dst_frame;
    long long bytes;                  // 0 = none
    int mode;                         // 0 = fill 128, 1 = copy
};

struct Split16 { long long head, nvec, tail; };

__device__ __forceinline__ Split16 split16(const void* p, long long n)
{
    Split16 s;
    s.head = (16 - (long long)((uintptr_t)p & 15)) & 15;
    if (s.head > n) s.head = n;
    s.nvec = (n - s.head) >> 4;
    s.tail = n - s.head - (s.nvec << 4);
    return s;
}

// Build self-test (mi_ctx_create): (1 + 2^-12)^2 - (1 + 2^-11) is 0 when the product is rounded before the add (ties to even)
// and 2^-24 when the compiler fused them into an FMA -- i.e. when this file was built without -ffp-contract=off.
__global__ void contract_probe_kernel(float a, float b, float c, float* out) { *out = __fadd_rn(__fmul_rn(a, b), c); }

__device__ __forceinline__ void lds_inc(uint32_t* h, uint32_t idx)
{
    __hip_atomic_fetch_add(h + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32
}

__device__ __forceinline__ void hist_add_dword(uint32_t* h, uint32_t w, uint32_t copy)
{
    lds_inc(h, ((w & 0xffu) << kCopyShift) + copy);
    lds_inc(h, (((w >> 8) & 0xffu) << kCopyShift) + copy);
    lds_inc(h, (((w >> 16) & 0xffu) << kCopyShift) + copy);
    lds_inc(h, ((w >> 24) << kCopyShift) + copy);
}

__device__ __forceinline__ void hist_add_vec(uint32_t* h, u32x4 q, uint32_t copy)
{
    hist_add_dword(h, q.x, copy);
    hist_add_dword(h, q.y, copy);
    hist_add_dword(h, q.z, copy);
    hist_add_dword(h, q.w, copy);
}

// Histogram of the bytes [p, p+n) shared between `nparts` workgroups of NT threads; this one is `part`.
template <int NT = kThreads>
__device__ __forceinline__ void hist_flat(uint32_t* h, const uint8_t* p, long long n, int part, int nparts)
{
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const Split16 s = split16(p, n);
    if (part == 0 && t < s.head) lds_inc(h, ((uint32_t)p[t] << kCopyShift) + copy);
    if (part == nparts - 1 && t < s.tail) lds_inc(h, ((uint32_t)p[s.head + (s.nvec << 4) + t] << kCopyShift) + copy);
    const long long v0 = s.nvec * part / nparts, v1 = s.nvec * (part + 1) / nparts;
    const u32x4* vp = reinterpret_cast<const u32x4*>(p + s.head);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (long long i = v0 + t; i < v1; i += 4 * NT) {      // 4 x 16 B in flight per lane, each load predicated on its own bound (no serial tail)
        u32x4 q[4]; bool qv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { qv[k] = i + (long long)k * NT < v1; q[k] = qv[k] ? vp[i + (long long)k * NT] : zero; }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (qv[k]) hist_add_vec(h, q[k], copy);
    }
}

__device__ __forceinline__ void lds_hist_zero(uint32_t* h)
{
    for (int i = threadIdx.x; i < 256 * kCopies; i += kThreads) h[i] = 0;
    __syncthreads();
}

// Sum of the 32 copies of bin `t` (skewed so that the 64 lanes of a wave hit 32 different banks).
__device__ __forceinline__ uint32_t lds_hist_bin(const uint32_t* h, int t)
{
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kCopies; ++k) s += h[(t << kCopyShift) + ((k + t) & (kCopies - 1))];
    return s;
}

// Block-wide helpers for 256 threads = 4 waves ------------------------------------------------
// Inclusive scan over the 64 lanes of a wave with DPP moves (no LDS traffic, ~8 VALU instructions instead of six ds_bpermute round
// trips): shifts by 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of rows 0 / 2 added to rows 1 / 3 (row_bcast15) and lane 31
// to rows 2 and 3 (row_bcast31).  Lanes without a source read 0 (old = 0, bound_ctrl off).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast31 -> rows 2, 3
    return (uint32_t)x;
}

// inclusive scan over the 256 threads of the block; *block_total receives the grand total.
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* block_total)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t incl = wave_incl_scan(v);
    __syncthreads();                               // s_wave may be in use by a previous call
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t x = s_wave[k]; if (k < w) off += x; tot += x; }
    if (block_total) *block_total = tot;
    return incl + off;
}

}  // namespace mi
