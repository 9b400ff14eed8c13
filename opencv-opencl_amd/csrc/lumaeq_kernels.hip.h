// lumaeq_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels for the luma-equalization path.
//
// What they compute is fixed by OpenCV 4.4's cv::equalizeHist / CLAHE::apply as the reference calls
// them (OpenCVequalHist.cpp:145, clahevideo.cpp:195); how they compute it is MI355X-first:
//   * everything is byte/LUT work bound by HBM (no MFMA): 16 B/lane coalesced loads and stores,
//     grid sized to ~8 workgroups/CU, a batch of frames per launch (grid.y / grid.z = frame);
//   * histograms are privatised in LDS as hist[bin][32]: the copy a lane uses is (lane & 31), so
//     its LDS bank is fixed by the lane and a ds_add_u32 wave-instruction is bank-conflict free
//     whatever the pixel values are (a constant frame costs the same as noise);
//   * the equalizeHist LUT is replicated the same way (lut[value][32]) so the per-pixel gather is
//     conflict free as well;
//   * float steps (LUT scale, CLAHE blend) use explicit __fmul_rn/__fadd_rn/__fsub_rn/__fdiv_rn so
//     no FMA contraction can change a rounding (the file is also built with -ffp-contract=off).
// Kernels never synchronise between workgroups inside a launch; stage results cross kernel
// boundaries only (partials -> LUT -> apply).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Histogram of the bytes [p, p+n) shared between `nparts` workgroups; this one is `part`.
__device__ __forceinline__ void hist_flat(uint32_t* h, const uint8_t* p, long long n, int part, int nparts)
{
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const Split16 s = split16(p, n);
    if (part == 0 && t < s.head) lds_inc(h, ((uint32_t)p[t] << kCopyShift) + copy);
    if (part == nparts - 1 && t < s.tail) lds_inc(h, ((uint32_t)p[s.head + (s.nvec << 4) + t] << kCopyShift) + copy);
    const long long v0 = s.nvec * part / nparts, v1 = s.nvec * (part + 1) / nparts;
    const u32x4* vp = reinterpret_cast<const u32x4*>(p + s.head);
    long long i = v0 + t;
    for (; i + 3 * kThreads < v1; i += 4 * kThreads) {      // 4 x 16 B in flight per lane
        const u32x4 a = vp[i], b = vp[i + kThreads], c = vp[i + 2 * kThreads], d = vp[i + 3 * kThreads];
        hist_add_vec(h, a, copy); hist_add_vec(h, b, copy); hist_add_vec(h, c, copy); hist_add_vec(h, d, copy);
    }
    for (; i < v1; i += kThreads) hist_add_vec(h, vp[i], copy);
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

// ---------------------------------------------------------------------------------------------
// K1  histogram partials.  grid = (B, n_frames); partial[f][b][256].
// Reads W*H bytes per frame once; writes B KiB per frame.  Bound: HBM read.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void hist_partial_kernel(PlaneBatch p, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[256 * kCopies];
    lds_hist_zero(h);
    const uint8_t* base = p.src + (long long)blockIdx.y * p.src_frame;
    if (p.rows == 1) {
        hist_flat(h, base, p.row_bytes, blockIdx.x, gridDim.x);
    } else {
        for (int r = blockIdx.x; r < p.rows; r += gridDim.x) hist_flat(h, base + (long long)r * p.src_step, p.row_bytes, 0, 1);
    }
    __syncthreads();
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = lds_hist_bin(h, threadIdx.x);
}

// Block-wide helpers for 256 threads = 4 waves ------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
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
    long long i = v0 + t;
    for (; i + 3 * kThreads < v1; i += 4 * kThreads) {
        const u32x4 a = sp[i], b = sp[i + kThreads], c = sp[i + 2 * kThreads], d = sp[i + 3 * kThreads];
        dp[i] = lut_vec(lut, a, copy);
        dp[i + kThreads] = lut_vec(lut, b, copy);
        dp[i + 2 * kThreads] = lut_vec(lut, c, copy);
        dp[i + 3 * kThreads] = lut_vec(lut, d, copy);
    }
    for (; i < v1; i += kThreads) dp[i] = lut_vec(lut, sp[i], copy);
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
    const int t = threadIdx.x, f = blockIdx.y;
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
    } else {
        for (int r = blockIdx.x; r < p.rows; r += gridDim.x)
            lut_flat(lut, src + (long long)r * p.src_step, dst + (long long)r * p.dst_step, p.row_bytes, 0, 1);
    }
    if (uv.bytes > 0)
        uv_flat(uv.src + (long long)f * uv.src_frame, uv.dst + (long long)f * uv.dst_frame, uv.bytes, uv.mode, blockIdx.x, gridDim.x);
}


// =============================================================================================
// KF  fused single-read equalizeHist (+ NV12 UV): histogram, CDF/LUT and LUT apply in ONE launch,
// the Y plane read from HBM once.  (SURVEY 8a rows A2+A3+A4+A7.)
//
// MI355X-first design: a 4K Y plane (8.3 MB) does not fit a CU, but it fits the chip: the frame is
// cut into 64 KiB slices, a workgroup keeps its slice in REGISTERS (256 threads x 16 x 16 B) from the
// histogram pass to the apply pass, and the ~127 workgroups holding one frame's slices meet once:
//   1. ticket = atomicAdd(work) -- persistent workgroups take (frame, slice) tickets in order, so the
//      slices of the oldest unfinished frame are always held by running workgroups (no deadlock for
//      any dispatch order as long as >= T workgroups are co-resident; the host guarantees T <= CUs/2);
//   2. slice histogram in LDS (bank-replicated, as K1) -> non-zero bins added to ghist[frame] with
//      agent-scope atomics; every wave waits vmcnt(0); one lane takes an arrival number;
//   3. the LAST arriver exchanges the 256 counts out (returning atomics: coherent by construction),
//      checks sum == W*H (an exact integrity test of the hand-off; retried, bounded), computes the LUT,
//      publishes it with write-through (sc1) stores + checksum, then sets ready[frame];
//   4. the others poll ready[frame] from one lane (relaxed sc1 loads + s_sleep), then ONE agent acquire,
//      then load the 256-byte LUT with sc1 loads and verify the checksum (cdna_hip_programming.md
//      Guideline 16: release on the producer side is replaced by write-through stores drained with
//      vmcnt(0); the consumer keeps the acquire);
//   5. everybody applies the LUT to its registers and streams the result out.
// UV planes are extra tickets (pure fill / copy).  HBM traffic per NV12 frame: read W*H, write
// W*H (+ UV) instead of reading W*H twice.  Every spin is bounded (s_memrealtime): on a timeout the
// workgroup sets *status and leaves, so the grid always drains.
// =============================================================================================
constexpr int kVPT = 20;                            // default: 16-byte vectors a thread keeps in registers (80 KiB slices)
constexpr int kLutPubWords = 128;                   // per frame: 64 LUT dwords + checksum, padded to 512 B
constexpr int kFlagStride = 32;                     // one 128-B line per frame flag / counter

struct FusedJob {
    const uint8_t* src; uint8_t* dst;               // Y plane of frame 0 (16-B aligned)
    long long src_frame, dst_frame;                 // bytes between frames (multiples of 16)
    long long nvec;                                 // W*H / 16 (exact)
    int total;                                      // W*H
    int n_frames;
    int T, U;                                       // Y tickets / UV tickets per frame
    int acquire;                                    // 1: consumers issue an agent acquire before reading the LUT
    int fault_inject;                               // test hook: the last arriver of frame 0 never publishes its LUT
    unsigned long long timeout_ticks;               // bound of every wait, in 100 MHz ticks
    UVJob uv;
    // Hand-off block.  Zeroed once when allocated; every launch leaves it clean again: the ticket counter only
    // grows (work_base = its value at launch), the last arriver of a frame drains ghist (exchange) and resets cnt,
    // ready/lutpub are stamped with a per-launch epoch.  No memset node per call.
    unsigned long long* work;                       // ticket dispenser (monotonic)
    unsigned long long work_base;
    uint32_t epoch;                                 // != 0, different for every launch of a context
    uint32_t* ghist;                                // [cap][256]
    uint32_t* cnt;                                  // [cap][kFlagStride]
    uint32_t* ready;                                // [cap][kFlagStride]
    uint32_t* lutpub;                               // [cap][kLutPubWords]
    uint32_t* status;                               // [0] != 0: a bounded wait expired (result invalid; sticky)
};

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Opaque identity: stops LICM/CSE from keeping hundreds of derived values (LDS addresses, extracted pixel
// bytes) alive across the phases of the persistent loop -- without it the kernel spills ~200 VGPRs.
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void launder(u32x4& v) { asm volatile("" : "+v"(v)); }

struct FusedShared {
    EqLutShared eq;
    unsigned long long ticket;
    uint32_t lut_words[64];
    uint32_t red[4];
    int last, ok, timeout;
};

// Zeroes the hand-off block (a plain kernel instead of hipMemsetAsync: it is captured into HIP graphs like any
// other launch; a memset node did not re-run on graph replay in testing).
__global__ __launch_bounds__(kThreads) void zero_words_kernel(uint32_t* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) p[i] = 0;
}

template <int VPT>
__global__ __launch_bounds__(kThreads, 4) void equalize_fused_kernel(FusedJob j)
{
    constexpr int kSliceVecs = kThreads * VPT;
    __shared__ uint32_t lds[256 * kCopies];          // slice histogram, then the replicated LUT
    __shared__ FusedShared sh;
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const unsigned long long P = (unsigned long long)(j.T + j.U);
    const unsigned long long total_tickets = P * (unsigned long long)j.n_frames;
    for (;;) {
        __syncthreads();
        if (t == 0) sh.ticket = __hip_atomic_fetch_add(j.work, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - j.work_base;
        __syncthreads();
        unsigned long long k = sh.ticket;                            // make it provably wave-uniform (SGPRs): all the
        k = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(k >> 32)) << 32) |   // per-ticket address math then
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)k);                              // stays scalar (guide T20)
        if (k >= total_tickets) break;
        const int f = (int)(k / P);
        const int r = (int)(k - (unsigned long long)f * P);
        if (r >= j.T) {                               // UV ticket (A7): 64 KiB of plain fill / copy
            uv_flat(j.uv.src ? j.uv.src + (long long)f * j.uv.src_frame : nullptr, j.uv.dst + (long long)f * j.uv.dst_frame,
                    j.uv.bytes, j.uv.mode, r - j.T, j.U);
            continue;
        }
        // ---- 1. slice -> registers (loads issued first, LDS zeroing overlaps their latency)
        const long long v0 = (long long)r * kSliceVecs;
        const long long rem = j.nvec - v0;            // vectors of this slice that exist (> 0)
        const int rem32 = (int)(rem < (long long)kSliceVecs ? rem : (long long)kSliceVecs);
        // buffer descriptors over exactly this slice: 32-bit lane offset + scalar offset, and the hardware range
        // check drops the lanes beyond a short last slice (loads return 0, stores are discarded) -- no predicates
        const auto srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(j.src + (long long)f * j.src_frame + v0 * 16), 0, rem32 * 16, 0x00020000);
        const auto drsrc = __builtin_amdgcn_make_buffer_rsrc(j.dst + (long long)f * j.dst_frame + v0 * 16, 0, rem32 * 16, 0x00020000);
        const int toff = t * 16;
        u32x4 q[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) q[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, toff, i * (kThreads * 16), 0);
        for (int i = t; i < 256 * kCopies; i += kThreads) lds[i] = 0;
        __syncthreads();
        // ---- 2. slice histogram, publish with agent-scope atomics
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            if (i * kThreads + t < rem32) hist_add_vec(lds, q[i], copy);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        {
            const uint32_t c = lds_hist_bin(lds, launder(t));
            if (c) __hip_atomic_fetch_add(j.ghist + (size_t)f * 256 + t, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's atomics have been performed
        __syncthreads();
        if (t == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(j.cnt + (size_t)f * kFlagStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.last = (arrived == (uint32_t)(j.T - 1));
            sh.ok = 1;
        }
        __syncthreads();
        uint8_t my_lut;
        if (sh.last && j.fault_inject && f == 0) break;              // test hook: simulate a lost producer (others must time out)
        if (sh.last) {
            // ---- 3. last arriver: collect, verify, compute and publish the LUT
            uint32_t h = 0;
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                h += __hip_atomic_exchange(j.ghist + (size_t)f * 256 + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t ws = wave_sum(h);
                __syncthreads();
                if ((t & 63) == 0) sh.red[t >> 6] = ws;
                if (t == 0) sh.timeout = (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks);   // one decision for the block
                __syncthreads();
                if (sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3] == (uint32_t)j.total) break;
                if (sh.timeout) {
                    if (t == 0) { sh.ok = 0; st_agent(j.status, 1u); }
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            __syncthreads();
            if (!sh.ok) break;
            my_lut = equalize_lut_value(h, j.total, &sh.eq);
            reinterpret_cast<uint8_t*>(sh.lut_words)[t] = my_lut;
            __syncthreads();
            if (t < 64) {
                const uint32_t w = sh.lut_words[t];
                uint32_t* pub = j.lutpub + (size_t)f * kLutPubWords;
                st_agent(pub + t, w);
                const uint32_t sum = wave_sum(w) + 0x5EED0001u + j.epoch;
                if (t == 0) {
                    st_agent(pub + 64, sum);
                    st_agent(j.cnt + (size_t)f * kFlagStride, 0u);  // all T arrivals are in: leave the counter clean for the next launch
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // write-through stores have left this CU
                if (t == 0) st_agent(j.ready + (size_t)f * kFlagStride, j.epoch);
            }
        } else {
            // ---- 4. wait for the frame's LUT
            if (t == 0) {
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                const uint32_t* flag = j.ready + (size_t)f * kFlagStride;
                while (ld_agent(flag) != j.epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks || ld_agent(j.status) != 0u) { sh.ok = 0; st_agent(j.status, 1u); break; }
                }
                if (j.acquire) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            if (!sh.ok) break;
            if (t < 64) {
                const uint32_t* pub = j.lutpub + (size_t)f * kLutPubWords;
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const uint32_t w = ld_agent(pub + t);
                    const uint32_t want = ld_agent(pub + 64);
                    if (wave_sum(w) + 0x5EED0001u + j.epoch == want) { sh.lut_words[t] = w; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t_start > j.timeout_ticks) { if (t == 0) { sh.ok = 0; st_agent(j.status, 2u); } break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
            if (!sh.ok) break;
            my_lut = reinterpret_cast<const uint8_t*>(sh.lut_words)[t];
        }
        // ---- 5. replicated LUT in LDS, apply to the registers, stream out
        __syncthreads();                                            // everyone is done with the histogram in lds[]
        {
            const uint32_t v = my_lut;
            const int tl = launder(t);
#pragma unroll
            for (int c = 0; c < kCopies; ++c) lds[(tl << kCopyShift) + ((c + tl) & (kCopies - 1))] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            launder(q[i]);                                          // re-extract the bytes here instead of keeping 256 of them live
            __builtin_amdgcn_raw_buffer_store_b128(lut_vec(lds, q[i], copy), drsrc, toff, i * (kThreads * 16), 0);
            __builtin_amdgcn_sched_barrier(0);                      // keep the bodies apart: the slice already owns 4*VPT VGPRs
        }
    }
}

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

// ---------------------------------------------------------------------------------------------
// K4  per-tile histogram partials.  grid = (S, tiles, n_frames); partial[f][tile][s][256].
// The padded image is never materialised: rows/columns beyond the frame are read by index
// reflection.  Work items are (row, 16-byte slot) pairs walked incrementally so short tile rows
// (480 B at 4K 8x8) still give every lane a vector load.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void tile_hist_kernel(const uint8_t* __restrict__ src_base, long long step, long long frame_stride,
                                                            ClaheGeom g, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[256 * kCopies];
    lds_hist_zero(h);
    const int t = threadIdx.x;
    const uint32_t copy = t & (kCopies - 1);
    const int S = gridDim.x, s = blockIdx.x, tile = blockIdx.y, f = blockIdx.z;
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
        const int drow = kThreads / slots, dslot = kThreads - drow * slots;
        auto item_ptr = [&]() -> const u32x4_u* {             // address of the current (row, slot), then advance by 256 items
            const int y = reflect101(ty * g.tile_h + r0 + row, g.height);
            const u32x4_u* p = reinterpret_cast<const u32x4_u*>(src + (long long)y * step + x0 + (slot << 4));
            row += drow; slot += dslot;
            if (slot >= slots) { slot -= slots; ++row; }
            return p;
        };
        long long it = t;
        for (; it + 3 * kThreads < items; it += 4 * kThreads) {          // 4 x 16 B in flight per lane
            const u32x4_u* p0 = item_ptr(); const u32x4_u* p1 = item_ptr(); const u32x4_u* p2 = item_ptr(); const u32x4_u* p3 = item_ptr();
            const u32x4 a = *p0, b = *p1, c = *p2, d = *p3;
            hist_add_vec(h, a, copy); hist_add_vec(h, b, copy); hist_add_vec(h, c, copy); hist_add_vec(h, d, copy);
        }
        for (; it < items; it += kThreads) hist_add_vec(h, *item_ptr(), copy);
    }
    if ((in_w & 15) != 0) {                                     // ragged right edge of the in-frame part: byte loads
        const int pw = in_w & 15, xs = x0 + (slots << 4);
        const long long items = (long long)(r1 - r0) * pw;
        for (long long it = t; it < items; it += kThreads) {
            const int row = (int)(it / pw), c = (int)(it - (long long)row * pw);
            const int y = reflect101(ty * g.tile_h + r0 + row, g.height);
            lds_inc(h, ((uint32_t)src[(long long)y * step + xs + c] << kCopyShift) + copy);
        }
    }
    if (in_w < g.tile_w) {                                      // reflected columns (right border tiles only)
        const int pw = g.tile_w - in_w;
        const long long items = (long long)(r1 - r0) * pw;
        for (long long it = t; it < items; it += kThreads) {
            const int row = (int)(it / pw), c = (int)(it - (long long)row * pw);
            const int y = reflect101(ty * g.tile_h + r0 + row, g.height);
            const int x = reflect101(x0 + in_w + c, g.width);
            lds_inc(h, ((uint32_t)src[(long long)y * step + x] << kCopyShift) + copy);
        }
    }
    __syncthreads();
    partial[(((size_t)f * gridDim.y + tile) * S + s) * 256 + t] = lds_hist_bin(h, t);
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
    luts[tile_id * 256 + t] = (uint8_t)r;
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

// res = (a*xa1 + b*xa)*ya1 + (c*xa1 + d*xa)*ya, nine individually rounded f32 ops, then round half to even.
__device__ __forceinline__ float clahe_blend(uint32_t q, float xa, float xa1, float ya, float ya1)
{
    const float a = (float)(q & 0xffu), b = (float)((q >> 8) & 0xffu), c = (float)((q >> 16) & 0xffu), d = (float)(q >> 24);
    const float top = __fmul_rn(__fadd_rn(__fmul_rn(a, xa1), __fmul_rn(b, xa)), ya1);
    const float bot = __fmul_rn(__fadd_rn(__fmul_rn(c, xa1), __fmul_rn(d, xa)), ya);
    return rintf(__fadd_rn(top, bot));                               // v_rndne_f32: cvRound
}
__device__ __forceinline__ uint32_t clahe_px(uint32_t q, float xa, float xa1, float ya, float ya1)
{
    int r = (int)clahe_blend(q, xa, xa1, ya, ya1);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);                             // saturate_cast<uchar>
    return (uint32_t)r;
}
// 16 pixels of one row: one ds_read_b32 per pixel, v_cvt_pk_u8_f32 (saturating, input already integral) packs the bytes
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
            acc = __builtin_amdgcn_cvt_pk_u8_f32(clahe_blend(quad[poff[j] + v], xa[j], xa1[j], ya, ya1), b, acc);
        }
        ow[k] = acc;
    }
    u32x4 o; o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
    return o;
}

// Float-table variant of the 16-pixel body: the LDS entry is {a, c, b, d} as f32, so one ds_read_b128 delivers
// two register pairs that feed v_pk_mul_f32 / v_pk_add_f32 directly (each lane of a packed op is an ordinary
// individually rounded f32 op): 4 packed ops + 1 add per pixel, no byte->float converts.
__device__ __forceinline__ u32x4 clahe_vec16_f32(const f32x4* quadf, u32x4 q, const int* poff, const float* xa, const float* xa1, float ya, float ya1)
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
            const f32x2 x1 = {xa1[j], xa1[j]}, x0v = {xa[j], xa[j]};
            tb[b] = (ac * x1 + bd * x0v) * yv;                   // -ffp-contract=off: pk_mul, pk_mul, pk_add, pk_mul
        }
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc = __builtin_amdgcn_cvt_pk_u8_f32(rintf(__fadd_rn(tb[b].x, tb[b].y)), b, acc);
        ow[k] = acc;
    }
    u32x4 o; o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
    return o;
}

template <bool FT>
__global__ __launch_bounds__(kThreads) void clahe_interp_kernel(PlaneBatch p, ClaheGeom g, const uint8_t* __restrict__ luts,
                                                               int subs, int groups, UVJob uv)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t quad[];   // [(tiles_x + 1)][256] u32 quads, or f32x4 when FT
    f32x4* quadf = reinterpret_cast<f32x4*>(quad);
    const int t = threadIdx.x, f = blockIdx.y;
    const int band = blockIdx.x / subs, sub = blockIdx.x - band * subs;
    const int ty1u = band - 1;                                // unclamped ty1 of every row of the band
    const int ty1 = max(ty1u, 0), ty2 = min(ty1u + 1, g.tiles_y - 1);
    const uint8_t* lf = luts + (size_t)f * g.tiles_x * g.tiles_y * 256;
    const uint8_t* l1 = lf + (size_t)ty1 * g.tiles_x * 256;
    const uint8_t* l2 = lf + (size_t)ty2 * g.tiles_x * 256;
    const int npairs = g.tiles_x + 1;
    for (int i = t; i < npairs * 256; i += kThreads) {
        const int pr = i >> 8, v = i & 255;
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
        int poff[kInterpPx];
#pragma unroll
        for (int j = 0; j < kInterpPx; ++j) {
            const float txf = __fsub_rn(__fmul_rn((float)(x0 + j), g.inv_tw), 0.5f);
            const int tx1 = floor_f32_to_int(txf);
            xa[j] = __fsub_rn(txf, (float)tx1);
            xa1[j] = __fsub_rn(1.0f, xa[j]);
            int pr = tx1 + 1;                                  // pair index; columns beyond the frame are never used
            pr = pr < 0 ? 0 : (pr > g.tiles_x ? g.tiles_x : pr);
            poff[j] = pr << 8;
        }
        const uint8_t* src = p.src + (long long)f * p.src_frame;
        uint8_t* dst = p.dst + (long long)f * p.dst_frame;
        const bool full = x0 + kInterpPx <= g.width;
        // ty1 is monotone in y: trim the widened range to the rows that really belong to this band, using the
        // reference's own float expression (at most kBandMargin+1 steps per end)
        auto ty1_of = [&](int y) { return floor_f32_to_int(__fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f)); };
        int ya_lo = y_lo, ya_hi = y_hi;
        while (ya_lo < ya_hi && ty1_of(ya_lo) != ty1u) ++ya_lo;
        while (ya_hi > ya_lo && ty1_of(ya_hi - 1) != ty1u) --ya_hi;
        // rows of this lane: ya_lo + phase, + phases, ...  (sub-ranges are contiguous per block, phases interleave inside)
        int y = ya_lo + ((phase - (ya_lo - y_lo) % phases) % phases + phases) % phases;
        if (full) {
            // The loop is VALU-issue bound (~290 instructions per 16 pixels: 64 byte->float converts, 144 blend
            // flops, 16 LDS reads); an explicit 2-row software pipeline measured 11 % SLOWER than letting the
            // other resident waves cover the load latency, so the row loop stays simple.
            for (; y < ya_hi; y += phases) {
                const u32x4 q = *reinterpret_cast<const u32x4_u*>(src + (long long)y * p.src_step + x0);
                const float tyf = __fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f);
                const float ya = __fsub_rn(tyf, (float)ty1u), ya1 = __fsub_rn(1.0f, ya);
                *reinterpret_cast<u32x4_u*>(dst + (long long)y * p.dst_step + x0) =
                    FT ? clahe_vec16_f32(quadf, q, poff, xa, xa1, ya, ya1) : clahe_vec16(quad, q, poff, xa, xa1, ya, ya1);
            }
        } else {
            for (; y < ya_hi; y += phases) {
                const float tyf = __fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f);
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
                        dr[j] = (uint8_t)clahe_px(e, xa[j], xa1[j], ya, ya1);
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
    const float txf = __fsub_rn(__fmul_rn((float)x, g.inv_tw), 0.5f);
    int tx1 = floor_f32_to_int(txf);
    const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
    int tx2 = tx1 + 1; tx1 = max(tx1, 0); tx2 = min(tx2, g.tiles_x - 1);
    const float tyf = __fsub_rn(__fmul_rn((float)y, g.inv_th), 0.5f);
    int ty1 = floor_f32_to_int(tyf);
    const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
    int ty2 = ty1 + 1; ty1 = max(ty1, 0); ty2 = min(ty2, g.tiles_y - 1);
    const uint32_t v = p.src[(long long)f * p.src_frame + (long long)y * p.src_step + x];
    const uint32_t q = (uint32_t)lf[((size_t)ty1 * g.tiles_x + tx1) * 256 + v] |
                       ((uint32_t)lf[((size_t)ty1 * g.tiles_x + tx2) * 256 + v] << 8) |
                       ((uint32_t)lf[((size_t)ty2 * g.tiles_x + tx1) * 256 + v] << 16) |
                       ((uint32_t)lf[((size_t)ty2 * g.tiles_x + tx2) * 256 + v] << 24);
    p.dst[(long long)f * p.dst_frame + (long long)y * p.dst_step + x] = (uint8_t)clahe_px(q, xa, xa1, ya, ya1);
}

// UV-only launch (used when the Y kernel cannot carry the UV job).
__global__ __launch_bounds__(kThreads) void uv_kernel(UVJob uv)
{
    const int f = blockIdx.y;
    uv_flat(uv.src + (long long)f * uv.src_frame, uv.dst + (long long)f * uv.dst_frame, uv.bytes, uv.mode, blockIdx.x, gridDim.x);
}


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

// =============================================================================================
// Colour-domain neighbours of the path: cv::cvtColor(COLOR_BGR2YUV / COLOR_YUV2BGR) on CV_8UC3
// and the split / merge around the luma op (singlecolor.cpp:39-66, clahe1frame.cpp:83-102).
// OpenCV 4.4 color_yuv.simd.hpp, 8-bit fixed point (yuv_shift = 14), restated in oracle/color_oracle.c:
//   Y = DESCALE(B*1868 + G*9617 + R*4899), U = DESCALE((B-Y)*8061 + (128<<14)), V = DESCALE((R-Y)*14369 + (128<<14))
//   B = Y + DESCALE((U-128)*33292), G = Y + DESCALE((U-128)*-6472 + (V-128)*-9519), R = Y + DESCALE((V-128)*18678)
// Pure integer work, 3 B/px streams: bound by HBM.  A lane handles 16 pixels = 3 x 16 B of interleaved data.
// =============================================================================================
struct ColorJob {
    const uint8_t* src; uint8_t* dst;        // interleaved CV_8UC3 side (src for MODE 0/1/2, dst for 0/1/3)
    long long src_step, dst_step;            // bytes between rows (interleaved side(s))
    long long src_frame, dst_frame;
    uint8_t* p0; uint8_t* p1; uint8_t* p2;   // planes (MODE 2: outputs Y,U,V; MODE 3: inputs Y,U,V), tightly packed W*H each
    long long plane_frame;                   // bytes between frames of each plane
    long long row_px;                        // pixels per row (contiguous images: W*H with rows == 1)
    int rows;
};

// CV_DESCALE(x, 14).  The empty asm keeps the shifted value opaque: hipcc (ROCm 7.2) otherwise folds pairs of
// "arithmetic shift right -> clamp to [0,255] -> pack" into gfx950's v_ashr_pk_u8_i32, and that lowering produced
// wrong bytes for the V plane here (tools/dbg_color.hip reproduces it: neighbouring bytes get OR-ed together).
__device__ __forceinline__ int yuv_descale(int x) { int r = (x + (1 << 13)) >> 14; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ uint32_t sat_u8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
__device__ __forceinline__ void px_bgr2yuv(uint32_t b, uint32_t g, uint32_t r, uint32_t& Y, uint32_t& U, uint32_t& V)
{
    const int y = yuv_descale((int)b * 1868 + (int)g * 9617 + (int)r * 4899);
    V = sat_u8(yuv_descale(((int)r - y) * 14369 + (128 << 14)));
    U = sat_u8(yuv_descale(((int)b - y) * 8061 + (128 << 14)));
    Y = sat_u8(y);
}
__device__ __forceinline__ void px_yuv2bgr(uint32_t Y, uint32_t U, uint32_t V, uint32_t& b, uint32_t& g, uint32_t& r)
{
    const int u = (int)U - 128, v = (int)V - 128;
    b = sat_u8((int)Y + yuv_descale(u * 33292));
    g = sat_u8((int)Y + yuv_descale(u * -6472 + v * -9519));
    r = sat_u8((int)Y + yuv_descale(v * 18678));
}

// MODE 0: BGR -> YUV interleaved.  1: YUV -> BGR interleaved.  2: BGR -> planes Y,U,V (cvtColor + split).
// 3: planes Y,U,V -> BGR (merge + cvtColor).   grid = (blocks, min(rows, 65535), frames)
template <int MODE>
__global__ __launch_bounds__(kThreads) void color_kernel(ColorJob j)
{
    const int f = blockIdx.z;
    for (int row = blockIdx.y; row < j.rows; row += gridDim.y) {
        const uint8_t* s3 = MODE != 3 ? j.src + (long long)f * j.src_frame + (long long)row * j.src_step : nullptr;
        uint8_t* d3 = MODE != 2 ? j.dst + (long long)f * j.dst_frame + (long long)row * j.dst_step : nullptr;
        const long long poff = (long long)f * j.plane_frame + (long long)row * j.row_px;
        const long long groups = j.row_px >> 4;
        const bool a3s = MODE == 3 || (((uintptr_t)s3 & 15) == 0), a3d = MODE == 2 || (((uintptr_t)d3 & 15) == 0);
        const bool ap = MODE < 2 || ((((uintptr_t)j.p0 | (uintptr_t)j.p1 | (uintptr_t)j.p2 | (uintptr_t)poff) & 15) == 0);
        if (a3s && a3d && ap) {
            for (long long gidx = (long long)blockIdx.x * kThreads + threadIdx.x; gidx < groups; gidx += (long long)gridDim.x * kThreads) {
                uint32_t c0[16], c1[16], c2[16];            // channel values of 16 pixels
                if (MODE != 3) {
                    const u32x4* sp = reinterpret_cast<const u32x4*>(s3 + gidx * 48);
                    const u32x4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
                    const uint32_t w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        c0[p] = (w[(3 * p) >> 2] >> (8 * ((3 * p) & 3))) & 0xffu;
                        c1[p] = (w[(3 * p + 1) >> 2] >> (8 * ((3 * p + 1) & 3))) & 0xffu;
                        c2[p] = (w[(3 * p + 2) >> 2] >> (8 * ((3 * p + 2) & 3))) & 0xffu;
                    }
                } else {
                    const u32x4 y = *reinterpret_cast<const u32x4*>(j.p0 + poff + gidx * 16);
                    const u32x4 u = *reinterpret_cast<const u32x4*>(j.p1 + poff + gidx * 16);
                    const u32x4 v = *reinterpret_cast<const u32x4*>(j.p2 + poff + gidx * 16);
                    const uint32_t wy[4] = {y.x, y.y, y.z, y.w}, wu[4] = {u.x, u.y, u.z, u.w}, wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        c0[p] = (wy[p >> 2] >> (8 * (p & 3))) & 0xffu;
                        c1[p] = (wu[p >> 2] >> (8 * (p & 3))) & 0xffu;
                        c2[p] = (wv[p >> 2] >> (8 * (p & 3))) & 0xffu;
                    }
                }
                uint32_t o0[16], o1[16], o2[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    if (MODE == 0 || MODE == 2) px_bgr2yuv(c0[p], c1[p], c2[p], o0[p], o1[p], o2[p]);
                    else px_yuv2bgr(c0[p], c1[p], c2[p], o0[p], o1[p], o2[p]);
                }
                if (MODE != 2) {
                    uint32_t w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        w[(3 * p) >> 2] |= o0[p] << (8 * ((3 * p) & 3));
                        w[(3 * p + 1) >> 2] |= o1[p] << (8 * ((3 * p + 1) & 3));
                        w[(3 * p + 2) >> 2] |= o2[p] << (8 * ((3 * p + 2) & 3));
                    }
                    u32x4* dp = reinterpret_cast<u32x4*>(d3 + gidx * 48);
                    const u32x4 r0 = {w[0], w[1], w[2], w[3]}, r1 = {w[4], w[5], w[6], w[7]}, r2 = {w[8], w[9], w[10], w[11]};
                    dp[0] = r0; dp[1] = r1; dp[2] = r2;
                } else {
                    uint32_t wy[4], wu[4], wv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        wy[k] = o0[4 * k] | (o0[4 * k + 1] << 8) | (o0[4 * k + 2] << 16) | (o0[4 * k + 3] << 24);
                        wu[k] = o1[4 * k] | (o1[4 * k + 1] << 8) | (o1[4 * k + 2] << 16) | (o1[4 * k + 3] << 24);
                        wv[k] = o2[4 * k] | (o2[4 * k + 1] << 8) | (o2[4 * k + 2] << 16) | (o2[4 * k + 3] << 24);
                    }
                    const u32x4 ry = {wy[0], wy[1], wy[2], wy[3]}, ru = {wu[0], wu[1], wu[2], wu[3]}, rv = {wv[0], wv[1], wv[2], wv[3]};
                    *reinterpret_cast<u32x4*>(j.p0 + poff + gidx * 16) = ry;
                    *reinterpret_cast<u32x4*>(j.p1 + poff + gidx * 16) = ru;
                    *reinterpret_cast<u32x4*>(j.p2 + poff + gidx * 16) = rv;
                }
            }
        }
        // ragged tail of the row (or the whole row when something is not 16-B aligned): one pixel per lane
        const long long first = (a3s && a3d && ap) ? (groups << 4) : 0;
        for (long long x = first + (long long)blockIdx.x * kThreads + threadIdx.x; x < j.row_px; x += (long long)gridDim.x * kThreads) {
            uint32_t a, b, c, o0, o1, o2;
            if (MODE != 3) { a = s3[3 * x]; b = s3[3 * x + 1]; c = s3[3 * x + 2]; }
            else { a = j.p0[poff + x]; b = j.p1[poff + x]; c = j.p2[poff + x]; }
            if (MODE == 0 || MODE == 2) px_bgr2yuv(a, b, c, o0, o1, o2); else px_yuv2bgr(a, b, c, o0, o1, o2);
            if (MODE != 2) { d3[3 * x] = (uint8_t)o0; d3[3 * x + 1] = (uint8_t)o1; d3[3 * x + 2] = (uint8_t)o2; }
            else { j.p0[poff + x] = (uint8_t)o0; j.p1[poff + x] = (uint8_t)o1; j.p2[poff + x] = (uint8_t)o2; }
        }
    }
}

}  // namespace mi
