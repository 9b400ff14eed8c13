// kbench.hip -- A/B microbenchmarks of kernel variants for the luma-equalization path (gfx950).
// Interleaved rounds in ONE process (cdna_hip_programming.md 5.4 rule 24); prints median/min us and
// algorithmic GB/s.  Development tool, not part of the shipped library.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/kbench.hip -o tools/kbench && tools/kbench [frames] [dist]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../opencv-opencl_amd/csrc/lumaeq_kernels.hip.h"

using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---------------------------------------------------------------- variants: read-only ceiling
__global__ __launch_bounds__(256) void read_sum_kernel(const u32x4* __restrict__ p, long long nvec_per_frame, long long frame_vec_stride, uint32_t* out)
{
    const u32x4* vp = p + (long long)blockIdx.y * frame_vec_stride;
    const long long v0 = nvec_per_frame * blockIdx.x / gridDim.x, v1 = nvec_per_frame * (blockIdx.x + 1) / gridDim.x;
    uint32_t acc = 0;
    long long i = v0 + threadIdx.x;
    for (; i + 3 * 256 < v1; i += 4 * 256) {
        const u32x4 a = vp[i], b = vp[i + 256], c = vp[i + 512], d = vp[i + 768];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < v1; i += 256) { const u32x4 a = vp[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// ---------------------------------------------------------------- variants: histogram
// COPIES replicated copies selected by lane (COPIES=32 is the shipped layout); PERWAVE: each wave has
// its own COPIES-replicated histogram (COPIES=1 + PERWAVE = the classic per-wave privatised layout).
template <int COPIES, bool PERWAVE, int UNROLL>
__global__ __launch_bounds__(256) void hist_var_kernel(const u32x4* __restrict__ p, long long nvec_per_frame, long long frame_vec_stride,
                                                      uint32_t* __restrict__ partial)
{
    constexpr int NW = PERWAVE ? 4 : 1;
    __shared__ uint32_t h[256 * COPIES * NW];
    for (int i = threadIdx.x; i < 256 * COPIES * NW; i += 256) h[i] = 0;
    __syncthreads();
    uint32_t* hw = h + (PERWAVE ? (threadIdx.x >> 6) * 256 * COPIES : 0);
    const uint32_t copy = threadIdx.x & (COPIES - 1);
    const u32x4* vp = p + (long long)blockIdx.y * frame_vec_stride;
    const long long v0 = nvec_per_frame * blockIdx.x / gridDim.x, v1 = nvec_per_frame * (blockIdx.x + 1) / gridDim.x;
    auto add4 = [&](uint32_t w) {
        lds_inc(hw, (w & 0xffu) * COPIES + copy);
        lds_inc(hw, ((w >> 8) & 0xffu) * COPIES + copy);
        lds_inc(hw, ((w >> 16) & 0xffu) * COPIES + copy);
        lds_inc(hw, (w >> 24) * COPIES + copy);
    };
    auto addv = [&](u32x4 q) { add4(q.x); add4(q.y); add4(q.z); add4(q.w); };
    long long i = v0 + threadIdx.x;
    if (UNROLL == 4) {
        for (; i + 3 * 256 < v1; i += 4 * 256) {
            const u32x4 a = vp[i], b = vp[i + 256], c = vp[i + 512], d = vp[i + 768];
            addv(a); addv(b); addv(c); addv(d);
        }
    } else if (UNROLL == 2) {
        for (; i + 256 < v1; i += 2 * 256) {
            const u32x4 a = vp[i], b = vp[i + 256];
            addv(a); addv(b);
        }
    }
    for (; i < v1; i += 256) addv(vp[i]);
    __syncthreads();
    uint32_t s = 0;
    for (int k = 0; k < COPIES * NW; ++k) {
        const int w = k / COPIES, c = k % COPIES;
        s += h[w * 256 * COPIES + threadIdx.x * COPIES + ((c + threadIdx.x) & (COPIES - 1))];
    }
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s;
}

// 512-thread variant: 8 waves share one 32-copy histogram (more waves per LDS byte)
__global__ __launch_bounds__(512) void hist_512_kernel(const u32x4* __restrict__ p, long long nvec_per_frame, long long frame_vec_stride,
                                                      uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += 512) h[i] = 0;
    __syncthreads();
    const uint32_t copy = threadIdx.x & 31;
    const u32x4* vp = p + (long long)blockIdx.y * frame_vec_stride;
    const long long v0 = nvec_per_frame * blockIdx.x / gridDim.x, v1 = nvec_per_frame * (blockIdx.x + 1) / gridDim.x;
    long long i = v0 + threadIdx.x;
    for (; i + 512 < v1; i += 2 * 512) {
        const u32x4 a = vp[i], b = vp[i + 512];
        hist_add_vec(h, a, copy); hist_add_vec(h, b, copy);
    }
    for (; i < v1; i += 512) hist_add_vec(h, vp[i], copy);
    __syncthreads();
    if (threadIdx.x < 256) partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = lds_hist_bin(h, threadIdx.x);
}

// pure LDS-atomic rate: no global traffic, `iters` ds_add_u32 per lane on pseudo-random bins
template <int COPIES>
__global__ __launch_bounds__(256) void lds_atomic_rate_kernel(int iters, uint32_t seed, uint32_t* out)
{
    __shared__ uint32_t h[256 * COPIES];
    for (int i = threadIdx.x; i < 256 * COPIES; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t copy = threadIdx.x & (COPIES - 1);
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        lds_inc(h, ((x >> 24) & 0xffu) * COPIES + copy);
        lds_inc(h, ((x >> 16) & 0xffu) * COPIES + copy);
        lds_inc(h, ((x >> 8) & 0xffu) * COPIES + copy);
        lds_inc(h, (x & 0xffu) * COPIES + copy);
    }
    __syncthreads();
    if (h[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

// ---------------------------------------------------------------- variants: LUT apply
// MODE 0: byte LUT in LDS (256 B).  MODE 1: replicated u32 lut[v][32] (shipped).  MODE 2: register LUT + ds_bpermute.
// NT: nontemporal stores.
template <int MODE, bool NT, bool REVERSE>
__global__ __launch_bounds__(256) void apply_var_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long nvec_per_frame,
                                                       long long frame_vec_stride, const uint8_t* __restrict__ luts)
{
    __shared__ uint32_t lut32[MODE == 1 ? 256 * 32 : 64];
    const int t = threadIdx.x;
    const int f = REVERSE ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    uint32_t myword = 0;
    if (MODE == 1) {
        const uint32_t v = luts[(size_t)f * 256 + t];
        for (int k = 0; k < 32; ++k) lut32[(t << 5) + ((k + t) & 31)] = v;
    } else if (MODE == 0) {
        if (t < 64) lut32[t] = reinterpret_cast<const uint32_t*>(luts + (size_t)f * 256)[t];
    } else {
        myword = reinterpret_cast<const uint32_t*>(luts + (size_t)f * 256)[t & 63];
    }
    __syncthreads();
    const uint8_t* lut8 = reinterpret_cast<const uint8_t*>(lut32);
    const uint32_t copy = t & 31;
    auto look = [&](uint32_t v) -> uint32_t {
        if (MODE == 1) return lut32[(v << 5) + copy];
        if (MODE == 0) return lut8[v];
        const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(v & 0xfcu), (int)myword);
        return (w >> ((v & 3u) * 8u)) & 0xffu;
    };
    auto map4 = [&](uint32_t w) -> uint32_t {
        return look(w & 0xffu) | (look((w >> 8) & 0xffu) << 8) | (look((w >> 16) & 0xffu) << 16) | (look(w >> 24) << 24);
    };
    auto mapv = [&](u32x4 q) { u32x4 r; r.x = map4(q.x); r.y = map4(q.y); r.z = map4(q.z); r.w = map4(q.w); return r; };
    auto st = [&](u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    const u32x4* sp = src + (long long)f * frame_vec_stride;
    u32x4* dp = dst + (long long)f * frame_vec_stride;
    const long long v0 = nvec_per_frame * blockIdx.x / gridDim.x, v1 = nvec_per_frame * (blockIdx.x + 1) / gridDim.x;
    long long i = v0 + t;
    for (; i + 3 * 256 < v1; i += 4 * 256) {
        const u32x4 a = sp[i], b = sp[i + 256], c = sp[i + 512], d = sp[i + 768];
        st(dp + i, mapv(a)); st(dp + i + 256, mapv(b)); st(dp + i + 512, mapv(c)); st(dp + i + 768, mapv(d));
    }
    for (; i < v1; i += 256) st(dp + i, mapv(sp[i]));
}

// plain copy with the same grid (write+read ceiling for our access pattern)
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long nvec_per_frame, long long frame_vec_stride)
{
    const u32x4* sp = src + (long long)blockIdx.y * frame_vec_stride;
    u32x4* dp = dst + (long long)blockIdx.y * frame_vec_stride;
    const long long v0 = nvec_per_frame * blockIdx.x / gridDim.x, v1 = nvec_per_frame * (blockIdx.x + 1) / gridDim.x;
    long long i = v0 + threadIdx.x;
    for (; i + 3 * 256 < v1; i += 4 * 256) {
        const u32x4 a = sp[i], b = sp[i + 256], c = sp[i + 512], d = sp[i + 768];
        dp[i] = a; dp[i + 256] = b; dp[i + 512] = c; dp[i + 768] = d;
    }
    for (; i < v1; i += 256) dp[i] = sp[i];
}

// copy Y + fill UV with the same grid: the HBM-traffic mix of the whole NV12 equalize path at minimum traffic
// (read W*H, write 1.5*W*H) -- the ceiling the fused kernel is compared with
__global__ __launch_bounds__(256) void copy_y_fill_uv_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long nvec_y, long long nvec_uv, long long frame_vec_stride)
{
    const u32x4* sp = src + (long long)blockIdx.y * frame_vec_stride;
    u32x4* dp = dst + (long long)blockIdx.y * frame_vec_stride;
    const long long v0 = nvec_y * blockIdx.x / gridDim.x, v1 = nvec_y * (blockIdx.x + 1) / gridDim.x;
    long long i = v0 + threadIdx.x;
    for (; i + 3 * 256 < v1; i += 4 * 256) {
        const u32x4 a = sp[i], b = sp[i + 256], c = sp[i + 512], d = sp[i + 768];
        dp[i] = a; dp[i + 256] = b; dp[i + 512] = c; dp[i + 768] = d;
    }
    for (; i < v1; i += 256) dp[i] = sp[i];
    const long long u0 = nvec_uv * blockIdx.x / gridDim.x, u1 = nvec_uv * (blockIdx.x + 1) / gridDim.x;
    const u32x4 g = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    for (long long k = u0 + threadIdx.x; k < u1; k += 256) dp[nvec_y + k] = g;
}
__global__ __launch_bounds__(256) void fill_kernel(u32x4* __restrict__ dst, long long nvec, long long frame_vec_stride)
{
    u32x4* dp = dst + (long long)blockIdx.y * frame_vec_stride;
    const long long v0 = nvec * blockIdx.x / gridDim.x, v1 = nvec * (blockIdx.x + 1) / gridDim.x;
    const u32x4 g = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    for (long long i = v0 + threadIdx.x; i < v1; i += 256) dp[i] = g;
}

// ---------------------------------------------------------------- harness
struct Variant { std::string name; double alg_bytes; std::function<void(hipStream_t)> run; std::vector<float> us; };

int main(int argc, char** argv)
{
    const int nf = argc > 1 ? atoi(argv[1]) : 64;
    const int dist = argc > 2 ? atoi(argv[2]) : 2;        // 1 uniform, 2 low-contrast, 3 constant
    const int rounds = argc > 3 ? atoi(argv[3]) : 15;
    const int W = 3840, H = 2160;
    const long long ysz = (long long)W * H, fb = ysz + ysz / 2;
    const long long nvec = ysz / 16, fvs = fb / 16;
    uint8_t *d_in, *d_out; uint32_t* d_partial; uint8_t* d_luts; uint32_t* d_dummy;
    CK(hipMalloc(&d_in, fb * nf)); CK(hipMalloc(&d_out, fb * nf));
    CK(hipMalloc(&d_partial, (size_t)nf * 2048 * 1024)); CK(hipMalloc(&d_luts, nf * 256)); CK(hipMalloc(&d_dummy, 4096));
    {
        std::vector<uint8_t> hbuf(fb);
        uint64_t s = 88172645463325252ull;
        for (int f = 0; f < nf; ++f) {
            for (long long i = 0; i < fb; ++i) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                uint8_t v = (uint8_t)(s >> 32);
                if (i < ysz) {
                    if (dist == 2) v = (uint8_t)std::min(200, std::max(16, 96 + (int)((i % W) * 32 / W) + (int)(v % 25) + (int)((s >> 40) % 25) - 24));
                    else if (dist == 3) v = 128;
                }
                hbuf[i] = v;
            }
            CK(hipMemcpy(d_in + f * fb, hbuf.data(), fb, hipMemcpyHostToDevice));
        }
        std::vector<uint8_t> l(nf * 256);
        for (size_t i = 0; i < l.size(); ++i) l[i] = (uint8_t)(255 - (i & 255));
        CK(hipMemcpy(d_luts, l.data(), l.size(), hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    const u32x4* vin = (const u32x4*)d_in; u32x4* vout = (u32x4*)d_out;
    const double yb = (double)ysz * nf;
    std::vector<Variant> vs;
    auto B = [&](int total) { return std::max(1, total / nf); };
    for (int tot : {1024, 2048, 4096}) {
        const int b = B(tot);
        vs.push_back({"read_sum grid=" + std::to_string(b * nf), yb, [=](hipStream_t s) { hipLaunchKernelGGL(read_sum_kernel, dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_dummy); }, {}});
    }
    for (int tot : {1280, 2048, 4096}) {
        const int b = B(tot);
        vs.push_back({"hist c32 u4 grid=" + std::to_string(b * nf), yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<32, false, 4>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
    }
    { const int b = B(2048);
      vs.push_back({"hist c32 u2", yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<32, false, 2>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
      vs.push_back({"hist c32 u1", yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<32, false, 1>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
      vs.push_back({"hist c16 u4", yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<16, false, 4>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
      vs.push_back({"hist c8 u4", yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<8, false, 4>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
      vs.push_back({"hist perwave c1 u4 (classic)", yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<1, true, 4>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
      vs.push_back({"hist perwave c8 u4", yb, [=](hipStream_t s) { hipLaunchKernelGGL((hist_var_kernel<8, true, 4>), dim3(b, nf), dim3(256), 0, s, vin, nvec, fvs, d_partial); }, {}});
      vs.push_back({"hist 512thr c32 u2", yb, [=](hipStream_t s) { hipLaunchKernelGGL(hist_512_kernel, dim3(b, nf), dim3(512), 0, s, vin, nvec, fvs, d_partial); }, {}});
    }
    // pure LDS atomic rate: 1280 blocks x 256 thr x 4*iters atomics ; report "bytes" = atomics (1 px each)
    { const int iters = 4096; const int blocks = 1280;
      const double px = (double)blocks * 256 * 4 * iters;
      vs.push_back({"lds_atomic_rate c32 (no HBM)", px, [=](hipStream_t s) { hipLaunchKernelGGL((lds_atomic_rate_kernel<32>), dim3(blocks), dim3(256), 0, s, iters, 7u, d_dummy); }, {}});
      vs.push_back({"lds_atomic_rate c1  (no HBM)", px, [=](hipStream_t s) { hipLaunchKernelGGL((lds_atomic_rate_kernel<1>), dim3(blocks), dim3(256), 0, s, iters, 7u, d_dummy); }, {}});
    }
    { const int b = B(2048);
      vs.push_back({"copy Y", 2 * yb, [=](hipStream_t s) { hipLaunchKernelGGL(copy_kernel, dim3(b, nf), dim3(256), 0, s, vin, vout, nvec, fvs); }, {}});
      vs.push_back({"copy Y + fill UV (min-traffic mix)", 2.5 * yb, [=](hipStream_t s) { hipLaunchKernelGGL(copy_y_fill_uv_kernel, dim3(b, nf), dim3(256), 0, s, vin, vout, nvec, nvec / 2, fvs); }, {}});
      vs.push_back({"fill whole frame (write only)", 1.5 * yb, [=](hipStream_t s) { hipLaunchKernelGGL(fill_kernel, dim3(b, nf), dim3(256), 0, s, vout, fvs, fvs); }, {}});
      vs.push_back({"apply lut8 (256B LDS)", 2 * yb, [=](hipStream_t s) { hipLaunchKernelGGL((apply_var_kernel<0, false, false>), dim3(b, nf), dim3(256), 0, s, vin, vout, nvec, fvs, d_luts); }, {}});
      vs.push_back({"apply lut32x32 (shipped)", 2 * yb, [=](hipStream_t s) { hipLaunchKernelGGL((apply_var_kernel<1, false, false>), dim3(b, nf), dim3(256), 0, s, vin, vout, nvec, fvs, d_luts); }, {}});
      vs.push_back({"apply bpermute", 2 * yb, [=](hipStream_t s) { hipLaunchKernelGGL((apply_var_kernel<2, false, false>), dim3(b, nf), dim3(256), 0, s, vin, vout, nvec, fvs, d_luts); }, {}});
      vs.push_back({"apply lut32x32 nt-store", 2 * yb, [=](hipStream_t s) { hipLaunchKernelGGL((apply_var_kernel<1, true, false>), dim3(b, nf), dim3(256), 0, s, vin, vout, nvec, fvs, d_luts); }, {}});
      // cache-reuse experiments: run read_sum (stands for the hist pass) then apply; time only the apply
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = -2; r < rounds; ++r) {
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st));
            v.run(st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0) v.us.push_back(ms * 1e3f);
        }
    }
    printf("frames=%d dist=%d  (4K Y plane %lld B/frame)\n", nf, dist, ysz);
    printf("%-34s %10s %10s %12s\n", "variant", "median_us", "min_us", "alg_GB/s(med)");
    for (auto& v : vs) {
        std::sort(v.us.begin(), v.us.end());
        const float med = v.us[v.us.size() / 2], mn = v.us[0];
        printf("%-34s %10.1f %10.1f %12.1f\n", v.name.c_str(), med, mn, v.alg_bytes / (med * 1e-6) / 1e9);
    }
    // --- ordering / Infinity-Cache reuse experiment: hist pass then apply pass, forward vs reverse, sub-batched
    auto seq = [&](int group, bool reverse, bool nt) {
        // processes nf frames in groups of `group`: hist(group) then apply(group)
        for (int g0 = 0; g0 < nf; g0 += group) {
            const int n = std::min(group, nf - g0);
            const int b = std::max(1, 2048 / n);
            hipLaunchKernelGGL((hist_var_kernel<32, false, 4>), dim3(std::min(b, 256), n), dim3(256), 0, st, vin + g0 * fvs, nvec, fvs, d_partial);
            if (reverse) {
                if (nt) hipLaunchKernelGGL((apply_var_kernel<1, true, true>), dim3(b, n), dim3(256), 0, st, vin + g0 * fvs, vout + g0 * fvs, nvec, fvs, d_luts + g0 * 256);
                else hipLaunchKernelGGL((apply_var_kernel<1, false, true>), dim3(b, n), dim3(256), 0, st, vin + g0 * fvs, vout + g0 * fvs, nvec, fvs, d_luts + g0 * 256);
            } else {
                if (nt) hipLaunchKernelGGL((apply_var_kernel<1, true, false>), dim3(b, n), dim3(256), 0, st, vin + g0 * fvs, vout + g0 * fvs, nvec, fvs, d_luts + g0 * 256);
                else hipLaunchKernelGGL((apply_var_kernel<1, false, false>), dim3(b, n), dim3(256), 0, st, vin + g0 * fvs, vout + g0 * fvs, nvec, fvs, d_luts + g0 * 256);
            }
        }
    };
    printf("\nhist+apply sequence over %d frames (3*W*H algorithmic bytes/frame):\n%-34s %10s %10s %12s\n", nf, "schedule", "median_us", "min_us", "alg_GB/s(med)");
    struct S { int group; bool rev, nt; };
    std::vector<S> scheds;
    for (int g : {nf, 32, 16, 12, 8, 4, 2, 1}) if (g <= nf) for (int rv = 0; rv < 2; ++rv) for (int nt = 0; nt < 2; ++nt) scheds.push_back({g, rv != 0, nt != 0});
    std::vector<std::vector<float>> t(scheds.size());
    for (int r = -1; r < rounds; ++r)
        for (size_t k = 0; k < scheds.size(); ++k) {
            CK(hipEventRecord(e0, st));
            seq(scheds[k].group, scheds[k].rev, scheds[k].nt);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0) t[k].push_back(ms * 1e3f);
        }
    for (size_t k = 0; k < scheds.size(); ++k) {
        std::sort(t[k].begin(), t[k].end());
        char name[64];
        snprintf(name, sizeof name, "group=%d %s %s", scheds[k].group, scheds[k].rev ? "reverse" : "forward", scheds[k].nt ? "nt-store" : "");
        const float med = t[k][t[k].size() / 2];
        printf("%-34s %10.1f %10.1f %12.1f\n", name, med, t[k][0], 3.0 * yb / (med * 1e-6) / 1e9);
    }
    return 0;
}
