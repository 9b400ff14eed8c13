// mall_lag_probe.hip -- can ONE persistent kernel that interleaves a read-only pass over frame f + D with a read + write pass over
// frame f beat the two passes run one after the other?  (docs/experiments.md R5.7.)  The second read of a frame then comes D frames
// after the first -- out of the 256 MiB Infinity Cache if D * (frame traffic) fits -- and nobody waits for anybody: dependencies
// point backwards.  This is the memory skeleton of a "histogram tickets ahead of interpolation tickets" CLAHE, with no compute at all:
// if the skeleton does not win, the kernel cannot.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_lag_probe.hip -o tools/mall_lag_probe && tools/mall_lag_probe [frames]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int W = 3840, H = 2160, kThreads = 256, kVPT = 8, kChunkVec = kThreads * kVPT;      // 32 KiB chunks
constexpr long long kYVec = (long long)W * H / 16, kFrameBytes = (long long)W * H * 3 / 2, kUVBytes = (long long)W * H / 2;
constexpr int kChunks = (int)((kYVec + kChunkVec - 1) / kChunkVec);                              // 254 per frame

__device__ __forceinline__ void read_chunk(const uint8_t* in, int f, int c, uint32_t* sink)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(in + (long long)f * kFrameBytes) + (long long)c * kChunkVec;
    const long long left = kYVec - (long long)c * kChunkVec;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kVPT; ++k) {
        const int i = k * kThreads + threadIdx.x;
        if (i < left) { const u32x4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
__device__ __forceinline__ void copy_chunk(const uint8_t* in, uint8_t* out, int f, int c)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(in + (long long)f * kFrameBytes) + (long long)c * kChunkVec;
    u32x4* q = reinterpret_cast<u32x4*>(out + (long long)f * kFrameBytes) + (long long)c * kChunkVec;
    const long long left = kYVec - (long long)c * kChunkVec;
    u32x4 v[kVPT];
#pragma unroll
    for (int k = 0; k < kVPT; ++k) { const int i = k * kThreads + threadIdx.x; v[k] = i < left ? p[i] : u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int k = 0; k < kVPT; ++k) { const int i = k * kThreads + threadIdx.x; if (i < left) q[i] = v[k]; }
    // this chunk's share of the UV plane: fill with 128
    u32x4* uv = reinterpret_cast<u32x4*>(out + (long long)f * kFrameBytes + (long long)W * H);
    const long long nuv = kUVBytes / 16, u0 = nuv * c / kChunks, u1 = nuv * (c + 1) / kChunks;
    const u32x4 g = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    for (long long i = u0 + threadIdx.x; i < u1; i += kThreads) uv[i] = g;
}
__global__ __launch_bounds__(kThreads) void read_kernel(const uint8_t* in, uint32_t* sink) { read_chunk(in, blockIdx.y, blockIdx.x, sink); }
__global__ __launch_bounds__(kThreads) void copy_kernel(const uint8_t* in, uint8_t* out) { copy_chunk(in, out, (int)gridDim.y - 1 - (int)blockIdx.y, blockIdx.x); }
// persistent: ticket t -> even: read chunk t/2 (frame-major order); odd: copy chunk t/2 - lag.  base: value of the counter at launch.
__global__ __launch_bounds__(kThreads) void mixed_kernel(const uint8_t* in, uint8_t* out, uint32_t* sink, unsigned long long* counter,
                                                        unsigned long long base, long long total, long long lag)
{
    __shared__ unsigned long long s_t;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_t = atomicAdd(counter, 1ULL) - base;
        __syncthreads();
        const long long t = (long long)s_t;
        if (t >= 2 * (total + lag)) break;
        const long long p = t >> 1;
        if (!(t & 1)) { if (p < total) read_chunk(in, (int)(p / kChunks), (int)(p % kChunks), sink); }
        else { const long long c = p - lag; if (c >= 0 && c < total) copy_chunk(in, out, (int)(c / kChunks), (int)(c % kChunks)); }
    }
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 64;
    uint8_t *in, *out; uint32_t* sink; unsigned long long* counter;
    CK(hipMalloc(&in, frames * kFrameBytes)); CK(hipMalloc(&out, frames * kFrameBytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&counter, 8));
    CK(hipMemset(in, 0x5a, frames * kFrameBytes)); CK(hipMemset(counter, 0, 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long issued = 0;
    const long long total = (long long)frames * kChunks;
    auto timeit = [&](auto&& fn) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f; };
    auto seq = [&] { hipLaunchKernelGGL(read_kernel, dim3(kChunks, frames), dim3(kThreads), 0, 0, in, sink); hipLaunchKernelGGL(copy_kernel, dim3(kChunks, frames), dim3(kThreads), 0, 0, in, out); };
    auto rd = [&] { hipLaunchKernelGGL(read_kernel, dim3(kChunks, frames), dim3(kThreads), 0, 0, in, sink); };
    auto cp = [&] { hipLaunchKernelGGL(copy_kernel, dim3(kChunks, frames), dim3(kThreads), 0, 0, in, out); };
    const int lags[] = {1, 2, 4, 8, 16};
    const int wgs[] = {4, 8};
    std::vector<std::vector<float>> res(3 + 10);
    for (int round = 0; round < 7; ++round) {
        res[0].push_back(timeit(rd)); res[1].push_back(timeit(cp)); res[2].push_back(timeit(seq));
        int slot = 3;
        for (int wg : wgs)
            for (int lag : lags) {
                const long long lagc = (long long)lag * kChunks;
                res[slot++].push_back(timeit([&] {
                    hipLaunchKernelGGL(mixed_kernel, dim3(cus * wg), dim3(kThreads), 0, 0, in, out, sink, counter, issued, total, lagc);
                    issued += 2 * (total + lagc) + (unsigned long long)cus * wg;       // every workgroup draws one ticket past the end
                }));
            }
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double gb_seq = frames * (2.0 * W * H + W * H * 1.5) / 1e9;
    printf("%d x 4K NV12 frames, 32 KiB chunks; us per pass over the batch (median of 7, interleaved rounds)\n", frames);
    printf("read-only pass (Y planes)                      %8.1f us  %6.2f TB/s\n", med(res[0]), frames * (double)W * H / med(res[0]) / 1e6);
    printf("copy pass (read Y, write Y, fill UV)           %8.1f us  %6.2f TB/s\n", med(res[1]), frames * (2.5 * W * H) / med(res[1]) / 1e6);
    printf("both, one after the other (two launches)       %8.1f us  %6.2f TB/s of the %.2f GB requested\n", med(res[2]), gb_seq * 1e3 / med(res[2]), gb_seq);
    int slot = 3;
    for (int wg : wgs)
        for (int lag : lags) {
            printf("mixed persistent kernel, %d WGs/CU, copy %2d frames behind the read %8.1f us  %6.2f TB/s requested\n", wg, lag, med(res[slot]), gb_seq * 1e3 / med(res[slot]));
            ++slot;
        }
    return 0;
}
