// Per-frame cost of moving a plane host -> device -> host with the copy engines (hipMemcpyAsync, what mi_pipe does) against KERNELS that
// read / write the pinned host memory themselves.  The copy engine pays ~13-18 us of turnaround per copy (a 1080p stream is bound by it:
// 17 k frames/s where the link allows 25 k); a kernel pays a launch.  Steady state, frames back to back, uploads and downloads of
// different frames concurrent (two streams), as in the pipe.
//     hipcc --offload-arch=gfx950 -O3 -o tools/zero_copy_probe tools/zero_copy_probe.hip && tools/zero_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long nvec)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        u32x4 q[U];
#pragma unroll
        for (int k = 0; k < U; ++k) q[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < U; ++k) __builtin_nontemporal_store(q[k], dst + i + k * stride);
    }
    for (; i < nvec; i += stride) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int frames = 400, ring = 8;
    for (int res = 0; res < 3; ++res) {
        const size_t bytes = res == 0 ? (size_t)3840 * 2160 : res == 1 ? (size_t)1920 * 1080 : (size_t)1280 * 720;
        uint8_t *h_in, *h_out, *d[ring];
        CK(hipHostMalloc(&h_in, bytes * ring)); CK(hipHostMalloc(&h_out, bytes * ring));
        for (int i = 0; i < ring; ++i) CK(hipMalloc(&d[i], bytes));
        for (size_t i = 0; i < bytes * ring; i += 4096) h_in[i] = (uint8_t)i;
        hipStream_t su, sd; CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
        hipEvent_t up[ring], dn[ring];
        for (int i = 0; i < ring; ++i) { CK(hipEventCreateWithFlags(&up[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&dn[i], hipEventDisableTiming)); }
        const long long nvec = (long long)(bytes / 16);
        printf("plane of %zu bytes (%s)\n", bytes, res == 0 ? "4K luma" : res == 1 ? "1080p luma" : "720p luma");
        for (int mode = 0; mode < 6; ++mode) {
            // mode 0: copy engines.  1..5: kernels with 8 / 16 / 32 / 64 / 128 workgroups per copy
            const int wgs = mode == 0 ? 0 : 4 << mode;
            for (int i = 0; i < ring; ++i) CK(hipEventRecord(dn[i], sd));
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int f = 0; f < frames; ++f) {
                const int s = f % ring;
                // slot s is free again once its previous download has finished
                CK(hipStreamWaitEvent(su, dn[s], 0));
                if (mode == 0) CK(hipMemcpyAsync(d[s], h_in + bytes * s, bytes, hipMemcpyHostToDevice, su));
                else hipLaunchKernelGGL(copy_kernel<8>, dim3(wgs), dim3(256), 0, su, (const u32x4*)(h_in + bytes * s), (u32x4*)d[s], nvec);
                CK(hipEventRecord(up[s], su));
                CK(hipStreamWaitEvent(sd, up[s], 0));
                if (mode == 0) CK(hipMemcpyAsync(h_out + bytes * s, d[s], bytes, hipMemcpyDeviceToHost, sd));
                else hipLaunchKernelGGL(copy_kernel<8>, dim3(wgs), dim3(256), 0, sd, (const u32x4*)d[s], (u32x4*)(h_out + bytes * s), nvec);
                CK(hipEventRecord(dn[s], sd));
            }
            CK(hipDeviceSynchronize());
            const double dt = now() - t0;
            if (mode == 0) printf("  copy engines (hipMemcpyAsync)      : %7.1f us per frame = %7.0f frames/s, %5.1f GB/s each way\n", dt / frames * 1e6, frames / dt, bytes * frames / dt / 1e9);
            else printf("  copy kernels, %3d workgroups each   : %7.1f us per frame = %7.0f frames/s, %5.1f GB/s each way\n", wgs, dt / frames * 1e6, frames / dt, bytes * frames / dt / 1e9);
            fflush(stdout);
        }
        for (int i = 0; i < ring; ++i) { CK(hipFree(d[i])); CK(hipEventDestroy(up[i])); CK(hipEventDestroy(dn[i])); }
        CK(hipStreamDestroy(su)); CK(hipStreamDestroy(sd));
        CK(hipHostFree(h_in)); CK(hipHostFree(h_out));
    }
    return 0;
}
