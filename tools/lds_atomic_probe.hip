// What is the issue rate of ds_add_u32 (the histogram primitive of every kernel here) on MI355X, with the bank-replicated layout
// hist[bin][32] (copy = lane & 31, so a wave instruction is bank-conflict free)?  No global memory traffic: values come from a
// register LCG.  Answers whether the histogram kernels (hist_partial 3.7 TB/s, CLAHE tile histograms 4.3 TB/s of pixels) sit under
// an LDS-atomic roof rather than the HBM one.
//     hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_probe tools/lds_atomic_probe.hip && tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

template <int NT, int MODE>   // MODE 0: ds_add_u32 no return, conflict-free; 1: all lanes same bin (same-address, 2 lanes per copy); 2: plain ds_write (no atomic)
__global__ __launch_bounds__(NT) void probe(uint32_t* out, int iters)
{
    __shared__ uint32_t h[256 * 32];
    const int t = threadIdx.x;
    for (int i = t; i < 256 * 32; i += NT) h[i] = 0;
    __syncthreads();
    const uint32_t copy = t & 31;
    uint32_t x = 0x9E3779B9u * (uint32_t)(t + 1 + blockIdx.x * NT);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {                     // 16 pixels, as one 16-byte vector
            x = x * 1664525u + 1013904223u;
            const uint32_t bin = MODE == 1 ? 77u : (x >> 24);
            if (MODE == 2) h[(bin << 5) + copy] = x;
            else __hip_atomic_fetch_add(h + (bin << 5) + copy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (int i = t; i < 256 * 32; i += NT) s += h[i];
    if (s == 0xdeadbeefu) out[blockIdx.x] = s;             // keep the work alive
}

template <int NT, int MODE>
static int run(const char* name, int wgs_per_cu)
{
    uint32_t* d; CK(hipMalloc(&d, 1 << 20));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * wgs_per_cu, iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<NT, MODE>), dim3(grid), dim3(NT), 0, 0, d, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((probe<NT, MODE>), dim3(grid), dim3(NT), 0, 0, d, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double adds = (double)grid * NT * iters * 16;
    printf("%-44s NT=%4d WGs/CU=%d: %8.2f G lane-adds/s/CU = %6.2f T pixels/s on the chip (clock %d MHz -> %.1f lanes/clk/CU)\n", name, NT, wgs_per_cu,
           adds / (ms * 1e-3) / 1e9 / p.multiProcessorCount, adds / (ms * 1e-3) / 1e12, p.clockRate / 1000,
           adds / (ms * 1e-3) / p.multiProcessorCount / (p.clockRate * 1e3));
    CK(hipFree(d));
    return 0;
}

int main()
{
    run<256, 0>("ds_add_u32, bank-replicated (conflict free)", 4);
    run<256, 0>("ds_add_u32, bank-replicated (conflict free)", 5);
    run<512, 0>("ds_add_u32, bank-replicated (conflict free)", 2);
    run<512, 0>("ds_add_u32, bank-replicated (conflict free)", 4);
    run<1024, 0>("ds_add_u32, bank-replicated (conflict free)", 2);
    run<256, 1>("ds_add_u32, one bin for everybody", 4);
    run<256, 2>("ds_write_b32 (no atomic), same addresses", 4);
    return 0;
}
