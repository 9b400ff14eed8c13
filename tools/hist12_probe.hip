// What bounds the 12-bit tile-histogram sweep of 16-bit CLAHE (kernels/clahe16.hip.h tile_hist12_kernel: 3.3 TB/s of pixels, a
// constant frame as slow as noise)?  Stand-alone variants of the sweep over a 4K batch of CV_16UC1 frames cut into 8x8 tiles:
// threads per workgroup x LDS copies (= workgroups per CU), loads only / loads + counting, loads in flight per lane.
//     hipcc --offload-arch=gfx950 -O3 -o tools/hist12_probe tools/hist12_probe.hip && tools/hist12_probe [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_inc(uint32_t* h, uint32_t i) { __hip_atomic_fetch_add(h + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// MODE 0: loads only (OR-reduced into a register); 1: + packed min/max + ds_add at (v & 4095) * COPIES + copy
// SETS: 16-byte loads per lane and iteration (SETS in flight, + SETS more when DB = 1: the next set is loaded before the current is counted)
// SPLIT: workgroups per tile (rows divided)
template <int NT, int COPIES, int MODE, int SETS, int DB>
__global__ __launch_bounds__(NT) void sweep(const uint8_t* __restrict__ src_base, long long step, long long frame_stride, int tile_w, int tile_h,
                                            int tiles_x, int split, uint32_t* __restrict__ out)
{
    extern __shared__ uint32_t h[];
    const int t = threadIdx.x;
    const int part = blockIdx.x % split, tile = blockIdx.x / split, f = blockIdx.y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int r0 = tile_h * part / split, r1 = tile_h * (part + 1) / split;
    const int slots = tile_w >> 3, vitems = (r1 - r0) * slots;
    const uint8_t* tbase = src_base + (long long)f * frame_stride + ((long long)ty * tile_h + r0) * step + (long long)tx * tile_w * 2;
    if (MODE) for (int i = t; i < 4096 * COPIES / 4; i += NT) reinterpret_cast<u32x4*>(h)[i] = u32x4{0u, 0u, 0u, 0u};
    const uint32_t cp = (uint32_t)t & (uint32_t)(COPIES - 1);
    int row = t / slots, slot = t - row * slots;
    const int vdrow = NT / slots, vdslot = NT - vdrow * slots;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    auto load_set = [&](int it, u32x4* q, bool* qv) {
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            qv[k] = it + k * NT < vitems;
            const u32x4* ptr = reinterpret_cast<const u32x4*>(tbase + (long long)row * step + (slot << 4));
            q[k] = qv[k] ? *ptr : zero;
            row += vdrow; slot += vdslot;
            if (slot >= slots) { slot -= slots; ++row; }
        }
    };
    u32x4 cur[SETS], nxt[SETS]; bool cv[SETS], nv[SETS];
    load_set(t, cur, cv);
    if (MODE) __syncthreads();
    uint32_t acc = 0;
    u16x2 pmin = {0xffff, 0xffff}, pmax = {0, 0};
    for (int it = t; it < vitems; it += SETS * NT) {
        const bool more = it + SETS * NT < vitems;
        if (DB && more) load_set(it + SETS * NT, nxt, nv);
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            if (!cv[k]) continue;
            const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) { acc |= w[j]; continue; }
                const u16x2 v = __builtin_bit_cast(u16x2, w[j]);
                pmin = __builtin_elementwise_min(pmin, v); pmax = __builtin_elementwise_max(pmax, v);
                lds_inc(h, ((w[j] & 4095u) * COPIES) | cp);
                lds_inc(h, (((w[j] >> 16) & 4095u) * COPIES) | cp);
            }
        }
        if (more) {
            if (DB) {
#pragma unroll
                for (int k = 0; k < SETS; ++k) { cur[k] = nxt[k]; cv[k] = nv[k]; }
            } else {
                load_set(it + SETS * NT, cur, cv);
            }
        }
    }
    if (MODE) {
        __syncthreads();
        for (int i = t; i < 4096 * COPIES; i += NT) acc += h[i];
        acc += __builtin_bit_cast(uint32_t, pmin) ^ __builtin_bit_cast(uint32_t, pmax);
    }
    if (acc == 0xdeadbeefu) out[blockIdx.x] = acc;
}

static int g_cus = 256;
static uint8_t* g_src; static uint32_t* g_out;
static int g_frames;
static const int W = 3840, H = 2160, TX = 8, TY = 8;

template <int NT, int COPIES, int MODE, int SETS, int DB>
static void run(const char* what, int split)
{
    const size_t lds = MODE ? (size_t)4096 * COPIES * 4 : 0;
    auto k = sweep<NT, COPIES, MODE, SETS, DB>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long step = (long long)W * 2, fs = step * H;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(TX * TY * split, g_frames), dim3(NT), lds, 0, g_src, step, fs, W / TX, H / TY, TX, split, g_out);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (double)fs * g_frames;
    printf("%-30s NT=%4d copies=%d sets=%d db=%d split=%d lds=%3zu KiB: avg %7.1f us  best %7.1f us  = %5.2f TB/s (avg), per 16 frames %6.1f us\n", what, NT, COPIES, SETS, DB, split,
           lds >> 10, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, sum / reps * 1e3 * 16 / g_frames);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    g_frames = argc > 1 ? atoi(argv[1]) : 32;
    const int noise = argc > 2 ? atoi(argv[2]) : 1;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); g_cus = p.multiProcessorCount;
    const size_t bytes = (size_t)W * H * 2 * g_frames;
    CK(hipMalloc(&g_src, bytes)); CK(hipMalloc(&g_out, 1 << 20));
    {
        std::vector<uint16_t> hbuf((size_t)W * H);
        uint32_t x = 12345u;
        for (auto& v : hbuf) { x = x * 1664525u + 1013904223u; v = noise ? (uint16_t)(x >> 20) : (uint16_t)777; }
        for (int f = 0; f < g_frames; ++f) CK(hipMemcpy(g_src + (size_t)f * W * H * 2, hbuf.data(), (size_t)W * H * 2, hipMemcpyHostToDevice));
    }
    printf("%d frames of 4K CV_16UC1 (%s), %d CUs\n", g_frames, noise ? "12-bit noise" : "constant 777", g_cus);
    run<1024, 4, 0, 4, 1>("loads only", 1);
    run<512, 4, 0, 4, 1>("loads only", 1);
    run<256, 4, 0, 4, 1>("loads only", 1);
    run<256, 4, 0, 4, 1>("loads only", 4);
    run<256, 4, 0, 8, 0>("loads only", 4);
    run<1024, 4, 1, 4, 1>("count (as shipped)", 1);
    run<1024, 4, 1, 4, 0>("count, no double buffer", 1);
    run<1024, 2, 1, 4, 1>("count", 1);
    run<512, 2, 1, 4, 1>("count", 1);
    run<512, 1, 1, 4, 1>("count", 1);
    run<256, 1, 1, 4, 1>("count", 1);
    run<512, 4, 1, 4, 1>("count", 1);
    run<512, 2, 1, 4, 1>("count", 2);
    run<256, 1, 1, 4, 1>("count", 4);
    run<256, 2, 1, 4, 1>("count", 4);
    run<256, 1, 1, 8, 0>("count", 4);
    run<512, 2, 1, 2, 1>("count", 1);
    run<512, 2, 1, 8, 0>("count", 1);
    return 0;
}
