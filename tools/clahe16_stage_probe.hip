// 16-bit CLAHE, 12-bit content: what does each of the shipped kernels cost ALONE (launched back to back with itself)
// and what do they cost in the shipped sequence?  A read-only kernel that follows a writing one is charged the writer's drain, so
// per-kernel times taken inside the sequence say little about either kernel.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/clahe16_stage_probe tools/clahe16_stage_probe.hip
//     tools/clahe16_stage_probe [frames = 16] [content: 0 noise | 1 constant | 2 smooth]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../opencv-opencl_amd/csrc/lumaeq_kernels.hip.h"
using namespace mi;
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void empty1024_kernel(const uint32_t* flags, uint32_t* out) { if (flags[blockIdx.y] == 0xdeadbeefu) out[0] = 1; }
__global__ __launch_bounds__(256) void empty256_kernel(const uint32_t* flags, uint32_t* out) { if (flags[blockIdx.y] == 0xdeadbeefu) out[0] = 1; }

template <class F>
static float time_us(F&& launch, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv)
{
    const int nf = argc > 1 ? atoi(argv[1]) : 16, content = argc > 2 ? atoi(argv[2]) : 0;
    const int W = argc > 3 ? atoi(argv[3]) : 3840, H = 2160, TX = 8, TY = 8, tiles = TX * TY;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    ClaheGeom g{};
    g.width = W; g.height = H; g.tiles_x = TX; g.tiles_y = TY; g.tile_w = W / TX; g.tile_h = H / TY;
    g.inv_tw = 1.0f / (float)g.tile_w; g.inv_th = 1.0f / (float)g.tile_h; g.contract = 0;
    const int area = g.tile_w * g.tile_h;
    const float lut_scale16 = 65535.0f / (float)area;
    const int clip16 = std::max(1, (int)(2.0 * area / 65536));
    const size_t plane = (size_t)W * H * 2, step = (size_t)W * 2;
    uint8_t *src, *dst, *scratch;
    CK(hipMalloc(&src, plane * nf)); CK(hipMalloc(&dst, plane * nf));
    const size_t per_frame = (size_t)tiles * kHist16 * 6 + ((size_t)tiles + 1) * sizeof(Range16) + 4;
    CK(hipMalloc(&scratch, per_frame * nf));
    uint32_t* hist = (uint32_t*)scratch;
    uint16_t* luts = (uint16_t*)(scratch + (size_t)nf * tiles * kHist16 * 4);
    Range16* ranges = (Range16*)(scratch + (size_t)nf * tiles * kHist16 * 6);
    Range16* franges = ranges + (size_t)nf * tiles;
    uint32_t* fdone = (uint32_t*)(franges + nf);
    uint32_t* sync; CK(hipMalloc(&sync, 16 * (size_t)nf + 16)); CK(hipMemset(sync, 0, 16 * (size_t)nf + 16)); uint32_t* hint = sync + 4 * (size_t)nf;
    {
        std::vector<uint16_t> hb((size_t)W * H);
        uint32_t x = 777u;
        for (int f = 0; f < nf; ++f) {
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx) {
                    x = x * 1664525u + 1013904223u;
                    uint16_t v = content == 0 ? (uint16_t)(x >> 20) : content == 1 ? (uint16_t)777
                               : (uint16_t)(2048.0 + 1500.0 * std::sin(xx * 0.004 + f) * std::cos(y * 0.006) + (double)((x >> 27) & 15));
                    hb[(size_t)y * W + xx] = v;
                }
            CK(hipMemcpy(src + plane * f, hb.data(), plane, hipMemcpyHostToDevice));
        }
    }
    CK(hipFuncSetAttribute((const void*)tile_hist12_kernel<kHist12Threads, kCopies12>, hipFuncAttributeMaxDynamicSharedMemorySize, kHist12Words * 4));
    CK(hipFuncSetAttribute((const void*)clahe_interp16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kInterp16Entries * 8));
    const int npairs = TX + 1, bands = TY + 1;
    // the library's rule (host/clahe16.inc.hpp): fill the chip when frames are few, two sub-bands per band otherwise
    long long want = ((long long)cus * 4 + (long long)npairs * bands * nf - 1) / ((long long)npairs * bands * nf);
    int subs_rule = (int)std::max<long long>(1, std::min<long long>({want, (long long)std::max(1, g.tile_h / 16), 16LL}));
    if (subs_rule < 2 && g.tile_h >= 128) subs_rule = 2;
    const int subs = argc > 4 ? atoi(argv[4]) : subs_rule;
    const long long rows = (long long)bands * subs * nf;
    const unsigned igrid = (unsigned)((rows + 7) / 8 * 8 * npairs);
    auto k_hist = [&] { hipLaunchKernelGGL((tile_hist12_kernel<kHist12Threads, kCopies12>), dim3(tiles, nf), dim3(kHist12Threads), kHist12Words * 4, 0, (const uint8_t*)src, (long long)step, (long long)plane, g, hist, ranges, lut_scale16, clip16, luts, sync, franges, fdone, hint); };
    auto k_lut = [&] { hipLaunchKernelGGL(tile_lut16_kernel, dim3(tiles, nf), dim3(1024), 0, 0, (const uint32_t*)hist, (const Range16*)ranges, g, lut_scale16, clip16, luts, franges, (const uint32_t*)fdone, hint); };
    auto k_int = [&] { hipLaunchKernelGGL(clahe_interp16_kernel<false>, dim3(igrid), dim3(kInterp16Threads), kInterp16Entries * 8, 0,
                                          (const uint8_t*)src, (long long)step, (long long)plane, dst, (long long)step, (long long)plane, g, (const uint16_t*)luts, (const Range16*)franges, subs, nf, (const Range16*)ranges, hint); };
    k_hist(); k_lut(); k_int(); CK(hipDeviceSynchronize());
    const char* names[3] = {"12-bit noise", "constant 777", "smooth 12-bit + 4 bits of noise"};
    printf("%d frames of %d x %d CV_16UC1, %s, subs = %d\n", nf, W, H, names[content], subs);
    const int reps = 20;
    const float th = time_us(k_hist, reps), tl = time_us(k_lut, reps), ti = time_us(k_int, reps);
    const float ts = time_us([&] { k_hist(); k_lut(); k_int(); }, reps);
    const float thi = time_us([&] { k_hist(); k_int(); }, reps);
    const double mb = (double)plane * nf / 1e6;
    const float te1 = time_us([&] { hipLaunchKernelGGL(empty1024_kernel, dim3(tiles, nf), dim3(1024), 0, 0, (const uint32_t*)franges, (uint32_t*)hist); }, reps);
    const float te2 = time_us([&] { hipLaunchKernelGGL(empty256_kernel, dim3(tiles, nf), dim3(256), 0, 0, (const uint32_t*)franges, (uint32_t*)hist); }, reps);
    const float the = time_us([&] { k_hist(); hipLaunchKernelGGL(empty1024_kernel, dim3(tiles, nf), dim3(1024), 0, 0, (const uint32_t*)franges, (uint32_t*)hist); k_int(); }, reps);
    printf("a kernel of the same grid that reads one word and returns: %.1f us with 1024 threads, %.1f us with 256; sequence hist, that, interp %.1f us\n", te1, te2, the);
    printf("tile_hist12_kernel alone   %7.1f us  (%.2f TB/s of the pixels read)\n", th, mb / th);
    printf("tile_lut16_kernel alone    %7.1f us\n", tl);
    printf("clahe_interp16_kernel alone%7.1f us  (%.2f TB/s of read + write)\n", ti, 2 * mb / ti);
    printf("sum of the three alone     %7.1f us\n", th + tl + ti);
    printf("sequence hist, lut, interp %7.1f us  = %.0f frames/s (%.2f TB/s of 3*W*H*2 B)\n", ts, nf / (ts * 1e-6), 3 * mb / ts);
    printf("sequence hist, interp      %7.1f us  (the LUT kernel has nothing to do for such frames)\n", thi);
    // the same sequence over CHUNKS of the batch: a chunk that fits the memory-side cache (256 MB) is still there when the interpolation re-reads it
    for (int chunk = nf / 2; chunk >= 2; chunk /= 2) {
        const long long crows = (long long)bands * subs * chunk;
        const unsigned cgrid = (unsigned)((crows + 7) / 8 * 8 * npairs);
        const float tc = time_us([&] {
            for (int f0 = 0; f0 < nf; f0 += chunk) {
                const uint8_t* sp = src + plane * f0; uint8_t* dp = dst + plane * f0;
                hipLaunchKernelGGL((tile_hist12_kernel<kHist12Threads, kCopies12>), dim3(tiles, chunk), dim3(kHist12Threads), kHist12Words * 4, 0, sp, (long long)step, (long long)plane, g, hist, ranges, lut_scale16, clip16, luts, sync, franges, fdone, hint);
                hipLaunchKernelGGL(tile_lut16_kernel, dim3(tiles, chunk), dim3(1024), 0, 0, (const uint32_t*)hist, (const Range16*)ranges, g, lut_scale16, clip16, luts, franges, (const uint32_t*)fdone, hint);
                hipLaunchKernelGGL(clahe_interp16_kernel<false>, dim3(cgrid), dim3(kInterp16Threads), kInterp16Entries * 8, 0,
                                   sp, (long long)step, (long long)plane, dp, (long long)step, (long long)plane, g, (const uint16_t*)luts, (const Range16*)franges, subs, chunk, (const Range16*)ranges, hint);
            }
        }, reps);
        printf("sequence in chunks of %2d    %7.1f us  = %.0f frames/s\n", chunk, tc, nf / (tc * 1e-6));
    }
    return 0;
}
