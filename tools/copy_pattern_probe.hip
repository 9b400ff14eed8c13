// Which property of clahe_interp16_kernel's access pattern costs the bandwidth?  With its table look-ups and blend compiled out the
// kernel still takes as long (a plain copy in its geometry: 3.7 TB/s of read + write against 6.3 for a linear copy).  This probe
// copies a batch of 4K CV_16UC1 frames rectangle by rectangle, one workgroup per rectangle, and varies: rectangle width and
// alignment, threads per workgroup, workgroups per CU (through a dummy LDS allocation), rows in flight per lane, lane -> row mapping.
//     hipcc --offload-arch=gfx950 -O3 -o tools/copy_pattern_probe tools/copy_pattern_probe.hip && tools/copy_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid = (rects_x * rects_y, frames).  A rectangle is seg_groups 16-byte groups wide, starting at byte x_off + rx * seg_groups * 16,
// and rect_rows rows high.  MAP 0: lane -> (group = t % groups, phase = t / groups), rows y = phase + k * phases (as shipped);
// MAP 1: a WAVE owns whole rows: lane -> group = lane + 64 * i, wave w takes rows w, w + nwaves, ...
template <int NT, int R, int MAP>
__global__ __launch_bounds__(NT) void copy_rects(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long step, long long frame,
                                                 int rects_x, int seg_groups, int x_off, int rect_rows, int height, int width_bytes, int xcd_rows)
{
    extern __shared__ uint32_t dummy[];
    const int t = threadIdx.x;
    if (t == 0) dummy[0] = 1;
    int rx = blockIdx.x % rects_x, ry = blockIdx.x / rects_x, f = blockIdx.y;
    if (xcd_rows) {
        // workgroups go to the 8 XCDs round-robin in linear order: give every XCD whole ROWS of rectangles (all rects_x of them, one
        // after the other), so that the two halves of a cache line cut by a rectangle edge meet in ONE L2
        const long long id = (long long)blockIdx.x + (long long)gridDim.x * blockIdx.y;
        const int xcd = (int)(id % 8);
        const long long k = id / 8;
        rx = (int)(k % rects_x);
        const long long rowgroup = (k / rects_x) * 8 + xcd;
        const int rects_y = gridDim.x / rects_x;
        if (rowgroup >= (long long)rects_y * gridDim.y) return;
        ry = (int)(rowgroup % rects_y); f = (int)(rowgroup / rects_y);
    }
    const uint8_t* s = src + (long long)f * frame;
    uint8_t* d = dst + (long long)f * frame;
    const int y_lo = ry * rect_rows, y_hi = min(height, y_lo + rect_rows);
    const long long xb0 = (long long)x_off + (long long)rx * seg_groups * 16;
    if (MAP == 0) {
        const int phases = max(1, NT / seg_groups);
        const int gi = t % seg_groups, phase = t / seg_groups;
        if (phase >= phases) return;
        for (int g0 = gi; g0 < seg_groups; g0 += NT) {
            const long long xb = xb0 + (long long)g0 * 16;
            if (xb + 16 > width_bytes) continue;
            int y = y_lo + phase;
            for (; y + (R - 1) * phases < y_hi; y += R * phases) {
                u32x4 q[R];
#pragma unroll
                for (int k = 0; k < R; ++k) q[k] = *reinterpret_cast<const u32x4*>(s + (long long)(y + k * phases) * step + xb);
#pragma unroll
                for (int k = 0; k < R; ++k) *reinterpret_cast<u32x4*>(d + (long long)(y + k * phases) * step + xb) = q[k] + 1u;
            }
            for (; y < y_hi; y += phases) *reinterpret_cast<u32x4*>(d + (long long)y * step + xb) = *reinterpret_cast<const u32x4*>(s + (long long)y * step + xb) + 1u;
        }
    } else {
        // items = rows x groups walked linearly: item i -> row i / seg_groups, group i % seg_groups; R items in flight per lane
        const int items = (y_hi - y_lo) * seg_groups;
        int i = t;
        for (; i + (R - 1) * NT < items; i += R * NT) {
            u32x4 q[R]; long long off[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int it = i + k * NT, row = it / seg_groups, gi = it - row * seg_groups;
                off[k] = (long long)(y_lo + row) * step + xb0 + (long long)gi * 16;
                q[k] = *reinterpret_cast<const u32x4*>(s + off[k]);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) *reinterpret_cast<u32x4*>(d + off[k]) = q[k] + 1u;
        }
        for (; i < items; i += NT) {
            const int row = i / seg_groups, gi = i - row * seg_groups;
            const long long off = (long long)(y_lo + row) * step + xb0 + (long long)gi * 16;
            *reinterpret_cast<u32x4*>(d + off) = *reinterpret_cast<const u32x4*>(s + off) + 1u;
        }
    }
}

// The shipped kernel's own geometry: (tiles_x + 1) x (tiles_y + 1) rectangles per frame, the outer ones half size, a margin of 4 pixels
// around each (lanes of the margin groups idle), lane -> (group, phase) as shipped.  STAGE: also fill 64 KiB of LDS from global memory first.
template <int R, int STAGE>
__global__ __launch_bounds__(512) void copy_pairs(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long step, long long frame,
                                                  int width, int height, int tile_w, int tile_h, int tiles_x, int tiles_y, const uint32_t* __restrict__ table,
                                                  int subs, int xcd_rows, int strip_px)
{
    extern __shared__ uint32_t lds[];
    const int t = threadIdx.x;
    const int npairs = tiles_x + 1, bands = tiles_y + 1;
    int f = blockIdx.y, pr, band, sub;
    {
        int id = blockIdx.x;
        sub = id % subs; id /= subs;
        pr = id % npairs; band = id / npairs;
    }
    if (xcd_rows) {                                               // one XCD gets all pairs of a (frame, band, sub) row, one after the other
        const long long id = (long long)blockIdx.x + (long long)gridDim.x * blockIdx.y;
        const int xcd = (int)(id % 8);
        const long long k = id / 8;
        pr = (int)(k % npairs);
        const long long rg = (k / npairs) * 8 + xcd;
        if (rg >= (long long)bands * subs * gridDim.y) return;
        sub = (int)(rg % subs); band = (int)((rg / subs) % bands); f = (int)(rg / ((long long)subs * bands));
    }
    if (STAGE) {
        const u32x4* tp = reinterpret_cast<const u32x4*>(table) + (size_t)(blockIdx.x % 64) * 8192;
        for (int i = t; i < 4096; i += 512) reinterpret_cast<u32x4*>(lds)[i] = tp[i & 2047];
        __syncthreads();
    }
    const int yb_lo = max(0, ((2 * band - 1) * tile_h) / 2), yb_hi = min(height, ((2 * band + 1) * tile_h + 1) / 2);
    const int y_lo = yb_lo + (yb_hi - yb_lo) * sub / subs, y_hi = yb_lo + (yb_hi - yb_lo) * (sub + 1) / subs;
    int x_lo, x_hi, own_lo, own_hi;
    if (strip_px > 0) {                                           // aligned strips instead of pairs: [pr * strip_px, (pr + 1) * strip_px)
        x_lo = own_lo = pr * strip_px; x_hi = own_hi = min(width, (pr + 1) * strip_px);
    } else {
        x_lo = max(0, ((2 * pr - 1) * tile_w) / 2 - 4); x_hi = min(width, ((2 * pr + 1) * tile_w + 1) / 2 + 4);
        own_lo = max(0, ((2 * pr - 1) * tile_w) / 2); own_hi = min(width, ((2 * pr + 1) * tile_w + 1) / 2);
    }
    if (x_lo >= x_hi) return;
    const int g_lo = x_lo >> 3, ngroups = ((x_hi + 7) >> 3) - g_lo;
    const int phases = max(1, 512 / ngroups);
    const int gi = t % ngroups, phase = t / ngroups;
    if (phase >= phases) return;
    const int x0 = (g_lo + gi) << 3;
    if (x0 < own_lo || x0 + 8 > own_hi) return;
    const uint8_t* s = src + (long long)f * frame + 2LL * x0;
    uint8_t* d = dst + (long long)f * frame + 2LL * x0;
    int y = y_lo + phase;
    for (; y + (R - 1) * phases < y_hi; y += R * phases) {
        u32x4 q[R];
#pragma unroll
        for (int k = 0; k < R; ++k) q[k] = *reinterpret_cast<const u32x4*>(s + (long long)(y + k * phases) * step);
#pragma unroll
        for (int k = 0; k < R; ++k) *reinterpret_cast<u32x4*>(d + (long long)(y + k * phases) * step) = q[k] + (STAGE ? lds[(q[k].x & 4095u) * 4] : 1u);
    }
    for (; y < y_hi; y += phases) *reinterpret_cast<u32x4*>(d + (long long)y * step) = *reinterpret_cast<const u32x4*>(s + (long long)y * step) + 1u;
}

static uint8_t *g_src, *g_dst;
static const int W = 3840, H = 2160, NF = 16;

template <int NT, int R, int MAP>
static void run(const char* what, int seg_groups, int x_off, int rect_rows, int lds_kib, int xcd_rows = 0)
{
    auto k = copy_rects<NT, R, MAP>;
    const size_t lds = (size_t)lds_kib << 10;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long step = (long long)W * 2, frame = step * H;
    const int rects_x = (int)((step - x_off + (long long)seg_groups * 16 - 1) / ((long long)seg_groups * 16)), rects_y = (H + rect_rows - 1) / rect_rows;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 10;
    hipLaunchKernelGGL(k, dim3(rects_x * rects_y, NF), dim3(NT), lds, 0, g_src, g_dst, step, frame, rects_x, seg_groups, x_off, rect_rows, H, (int)step, xcd_rows);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k, dim3(rects_x * rects_y, NF), dim3(NT), lds, 0, g_src, g_dst, step, frame, rects_x, seg_groups, x_off, rect_rows, H, (int)step, xcd_rows);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, bytes = 2.0 * (frame - (double)x_off * H) * NF;
    printf("%-28s NT=%4d R=%d map=%d seg=%4d B off=%3d rows=%3d lds=%3d KiB  wgs=%5d: %7.1f us  %5.2f TB/s\n", what, NT, R, MAP, seg_groups * 16, x_off, rect_rows, lds_kib,
           rects_x * rects_y * NF, us, bytes / us / 1e6);
    fflush(stdout);
}

template <int R, int STAGE>
static void run_pairs(const char* what, int subs = 1, int xcd_rows = 0, int strip_px = 0)
{
    auto k = copy_pairs<R, STAGE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const long long step = (long long)3840 * 2, frame = step * 2160;
    uint32_t* table; CK(hipMalloc(&table, 64 * 8192 * 16)); CK(hipMemset(table, 0, 64 * 8192 * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 10;
    // with the XCD mapping the grid is padded so that 8 * ceil(rows / 8) rows of 9 workgroups exist
    const int rows = 9 * subs * 16, grid_x = xcd_rows ? ((rows + 7) / 8 * 8 * 9 + 15) / 16 : 81 * subs;
    auto launch = [&] { hipLaunchKernelGGL(k, dim3(grid_x, 16), dim3(512), 65536, 0, g_src, g_dst, step, frame, 3840, 2160, 480, 270, 8, 8, (const uint32_t*)table, subs, xcd_rows, strip_px); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, bytes = 2.0 * frame * 16;
    printf("%-40s R=%d stage=%d subs=%d xcd_rows=%d strip=%4d px wgs=%5d: %7.1f us  %5.2f TB/s\n", what, R, STAGE, subs, xcd_rows, strip_px, grid_x * 16, us, bytes / us / 1e6);
    CK(hipFree(table));
}

int main()
{
    const size_t bytes = (size_t)W * H * 2 * NF;
    CK(hipMalloc(&g_src, bytes)); CK(hipMalloc(&g_dst, bytes));
    CK(hipMemset(g_src, 1, bytes)); CK(hipMemset(g_dst, 0, bytes));
    run_pairs<4, 0>("the shipped geometry, copy only");
    run_pairs<4, 1>("the shipped geometry, staging + copy");
    run_pairs<2, 0>("the shipped geometry, copy only");
    // 9 aligned strips of 448 pixels = 896 bytes = 7 lines (the 9th is 256 pixels wide) instead of the 9 pairs, same bands
    for (int xr = 0; xr < 2; ++xr)
        for (int sb = 1; sb <= 2; ++sb) { run_pairs<4, 1>("aligned 896-byte strips, bands as shipped", sb, xr, 448); run_pairs<4, 1>("pairs as shipped", sb, xr, 0); }
    for (int xr = 0; xr < 2; ++xr)
        for (int sb = 1; sb <= 8; sb *= 2) { run_pairs<4, 0>("shipped geometry", sb, xr); run_pairs<4, 1>("shipped geometry", sb, xr); }
    // linear reference: full rows, many workgroups
    run<256, 4, 1>("full rows, 8 rows per wg", 480, 0, 8, 0);
    run<256, 4, 1>("full rows, 32 rows per wg", 480, 0, 32, 0);
    // the shipped geometry: 62 groups (992 B) at byte offset 464 (first rectangle starts at pixel 232), ~270 rows, 512 threads, 64 KiB
    run<512, 4, 0>("as shipped (approx.)", 60, 480, 270, 64);
    run<512, 4, 0>("as shipped + XCD rows", 60, 480, 270, 64, 1);
    run<512, 4, 0>("135 rows + XCD rows", 60, 480, 135, 64, 1);
    run<512, 4, 0>("aligned + XCD rows", 64, 0, 270, 64, 1);
    run<512, 4, 0>("no LDS limit", 60, 480, 270, 0);
    run<512, 4, 0>("aligned 1024 B segments", 64, 0, 270, 64);
    run<512, 4, 0>("aligned, no LDS limit", 64, 0, 270, 0);
    run<512, 4, 1>("linear items in the rect", 60, 480, 270, 64);
    run<512, 4, 1>("linear items, aligned", 64, 0, 270, 64);
    run<512, 8, 1>("linear items, R=8", 60, 480, 270, 64);
    run<1024, 4, 1>("linear items, 1024 thr", 60, 480, 270, 64);
    run<512, 4, 1>("linear, 135 rows", 60, 480, 135, 64);
    run<512, 4, 1>("linear, 68 rows", 60, 480, 68, 64);
    run<512, 4, 1>("linear, 34 rows", 60, 480, 34, 64);
    run<512, 4, 1>("linear, 34 rows, no LDS", 60, 480, 34, 0);
    run<256, 4, 1>("linear, 34 rows, 256 thr", 60, 480, 34, 32);
    run<512, 4, 0>("shipped map, 135 rows", 60, 480, 135, 64);
    run<512, 4, 0>("shipped map, 68 rows", 60, 480, 68, 64);
    run<512, 2, 0>("shipped map, R=2", 60, 480, 270, 64);
    run<512, 1, 0>("shipped map, R=1", 60, 480, 270, 64);
    run<512, 8, 0>("shipped map, R=8", 60, 480, 270, 64);
    return 0;
}
