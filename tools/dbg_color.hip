#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../opencv-opencl_amd/csrc/lumaeq_kernels.hip.h"
using namespace mi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main() {
    const int W = 16, H = 4, N = W * H;
    std::vector<uint8_t> bgr(N * 3), y(N), u(N), v(N), back(N * 3);
    for (int i = 0; i < N * 3; ++i) bgr[i] = (uint8_t)((i * 37 + 11) % 251);
    uint8_t *d_bgr, *d_p, *d_out;
    CK(hipMalloc(&d_bgr, N * 3)); CK(hipMalloc(&d_p, N * 4)); CK(hipMalloc(&d_out, N * 3));
    CK(hipMemcpy(d_bgr, bgr.data(), N * 3, hipMemcpyHostToDevice));
    ColorJob j{}; j.src = d_bgr; j.rows = 1; j.row_px = N; j.src_step = j.dst_step = N * 3; j.p0 = d_p; j.p1 = d_p + N; j.p2 = d_p + 2 * N; j.plane_frame = N * 4;
    hipLaunchKernelGGL(color_kernel<2>, dim3(1, 1, 1), dim3(256), 0, 0, j);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y.data(), d_p, N, hipMemcpyDeviceToHost)); CK(hipMemcpy(u.data(), d_p + N, N, hipMemcpyDeviceToHost)); CK(hipMemcpy(v.data(), d_p + 2 * N, N, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        int Y = (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14, U = ((b - Y) * 8061 + (128 << 14) + 8192) >> 14, V = ((r - Y) * 14369 + (128 << 14) + 8192) >> 14;
        U = U < 0 ? 0 : U > 255 ? 255 : U; V = V < 0 ? 0 : V > 255 ? 255 : V;
        if (y[i] != Y || u[i] != U || v[i] != V) { if (bad < 8) printf("px %d: got %d %d %d want %d %d %d\n", i, y[i], u[i], v[i], Y, U, V); ++bad; }
    }
    printf("MODE2 bad %d\n", bad);
    ColorJob k{}; k.dst = d_out; k.rows = 1; k.row_px = N; k.src_step = k.dst_step = N * 3; k.p0 = d_p; k.p1 = d_p + N; k.p2 = d_p + 2 * N; k.plane_frame = N * 4;
    hipLaunchKernelGGL(color_kernel<3>, dim3(1, 1, 1), dim3(256), 0, 0, k);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(back.data(), d_out, N * 3, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < N; ++i) {
        int Y = y[i], U = u[i] - 128, V = v[i] - 128;
        int b = Y + ((U * 33292 + 8192) >> 14), g = Y + ((U * -6472 + V * -9519 + 8192) >> 14), r = Y + ((V * 18678 + 8192) >> 14);
        b = b < 0 ? 0 : b > 255 ? 255 : b; g = g < 0 ? 0 : g > 255 ? 255 : g; r = r < 0 ? 0 : r > 255 ? 255 : r;
        if (back[3 * i] != b || back[3 * i + 1] != g || back[3 * i + 2] != r) { if (bad < 8) printf("px %d: got %d %d %d want %d %d %d (yuv %d %d %d)\n", i, back[3 * i], back[3 * i + 1], back[3 * i + 2], b, g, r, y[i], u[i], v[i]); ++bad; }
    }
    printf("MODE3 bad %d\n", bad);
    return 0;
}
