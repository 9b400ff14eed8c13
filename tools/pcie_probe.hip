// Raw copy-engine ceilings for one 4K Y plane: hipHostMalloc'd vs hipHostRegister'd vs pageable host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 3840 * 2160;
    void *d_in, *d_out, *h_m_in, *h_m_out;
    CK(hipMalloc(&d_in, n)); CK(hipMalloc(&d_out, n));
    CK(hipHostMalloc(&h_m_in, n, hipHostMallocDefault)); CK(hipHostMalloc(&h_m_out, n, hipHostMallocDefault));
    void* h_r_in = aligned_alloc(4096, n); void* h_r_out = aligned_alloc(4096, n);
    memset(h_r_in, 1, n); memset(h_r_out, 1, n);
    CK(hipHostRegister(h_r_in, n, hipHostRegisterPortable)); CK(hipHostRegister(h_r_out, n, hipHostRegisterPortable));
    void* h_p_in = aligned_alloc(4096, n); void* h_p_out = aligned_alloc(4096, n);
    memset(h_p_in, 1, n); memset(h_p_out, 1, n);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct { const char* name; void* in; void* out; } cases[] = {{"hipHostMalloc", h_m_in, h_m_out}, {"hipHostRegister", h_r_in, h_r_out}, {"pageable", h_p_in, h_p_out}};
    for (auto& c : cases) {
        for (int mode = 0; mode < 3; ++mode) {               // 0: H2D, 1: D2H, 2: H2D then D2H then sync (one frame, serial)
            const int reps = 50;
            for (int w = 0; w < 3; ++w) { CK(hipMemcpyAsync(d_in, c.in, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
            const double t0 = now();
            for (int r = 0; r < reps; ++r) {
                if (mode == 0 || mode == 2) CK(hipMemcpyAsync(d_in, c.in, n, hipMemcpyHostToDevice, s));
                if (mode == 1 || mode == 2) CK(hipMemcpyAsync(c.out, d_out, n, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
            }
            const double t = (now() - t0) / reps;
            printf("%-16s %-9s %.3f ms  %.1f GB/s\n", c.name, mode == 0 ? "h2d" : mode == 1 ? "d2h" : "h2d+d2h", t * 1e3, (mode == 2 ? 2 : 1) * n / t / 1e9);
        }
    }
    // host memset of a UV plane and memcpy of a Y plane (what the staging path pays on the CPU)
    double t0 = now(); for (int r = 0; r < 20; ++r) memset((char*)h_p_out, 128, n / 2); printf("memset UV %.3f ms\n", (now() - t0) / 20 * 1e3);
    t0 = now(); for (int r = 0; r < 20; ++r) memcpy(h_m_in, h_p_in, n); printf("memcpy Y -> pinned %.3f ms\n", (now() - t0) / 20 * 1e3);
    t0 = now(); for (int r = 0; r < 20; ++r) memcpy(h_p_out, h_m_out, n); printf("memcpy pinned -> Y %.3f ms\n", (now() - t0) / 20 * 1e3);
    return 0;
}
