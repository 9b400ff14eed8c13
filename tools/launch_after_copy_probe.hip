// How long does hipLaunchKernel take on a stream that has just been told to wait for an event recorded behind a
// host->device copy on ANOTHER stream (what mi_pipe_submit does per frame), compared with the same sequence on one stream?
//     hipcc --offload-arch=gfx950 -O2 -o tools/launch_after_copy_probe tools/launch_after_copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
__global__ void touch(unsigned char* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
static double us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 3840 * 2160;
    void *d[8], *h;
    for (auto& p : d) CK(hipMalloc(&p, n));
    CK(hipHostMalloc(&h, n, hipHostMallocDefault)); memset(h, 1, n);
    hipStream_t sc, sk, sd; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(64); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int mode = 0; mode < 4; ++mode) {
        // 0: copy on sc, event, sk waits, launch on sk        (the pipe)
        // 1: copy and launch on the same stream sk            (no cross-stream wait)
        // 2: kernel on sc, event, sk waits, launch on sk      (cross-stream wait on a KERNEL)
        // 3: as 0, plus D2H on sd waiting for the kernel      (the whole frame of the pipe)
        double t_copy = 0, t_rec = 0, t_wait = 0, t_launch = 0, t_rest = 0; int reps = 200;
        CK(hipDeviceSynchronize());
        const double t0 = us();
        for (int r = 0; r < reps; ++r) {
            void* dd = d[r & 7];
            double a = us();
            if (mode == 2) hipLaunchKernelGGL(touch, dim3(32400), dim3(256), 0, sc, (unsigned char*)dd, n);
            else CK(hipMemcpyAsync(dd, h, n, hipMemcpyHostToDevice, mode == 1 ? sk : sc));
            double b = us(); t_copy += b - a;
            if (mode != 1) { CK(hipEventRecord(ev[r & 63], sc)); double c = us(); t_rec += c - b; CK(hipStreamWaitEvent(sk, ev[r & 63], 0)); b = us(); t_wait += b - c; }
            hipLaunchKernelGGL(touch, dim3(32400), dim3(256), 0, sk, (unsigned char*)dd, n);
            double e = us(); t_launch += e - b;
            if (mode == 3) {
                CK(hipEventRecord(ev[(r + 32) & 63], sk)); CK(hipStreamWaitEvent(sd, ev[(r + 32) & 63], 0));
                CK(hipMemcpyAsync(h, dd, n, hipMemcpyDeviceToHost, sd));
                t_rest += us() - e;
            }
            if ((r & 3) == 3) { CK(hipStreamSynchronize(sk)); if (mode == 3) CK(hipStreamSynchronize(sd)); }     // four frames in flight at most
        }
        CK(hipDeviceSynchronize());
        const double wall = us() - t0;
        printf("mode %d: per frame: copy-call %.1f us, eventRecord %.1f, streamWaitEvent %.1f, hipLaunchKernel %.1f, rest %.1f; wall %.1f us/frame\n",
               mode, t_copy / reps, t_rec / reps, t_wait / reps, t_launch / reps, t_rest / reps, wall / reps);
    }
    return 0;
}
