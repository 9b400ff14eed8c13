// N host threads, each with its own stream and pinned buffers: H2D (8.3 MB) -> tiny kernel -> D2H -> sync, in a loop.
// Aggregate frames/s = the ceiling of any frame-per-worker host pipeline on this box.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void touch(unsigned char* p) { p[threadIdx.x] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t n = 3840 * 2160;
    for (int T : {1, 2, 4, 8}) {
        for (int mode = 0; mode < 2; ++mode) {          // 0: stream sync per frame; 1: two frames in flight per thread (double-buffered)
            std::atomic<long> frames{0};
            std::atomic<bool> stop{false};
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                hipSetDevice(0);
                hipStream_t s[2]; void *d[2], *hi[2], *ho[2]; hipEvent_t ev[2];
                for (int k = 0; k < 2; ++k) {
                    hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking); hipMalloc(&d[k], n);
                    hipHostMalloc(&hi[k], n, hipHostMallocDefault); hipHostMalloc(&ho[k], n, hipHostMallocDefault);
                    hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
                }
                int k = 0; bool pending[2] = {false, false};
                while (!stop.load()) {
                    if (mode == 1 && pending[k]) { hipEventSynchronize(ev[k]); frames.fetch_add(1); }
                    hipMemcpyAsync(d[k], hi[k], n, hipMemcpyHostToDevice, s[k]);
                    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s[k], (unsigned char*)d[k]);
                    hipMemcpyAsync(ho[k], d[k], n, hipMemcpyDeviceToHost, s[k]);
                    if (mode == 0) { hipStreamSynchronize(s[k]); frames.fetch_add(1); }
                    else { hipEventRecord(ev[k], s[k]); pending[k] = true; k ^= 1; }
                }
                hipDeviceSynchronize();
            });
            std::this_thread::sleep_for(std::chrono::milliseconds(300));
            const long f0 = frames.load(); const double t0 = now();
            std::this_thread::sleep_for(std::chrono::milliseconds(1500));
            const long f1 = frames.load(); const double t1 = now();
            stop.store(true);
            for (auto& x : th) x.join();
            printf("threads=%d %s: %.0f frames/s (%.1f GB/s per direction)\n", T, mode ? "2 in flight/thread" : "sync per frame     ",
                   (f1 - f0) / (t1 - t0), (f1 - f0) / (t1 - t0) * n / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
