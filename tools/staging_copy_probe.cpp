// staging_copy_probe.cpp -- how fast can the CPU copy a 4K plane between pageable and pinned memory, and do more threads help?
//     /opt/rocm/bin/hipcc -O2 -pthread -o tools/staging_copy_probe tools/staging_copy_probe.cpp && tools/staging_copy_probe
// Why it exists: since late round 2 the library packs unpinned host memory through pinned staging buffers it owns;
// a helper thread for those copies was considered and dropped -- on the GPU box one thread already moves a (cache-resident) 8.3 MB
// plane in 168 us (pageable -> pinned) / 106 us (pinned -> pageable), two threads in 118 / 100 us: what the staged host form loses
// against handing pageable memory to the runtime (0.56-0.66 ms vs 0.35 ms per synchronous 4K call) is the chunked
// copy -> DMA -> copy pipeline's fill and drain, not memcpy bandwidth.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
int main() {
    const size_t n = 3840 * 2160;
    std::vector<unsigned char> src(n, 7), dst(n);
    void* pin = nullptr; hipHostMalloc(&pin, n, hipHostMallocDefault);
    memset(pin, 1, n);
    auto t = [&](int threads, bool to_pin) {
        double best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int k = 1; k < threads; ++k) th.emplace_back([&, k] { size_t a = n * k / threads, b = n * (k + 1) / threads; if (to_pin) memcpy((char*)pin + a, src.data() + a, b - a); else memcpy(dst.data() + a, (char*)pin + a, b - a); });
            { size_t b = n / threads; if (to_pin) memcpy(pin, src.data(), b); else memcpy(dst.data(), pin, b); }
            for (auto& x : th) x.join();
            best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%s %d thread(s): %.0f us (%.1f GB/s)\n", to_pin ? "pageable->pinned" : "pinned->pageable", threads, best * 1e6, n / best / 1e9);
    };
    for (int th : {1, 2, 4}) { t(th, true); t(th, false); }
    return 0;
}
