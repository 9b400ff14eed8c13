// nv12_stream.cpp -- the reference's live pipeline with the media I/O removed (GStreamer is out of
// scope): frames come from a synthetic generator or a raw .nv12 file instead of
// `v4l2src ... appsink`, go through the same worker-pool + luma-op + NV12-rebuild stage, and leave
// in frame order to a sink (optional raw file) instead of `appsrc ... omxh264enc`.
//
// Mirrors, flag for flag where they still apply:
//   --workers N (reference: 1..8; here 1..64)  --width W  --height H  --fps F           OpenCVequalHist.cpp:262-284
//   --clipLimit C  --tile T                                      clahevideo.cpp:374-452
//   2 s status tick: in/out fps, queue depth, errors, backlog    OpenCVequalHist.cpp:200-234,
//                                                                OpenCLequalHist.cpp:439-508
// New: --op equalize|clahe|channels (channels = NV12 -> BGR -> equalizeHist per channel -> NV12), --uv fill128|copy (OpenCVequalHist.cpp:160-162 vs
// ColoropenCVCwqualHist.cpp:165), --frames N, --input/--output raw NV12 files, --paced, --depth D (frames a worker keeps in flight on
// its GPU), --uv-policy host|device (who writes the UV half), --no-pin (leave the frame ring pageable; by default it is registered
// once, the way a GstBufferPool's memory would be), --loop (rewind --input at its end: clahevideo.cpp:294-302), --dump-every K
// (write only every K-th delivered frame to --output), --no-numa-bind (do not bind each worker to the CPUs of its GPU's NUMA node;
// the binding is printed in the banner), --max-workers-per-gpu K (default 2; 0 = no cap).  --workers may exceed the GPU count
// (worker w -> GPU w mod N), up to 64; at most K of them are started per GPU.
#include <sched.h>
#include <sys/resource.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../mi_pool.hpp"

static bool kv(const char* arg, const char* key, std::string& v, int& i, int argc, char** argv)
{
    const std::string k = std::string("--") + key;
    if (k == arg && i + 1 < argc) { v = argv[++i]; return true; }            // --k v
    if (strncmp(arg, (k + "=").c_str(), k.size() + 1) == 0) { v = arg + k.size() + 1; return true; }   // --k=v
    return false;
}

int main(int argc, char** argv)
{
    using namespace micv;
    int workers = 1, width = 1920, height = 1080, fps = 60, frames = 600, tile = 8, depth = 0, per_gpu = 2;
    double clip = 2.0;
    bool paced = false, pin = true, loop = false, numa_bind = true;
    int dump_every = 1;
    std::string op = "equalize", uv = "fill128", uv_policy = "host", input, output, v;
    for (int i = 1; i < argc; ++i) {
        if (kv(argv[i], "workers", v, i, argc, argv)) workers = std::max(1, std::min(64, atoi(v.c_str())));    // reference: 1..8 (OpenCVequalHist.cpp:274)
        else if (kv(argv[i], "width", v, i, argc, argv)) width = atoi(v.c_str());
        else if (kv(argv[i], "height", v, i, argc, argv)) height = atoi(v.c_str());
        else if (kv(argv[i], "fps", v, i, argc, argv)) fps = atoi(v.c_str());
        else if (kv(argv[i], "frames", v, i, argc, argv)) frames = atoi(v.c_str());
        else if (kv(argv[i], "clipLimit", v, i, argc, argv)) clip = atof(v.c_str());
        else if (kv(argv[i], "tile", v, i, argc, argv)) tile = std::max(1, atoi(v.c_str()));
        else if (kv(argv[i], "op", v, i, argc, argv)) op = v;
        else if (kv(argv[i], "uv", v, i, argc, argv)) uv = v;
        else if (kv(argv[i], "input", v, i, argc, argv)) input = v;
        else if (kv(argv[i], "output", v, i, argc, argv)) output = v;
        else if (strcmp(argv[i], "--paced") == 0) paced = true;
        else if (kv(argv[i], "depth", v, i, argc, argv)) depth = atoi(v.c_str());
        else if (kv(argv[i], "uv-policy", v, i, argc, argv)) uv_policy = v;
        else if (strcmp(argv[i], "--pin") == 0) pin = true;          // (default) register the frame ring like a pinned GstBufferPool
        else if (strcmp(argv[i], "--no-pin") == 0) pin = false;
        else if (strcmp(argv[i], "--loop") == 0) loop = true;
        else if (strcmp(argv[i], "--no-numa-bind") == 0) numa_bind = false;
        else if (kv(argv[i], "max-workers-per-gpu", v, i, argc, argv)) per_gpu = atoi(v.c_str());    // 0 = no cap (measurements only)
        else if (kv(argv[i], "dump-every", v, i, argc, argv)) dump_every = std::max(1, atoi(v.c_str()));
        else fprintf(stderr, "Warning: ignoring unknown arg: %s\n", argv[i]);
    }
    if (width <= 0 || height <= 0 || frames <= 0) { fprintf(stderr, "bad size\n"); return 1; }
    const size_t fb = (size_t)width * height + (size_t)width * height / 2;
    if (depth <= 0) depth = fb >= ((size_t)8 << 20) ? 4 : 6;      // a pool worker (fed by the submitting thread): four 4K frames in flight, six of 1080p or less (profiles/r04_t_*)
    const int ring = 32;                                            // frames in flight (input + output ring)
    // every worker on one GPU (one worker, or a one-GPU process -- how bench.py runs one streamer per GPU): the submitting thread
    // and the frame ring it first-touches below belong next to that GPU as well
    std::string main_placement = "submitting thread: not bound (workers spread over several GPUs)";
    if (!numa_bind) main_placement = "submitting thread: NUMA binding off";
    else if (workers == 1 || getDeviceCount() == 1) {
        mi_numa_binding nb{};
        main_placement = std::string("submitting thread + frame ring: ") + (mi_thread_bind_near_device(0, &nb) == MI_OK ? nb.why : "not bound");
    }
    // Workers on SEVERAL GPUs in this one process: frame k goes to worker k mod N and lives in ring slot k mod 32, so when N divides 32
    // slot s always feeds the same worker -- its two frames are first-touched from a thread bound next to that worker's GPU.
    const int ndev = std::max(1, getDeviceCount());
    const int eff_workers = FramePool::workers_started(workers, ndev, per_gpu);
    std::vector<std::vector<unsigned char>> in(ring), out(ring);
    if (numa_bind && eff_workers > 1 && ndev > 1 && ring % eff_workers == 0) {
        cpu_set_t all;
        CPU_ZERO(&all);
        const bool have_all = sched_getaffinity(0, sizeof all, &all) == 0;
        for (int s = 0; s < ring; ++s) {
            mi_numa_binding nb{};
            (void)mi_thread_bind_near_device(FramePool::device_of_frame((uint64_t)s, eff_workers, ndev), &nb);     // ring % workers == 0: slot s only ever carries frames of that worker
            in[s].assign(fb, 0); out[s].assign(fb, 0);                                   // value-initialisation = first touch
            if (have_all) (void)sched_setaffinity(0, sizeof all, &all);
        }
        main_placement = "submitting thread: not bound; frame ring: each slot first-touched next to the GPU of the worker it feeds";
    } else {
        for (int s = 0; s < ring; ++s) { in[s].assign(fb, 0); out[s].assign(fb, 0); }
    }
    FILE* fin = input.empty() ? nullptr : fopen(input.c_str(), "rb");
    FILE* fout = output.empty() ? nullptr : fopen(output.c_str(), "wb");
    if (!input.empty() && !fin) { fprintf(stderr, "cannot open %s\n", input.c_str()); return 1; }
    uint64_t seed = 0x5EED0000;
    auto synth = [&](std::vector<unsigned char>& f, int k) {
        uint64_t s = seed + k;
        for (size_t i = 0; i < fb; i += 8) {
            s += 0x9E3779B97F4A7C15ull; uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            for (int b = 0; b < 8 && i + b < fb; ++b) f[i + b] = (unsigned char)(64 + ((z >> (8 * b)) & 0x7f));
        }
    };
    std::vector<std::chrono::steady_clock::time_point> t_submit((size_t)frames);
    std::vector<float> latency_ms((size_t)frames, -1.f);
    try {
        if (pin) for (int k = 0; k < ring; ++k) { registerHostBuffer(in[k].data(), fb); registerHostBuffer(out[k].data(), fb); }
        std::atomic<uint64_t> delivered{0};
        FramePool pool(workers, width, height, op == "clahe" ? FramePool::CLAHE_OP : (op == "channels" ? FramePool::CHANNELS_EQ : FramePool::EQUALIZE),
                       uv == "copy" ? UV_COPY : UV_FILL128,
                       [&](const FrameJob& j) {
                           if (!j.ok) fprintf(stderr, "frame %llu error: %s\n", (unsigned long long)j.index, j.error.c_str());
                           else if (fout && j.index % (uint64_t)dump_every == 0) fwrite(j.out, 1, fb, fout);
                           if (j.index < latency_ms.size())
                               latency_ms[j.index] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_submit[j.index]).count();
                           delivered.fetch_add(1);
                       },
                       clip, Size(tile, tile), ring / (size_t)workers > 2 ? ring / (size_t)workers - 1 : 1, depth,
                       uv_policy == "device" ? MI_PIPE_UV_DEVICE : MI_PIPE_UV_HOST, numa_bind, per_gpu);
        printf("nv12_stream: %dx%d %s uv=%s (uv-policy %s) workers=%d depth=%d gpus=%d frames=%d%s%s\n", width, height, op.c_str(), uv.c_str(),
               uv_policy.c_str(), pool.workers(), depth, getDeviceCount(), frames, paced ? " paced" : "", pin ? " pinned-ring" : " pageable-ring");
        if (pool.workers() != pool.requested())
            printf("workers: %d requested, %d started (at most %d per GPU: one keeps the link busy, more never help)\n", pool.requested(), pool.workers(), per_gpu);
        printf("placement: %s\n", main_placement.c_str());
        for (const std::string& line : pool.placement()) printf("placement: %s\n", line.c_str());
        // synthetic source: the ring's frames exist before the clock starts (a camera / decoder hands over finished frames;
        // generating 12 MB of noise per frame on the submitting thread would otherwise be the slowest stage of the first lap)
        if (!fin) for (int k = 0; k < ring && k < frames; ++k) synth(in[k], k);
        // what the stream costs the HOST: CPU time of the whole process (submitting thread, workers, the library's helper threads, the
        // runtime's own threads) between the first submit and the last delivery -- the cores a capture / encode stage would not get
        auto cpu_seconds = [] {
            rusage ru{};
            getrusage(RUSAGE_SELF, &ru);
            return std::pair<double, double>(ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6);
        };
        const auto cpu0 = cpu_seconds();
        const auto t0 = std::chrono::steady_clock::now();
        auto last_tick = t0;
        uint64_t last_out = 0;
        for (int k = 0; k < frames; ++k) {
            // a ring slot may be reused only after its frame was delivered
            while (delivered.load() + ring <= (uint64_t)k) std::this_thread::sleep_for(std::chrono::microseconds(50));
            auto& f = in[k % ring];
            if (fin) {
                size_t got = fread(f.data(), 1, fb, fin);
                if (got != fb && loop && k > 0) { rewind(fin); got = fread(f.data(), 1, fb, fin); }     // --loop: seek to 0 at end of input
                if (got != fb) { frames = k; break; }
            }
            if (paced) std::this_thread::sleep_until(t0 + std::chrono::microseconds((int64_t)k * 1000000 / fps));
            t_submit[k] = std::chrono::steady_clock::now();
            pool.submit(f.data(), out[k % ring].data());
            const auto now = std::chrono::steady_clock::now();
            if (now - last_tick >= std::chrono::seconds(2)) {        // status tick (OpenCVequalHist.cpp:200-234)
                const uint64_t o = pool.stats().frames_out.load();
                const double dt = std::chrono::duration<double>(now - last_tick).count();
                const size_t q = pool.queue_depth();
                printf("[status] in=%llu out=%llu out_fps=%.1f queue=%zu errors=%llu%s\n", (unsigned long long)pool.stats().frames_in.load(),
                       (unsigned long long)o, (o - last_out) / dt, q, (unsigned long long)pool.stats().processing_errors.load(),
                       q > 5 ? "  QUEUE BACKLOG" : "");
                last_tick = now; last_out = o;
            }
        }
        pool.finish();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const auto cpu1 = cpu_seconds();
        printf("done: %llu frames in %.3f s = %.1f frames/s (host NV12 in -> host NV12 out, PCIe inclusive), errors=%llu\n",
               (unsigned long long)pool.stats().frames_out.load(), el, pool.stats().frames_out.load() / el,
               (unsigned long long)pool.stats().processing_errors.load());
        {
            const double n = (double)std::max<uint64_t>(1, pool.stats().frames_out.load());
            printf("worker time per frame (us, summed over %d worker(s)): submit %.1f, wait %.1f, deliver %.1f, idle %.1f\n", pool.workers(),
                   pool.stats().ns_submit.load() / n / 1e3, pool.stats().ns_wait.load() / n / 1e3, pool.stats().ns_deliver.load() / n / 1e3,
                   pool.stats().ns_idle.load() / n / 1e3);
        }
        {
            const double n = (double)std::max<uint64_t>(1, pool.stats().frames_out.load());
            const double us = cpu1.first - cpu0.first, ss = cpu1.second - cpu0.second;
            printf("host CPU: %.1f us per frame (user %.1f + sys %.1f), %.2f host cores busy on average over the %.3f s (whole process, getrusage)\n",
                   (us + ss) / n * 1e6, us / n * 1e6, ss / n * 1e6, (us + ss) / el, el);
        }
        // submit -> in-order delivery latency (the reference only prints averages: clahevideo.cpp:54-84)
        std::vector<float> lat;
        for (int k = 0; k < frames; ++k) if (latency_ms[k] >= 0.f) lat.push_back(latency_ms[k]);
        if (!lat.empty()) {
            std::sort(lat.begin(), lat.end());
            const double budget = 1000.0 / fps;
            size_t late = 0;
            for (float v : lat) if (v > budget) ++late;
            printf("latency ms (submit -> in-order delivery): p50=%.3f p90=%.3f p99=%.3f max=%.3f; frames over the %.2f ms frame budget: %zu%s\n",
                   lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat[std::min(lat.size() - 1, lat.size() * 99 / 100)], lat.back(), budget, late,
                   paced ? "" : "  (unpaced: latency includes queueing)");
        }
        if (pin)                                                  // every frame was delivered (pool.finish()): nothing is pending on the ring
            for (int k = 0; k < ring; ++k) {
                const mi_status si = tryUnregisterHostBuffer(in[k].data());     // both halves of the slot, whatever the first one answers;
                const mi_status so = tryUnregisterHostBuffer(out[k].data());    // clean-up path: the non-throwing form
                if (si != MI_OK || so != MI_OK) fprintf(stderr, "frame ring slot %d not unpinned: in %s, out %s\n", k, mi_status_str(si), mi_status_str(so));
            }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    if (fin) fclose(fin);
    if (fout) fclose(fout);
    return 0;
}
