// test_pool_multidev.cpp -- micv::FramePool (cxx/mi_pool.hpp) with MORE THAN ONE device, on the CPU.
//
// Every GPU box this project can reach has one GPU, so until round 5 the pool's multi-device branches -- worker w -> GPU w mod N,
// at most two workers per GPU, one context per worker created on the worker's own thread after it placed itself, the in-order
// re-sequencer under out-of-order completion, the per-frame drop-and-count with a device that fails -- had only ever run with
// getDeviceCount() == 1.  Here the C ABI underneath the pool is the test-only stand-in of stub_mi_lumaeq.hpp (2, 3 or 8 "devices",
// randomised completion delays, injected failures); the pool itself is the shipped header, unmodified.
// Reference: the 1..8-worker pool of OpenCVequalHist.cpp:274, :397-402, its unordered push (:183) and its drop-and-count (:183-193).
// Built plain and under ThreadSanitizer (tests/test_pool_multidev.py).
#include "stub_mi_lumaeq.hpp"

#include "../../opencv-opencl_amd/cxx/mi_pool.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

static int g_fail = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

namespace {

constexpr int W = 64, H = 32;
constexpr size_t FB = (size_t)W * H * 3 / 2;

struct Delivered { uint64_t index; bool ok; int device; std::string error; bool bytes_ok; };

struct Run {
    std::vector<Delivered> got;                 // written by the sink (one pool thread at a time), read after finish()
    std::vector<std::vector<unsigned char>> in, out;
    int workers_started = 0, requested = 0;
    std::vector<std::string> placement;
    uint64_t frames_in = 0, frames_out = 0, errors = 0;
};

// n_frames through a pool of `workers` on the stub world as it is configured now
Run run_pool(int workers, int n_frames, int ring = 24, size_t max_queue = 4, int depth = 0, int sink_sleep_us = 0, int max_per_gpu = 2)
{
    Run r;
    r.in.assign(ring, std::vector<unsigned char>(FB));
    r.out.assign(ring, std::vector<unsigned char>(FB));
    for (int k = 0; k < ring; ++k)
        for (size_t i = 0; i < FB; ++i) r.in[k][i] = (unsigned char)(k * 31 + i * 7);
    std::mutex ring_mu;
    std::condition_variable ring_cv;
    std::vector<char> busy(ring, 0);
    micv::FramePool pool(workers, W, H, micv::FramePool::EQUALIZE, micv::UV_FILL128,
        [&](const micv::FrameJob& j) {
            Delivered d{j.index, j.ok, -1, j.error, false};
            const int slot = (int)(j.index % (uint64_t)ring);
            if (j.ok) {
                uint64_t tag = 0; int32_t dev = -1;
                std::memcpy(&tag, j.out, 8);
                std::memcpy(&dev, j.out + 8, 4);
                d.device = dev;
                d.bytes_ok = tag == j.index && j.out == r.out[slot].data() && j.in == r.in[slot].data();
                for (size_t i = 16; i < 64; ++i) d.bytes_ok = d.bytes_ok && j.out[i] == (unsigned char)(j.in[i] + 1);
            }
            r.got.push_back(d);
            if (sink_sleep_us) std::this_thread::sleep_for(std::chrono::microseconds(sink_sleep_us));
            std::lock_guard<std::mutex> lk(ring_mu);
            busy[slot] = 0;                                  // the slot may be reused only after delivery (caller-owned until then)
            ring_cv.notify_all();
        },
        2.0, micv::Size(8, 8), max_queue, depth, MI_PIPE_UV_AUTO, true, max_per_gpu);
    r.workers_started = pool.workers();
    r.requested = pool.requested();
    r.placement = pool.placement();
    for (int k = 0; k < n_frames; ++k) {
        const int slot = k % ring;
        {
            std::unique_lock<std::mutex> lk(ring_mu);
            ring_cv.wait(lk, [&] { return !busy[slot]; });
            busy[slot] = 1;
        }
        std::memset(r.out[slot].data(), 0, 64);
        const uint64_t idx = pool.submit(r.in[slot].data(), r.out[slot].data());
        CHECK(idx == (uint64_t)k);
    }
    pool.finish();
    r.frames_in = pool.stats().frames_in.load();
    r.frames_out = pool.stats().frames_out.load();
    r.errors = pool.stats().processing_errors.load();
    CHECK(pool.queue_depth() == 0);
    return r;
}

void check_in_order(const Run& r, int n_frames)
{
    CHECK((int)r.got.size() == n_frames);
    for (size_t k = 0; k < r.got.size(); ++k) CHECK(r.got[k].index == k);        // strictly in frame order, none missing, none twice
    CHECK(r.frames_in == (uint64_t)n_frames && r.frames_out == (uint64_t)n_frames);
}

int inversions_at_the_devices()
{
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    int inv = 0;
    for (size_t k = 1; k < w.completions.size(); ++k) inv += w.completions[k].tag < w.completions[k - 1].tag;
    return inv;
}

// ---- 1: eight devices, eight workers: worker w -> device w, frame k -> device k mod 8, delivered in order --------------------------
void test_eight_devices_eight_workers()
{
    stub::world().reset(8);
    const int N = 400;
    Run r = run_pool(8, N);
    check_in_order(r, N);
    CHECK(r.workers_started == 8 && r.requested == 8 && r.errors == 0);
    for (const auto& d : r.got) {
        CHECK(d.ok && d.bytes_ok);
        CHECK(d.device == (int)(d.index % 8));              // frame k -> worker k mod N -> GPU (worker mod GPUs)
    }
    CHECK(r.placement.size() == 8);
    for (int w = 0; w < 8; ++w) {
        const std::string want = "worker " + std::to_string(w) + " -> GPU " + std::to_string(w) + ": stub: GPU " + std::to_string(w);
        CHECK(r.placement[w].compare(0, want.size(), want) == 0);
    }
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    CHECK(w.peak_pipes == 8 && w.live_pipes == 0);
    for (int d = 0; d < 8; ++d) {
        CHECK(w.ctx_created[d] == 1 && w.peak_ctx[d] == 1 && w.live_ctx[d] == 0);      // one context per worker, destroyed at finish()
        CHECK(w.frames_by_device[d] == (uint64_t)N / 8);
    }
    // every worker placed itself next to ITS device and created its context on the same thread, after placing itself
    CHECK(w.bound.size() == 8);
    for (const auto& b : w.bound) {
        auto it = w.ctx_by_thread.find(b.first);
        CHECK(it != w.ctx_by_thread.end() && it->second.size() == 1 && *it->second.begin() == b.second);
    }
}

// ---- 2: the re-sequencer really had something to do: completion at the devices was out of order ---------------------------------
void test_out_of_order_completion_is_resequenced()
{
    stub::world().reset(8);
    {
        std::lock_guard<std::mutex> lk(stub::world().mu);
        stub::world().slow_devices = {0, 3};                  // frames 0, 3, 8, 11, ... finish long after their successors
        stub::world().max_delay_us = 800;
    }
    const int N = 240;
    Run r = run_pool(8, N, 40, 4);
    check_in_order(r, N);
    CHECK(r.errors == 0);
    for (const auto& d : r.got) CHECK(d.ok && d.bytes_ok && d.device == (int)(d.index % 8));
    const int inv = inversions_at_the_devices();
    CHECK(inv > 10);                                          // the devices did NOT finish in frame order ...
    printf("  out-of-order completions at the devices: %d of %d (delivered strictly in order)\n", inv, N);
}

// ---- 3: more workers asked for than two per device -------------------------------------------------------------------------------
void test_at_most_two_workers_per_device()
{
    stub::world().reset(8);
    Run r = run_pool(64, 320, 48, 4);
    check_in_order(r, 320);
    CHECK(r.requested == 64 && r.workers_started == 16 && r.placement.size() == 16);
    for (int w = 0; w < 16; ++w) {
        const std::string want = "worker " + std::to_string(w) + " -> GPU " + std::to_string(w % 8) + ":";
        CHECK(r.placement[w].compare(0, want.size(), want) == 0);
    }
    for (const auto& d : r.got) CHECK(d.ok && d.bytes_ok && d.device == (int)((d.index % 16) % 8));
    {
        auto& w = stub::world();
        std::lock_guard<std::mutex> lk(w.mu);
        for (int d = 0; d < 8; ++d) CHECK(w.peak_ctx[d] == 2 && w.ctx_created[d] == 2);
        CHECK(w.peak_pipes == 16);
    }
    // three devices, eight workers asked for: six started, worker w -> device w mod 3
    stub::world().reset(3);
    Run q = run_pool(8, 120);
    check_in_order(q, 120);
    CHECK(q.workers_started == 6);
    for (const auto& d : q.got) CHECK(d.ok && d.device == (int)((d.index % 6) % 3));
    // one worker per device when the caller says so
    stub::world().reset(8);
    Run o = run_pool(64, 64, 24, 4, 0, 0, 1);
    CHECK(o.workers_started == 8);
    check_in_order(o, 64);
    // two devices, one worker: device 1 stays idle
    stub::world().reset(2);
    Run s = run_pool(1, 30);
    check_in_order(s, 30);
    for (const auto& d : s.got) CHECK(d.ok && d.device == 0);
}

// ---- 4: a device whose submits fail: drop-and-count, frame by frame, order kept (OpenCVequalHist.cpp:183-193) -----------------------
void test_device_that_fails_submit()
{
    stub::world().reset(8);
    { std::lock_guard<std::mutex> lk(stub::world().mu); stub::world().submit_fails = {3}; }
    const int N = 200;
    Run r = run_pool(8, N);
    check_in_order(r, N);
    uint64_t bad = 0;
    for (const auto& d : r.got) {
        if (d.index % 8 == 3) {
            ++bad;
            CHECK(!d.ok && d.error.find("mi_pipe_submit: MI_ERR_HIP") != std::string::npos && d.error.find("device 3") != std::string::npos);
        } else {
            CHECK(d.ok && d.bytes_ok && d.device == (int)(d.index % 8));
        }
    }
    CHECK(bad == (uint64_t)N / 8 && r.errors == bad);
}

// ---- 5: a device whose context cannot be created: the pool comes up, its frames are reported, finish() returns ----------------------
void test_device_that_fails_ctx_create()
{
    stub::world().reset(8);
    { std::lock_guard<std::mutex> lk(stub::world().mu); stub::world().ctx_fails = {5}; }
    const int N = 160;
    Run r = run_pool(16, N, 48);                                // workers 5 and 13 sit on the dead device
    check_in_order(r, N);
    CHECK(r.workers_started == 16);
    uint64_t bad = 0;
    for (const auto& d : r.got) {
        const int worker = (int)(d.index % 16);
        if (worker % 8 == 5) {
            ++bad;
            CHECK(!d.ok && d.error.find("mi_ctx_create(device=5) failed: MI_ERR_HIP") != std::string::npos);
        } else {
            CHECK(d.ok && d.bytes_ok && d.device == worker % 8);
        }
    }
    CHECK(bad == (uint64_t)N / 8 && r.errors == bad);
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    CHECK(w.ctx_created[5] == 0 && w.peak_pipes == 14 && w.live_pipes == 0);
}

// ---- 6: a device whose waits fail now and then: the frame is still retired, counted, delivered in its place -------------------------
void test_device_that_fails_wait()
{
    stub::world().reset(4);
    { std::lock_guard<std::mutex> lk(stub::world().mu); stub::world().wait_fails = {1}; stub::world().wait_fail_every = 3; }
    const int N = 240;
    Run r = run_pool(4, N);
    check_in_order(r, N);
    uint64_t bad = 0;
    for (const auto& d : r.got) {
        if (!d.ok) {
            ++bad;
            CHECK(d.index % 4 == 1 && d.error.find("mi_pipe_wait: MI_ERR_HIP") != std::string::npos);
        } else {
            CHECK(d.bytes_ok && d.device == (int)(d.index % 4));
        }
    }
    CHECK(bad == (uint64_t)(N / 4) / 3 && r.errors == bad);
    std::lock_guard<std::mutex> lk(stub::world().mu);
    CHECK(stub::world().wait_failures == bad);
}

// ---- 7: back-pressure with a slow sink and a short queue; several failures at once; every device dead -------------------------------
void test_backpressure_and_everything_failing()
{
    stub::world().reset(8);
    { std::lock_guard<std::mutex> lk(stub::world().mu); stub::world().submit_fails = {2}; stub::world().ctx_fails = {6}; stub::world().wait_fails = {0};
      stub::world().wait_fail_every = 2; stub::world().slow_devices = {7}; }
    Run r = run_pool(8, 160, 12, 1, 2, 300);                    // ring of 12 < workers x depth: the submitter really blocks
    check_in_order(r, 160);
    for (const auto& d : r.got) {
        const int dev = (int)(d.index % 8);
        if (dev == 2 || dev == 6) CHECK(!d.ok);
        else if (dev != 0) CHECK(d.ok && d.bytes_ok && d.device == dev);
    }
    CHECK(r.errors == 20 + 20 + 10);
    stub::world().reset(2);
    { std::lock_guard<std::mutex> lk(stub::world().mu); stub::world().ctx_fails = {0, 1}; }
    Run q = run_pool(4, 50);
    check_in_order(q, 50);
    CHECK(q.errors == 50);
    for (const auto& d : q.got) CHECK(!d.ok && !d.error.empty());
    // no device at all: the constructor throws (no CPU fallback), as with the real library
    stub::world().reset(0);
    bool threw = false;
    try {
        micv::FramePool p(2, W, H, micv::FramePool::EQUALIZE, micv::UV_FILL128, nullptr);
    } catch (const std::exception& e) {
        threw = std::string(e.what()).find("no HIP device") != std::string::npos;
    }
    CHECK(threw);
}

// ---- 8: finish() twice, destructor after finish, a pool that never got a frame ------------------------------------------------------
void test_lifecycle()
{
    stub::world().reset(8);
    {
        micv::FramePool p(8, W, H, micv::FramePool::CLAHE_OP, micv::UV_COPY, nullptr);
        CHECK(p.workers() == 8);
        p.finish();
        p.finish();
    }
    {
        micv::FramePool p(3, W, H, micv::FramePool::EQUALIZE, micv::UV_FILL128, nullptr);      // destructor alone
    }
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    CHECK(w.live_pipes == 0);
    for (int d = 0; d < 8; ++d) CHECK(w.live_ctx[d] == 0);
}

}  // namespace

int main()
{
    test_eight_devices_eight_workers();
    test_out_of_order_completion_is_resequenced();
    test_at_most_two_workers_per_device();
    test_device_that_fails_submit();
    test_device_that_fails_ctx_create();
    test_device_that_fails_wait();
    test_backpressure_and_everything_failing();
    test_lifecycle();
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("pool multi-device ok\n");
    return 0;
}
