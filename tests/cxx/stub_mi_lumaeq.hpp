// stub_mi_lumaeq.hpp -- TEST-ONLY stand-in for the part of the C ABI (include/mi_lumaeq.h) that cxx/mi_pool.hpp binds, so that
// micv::FramePool can be given two, three or eight "devices" on a machine that has none.  Never part of the product tree, never
// linked into libmi_lumaeq.so; it computes nothing that anybody ships (the "op" is out[i] = in[i] + 1 on a handful of bytes plus a
// stamp saying which device and which thread did it).
//
// What it models of the real library, because FramePool's branches depend on it:
//   * mi_device_count() = StubWorld::devices
//   * mi_ctx_create(device): fails with MI_ERR_HIP on devices in `ctx_fails`; counts live contexts per device (and the peak)
//   * one mi_pipe per context, `depth` frames in flight, MI_ERR_BUSY beyond that; completion strictly in submission order PER PIPE,
//     after a pseudo-random delay per frame (so pipes on different devices finish out of order with respect to one another)
//   * mi_pipe_submit fails (occupying no slot) on devices in `submit_fails`; mi_pipe_wait fails on every `wait_fail_every`-th wait of
//     devices in `wait_fails` -- and has still retired the frame, as include/mi_lumaeq.h promises
//   * mi_thread_bind_near_device(device) records which thread asked for which device
// The reference analogue of what is being tested: the worker pool of OpenCVequalHist.cpp:102-196, :397-402 and its per-frame
// drop-and-count (:183-193).
#ifndef STUB_MI_LUMAEQ_HPP_
#define STUB_MI_LUMAEQ_HPP_

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi_lumaeq.h"

namespace stub {

struct Completion { uint64_t tag; int device; };

struct World {
    std::mutex mu;
    int devices = 8;
    std::set<int> ctx_fails, submit_fails, wait_fails;
    int wait_fail_every = 3;
    unsigned max_delay_us = 1500;
    std::vector<int> slow_devices;                         // these take 4x as long per frame
    // observations
    std::map<int, int> live_ctx, peak_ctx, ctx_created;    // per device
    std::map<std::thread::id, int> bound;                  // thread -> device it asked to be placed next to
    std::map<std::thread::id, std::set<int>> ctx_by_thread;
    std::vector<Completion> completions;                   // in the order the "devices" finished frames
    std::map<int, uint64_t> frames_by_device;
    int live_pipes = 0, peak_pipes = 0;
    uint64_t submit_failures = 0, wait_failures = 0;
    std::map<void*, size_t> registered;                    // mi_host_register
    std::vector<std::pair<std::thread::id, int>> bind_log; // every mi_thread_bind_near_device call, in order

    World()
    {   // a program built against the stub (nv12_stream in tests/test_pool_multidev.py) is configured through the environment
        if (const char* e = std::getenv("MI_STUB_DEVICES")) devices = std::atoi(e);
        if (const char* e = std::getenv("MI_STUB_SUBMIT_FAILS")) submit_fails.insert(std::atoi(e));
        if (const char* e = std::getenv("MI_STUB_CTX_FAILS")) ctx_fails.insert(std::atoi(e));
    }

    void reset(int ndev)
    {
        std::lock_guard<std::mutex> lk(mu);
        devices = ndev;
        ctx_fails.clear(); submit_fails.clear(); wait_fails.clear(); slow_devices.clear();
        wait_fail_every = 3; max_delay_us = 1500;
        live_ctx.clear(); peak_ctx.clear(); ctx_created.clear(); bound.clear(); ctx_by_thread.clear(); completions.clear();
        frames_by_device.clear(); live_pipes = peak_pipes = 0; submit_failures = wait_failures = 0;
        registered.clear(); bind_log.clear();
    }
};
inline World& world() { static World w; return w; }

// the bytes a frame must carry after "device" d processed tag t: checked by the sink
inline void stamp(unsigned char* out, const unsigned char* in, size_t bytes, uint64_t tag, int device)
{
    const size_t n = bytes < 64 ? bytes : 64;
    for (size_t i = 16; i < n; ++i) out[i] = (unsigned char)(in[i] + 1);
    std::memcpy(out, &tag, 8);
    const int32_t d = device;
    std::memcpy(out + 8, &d, 4);
}

}  // namespace stub

struct mi_ctx {
    int device = 0;
    std::string last_err = "no error";
    bool has_pipe = false;
};

struct mi_pipe {
    mi_ctx* ctx = nullptr;
    mi_pipe_config cfg{};
    struct Slot { const uint8_t* in; uint8_t* out; uint64_t tag; std::chrono::steady_clock::time_point ready; };
    std::deque<Slot> pending;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    uint64_t waits = 0;
    std::chrono::steady_clock::time_point engine_free = std::chrono::steady_clock::now();
};

extern "C" {

inline int mi_device_count(void) { std::lock_guard<std::mutex> lk(stub::world().mu); return stub::world().devices; }

inline const char* mi_status_str(mi_status s)
{
    switch (s) {
        case MI_OK: return "MI_OK"; case MI_ERR_BAD_ARG: return "MI_ERR_BAD_ARG"; case MI_ERR_UNSUPPORTED: return "MI_ERR_UNSUPPORTED";
        case MI_ERR_HIP: return "MI_ERR_HIP"; case MI_ERR_OOM: return "MI_ERR_OOM"; case MI_ERR_NO_DEVICE: return "MI_ERR_NO_DEVICE";
        case MI_ERR_BUSY: return "MI_ERR_BUSY";
    }
    return "?";
}

inline const char* mi_ctx_last_error_msg(const mi_ctx* c) { return c ? c->last_err.c_str() : "no context"; }

inline mi_status mi_ctx_create(int device, mi_ctx** out)
{
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    if (!out) return MI_ERR_BAD_ARG;
    *out = nullptr;
    if (device < 0 || device >= w.devices) return MI_ERR_NO_DEVICE;
    if (w.ctx_fails.count(device)) return MI_ERR_HIP;
    auto* c = new mi_ctx;
    c->device = device;
    ++w.ctx_created[device];
    if (++w.live_ctx[device] > w.peak_ctx[device]) w.peak_ctx[device] = w.live_ctx[device];
    w.ctx_by_thread[std::this_thread::get_id()].insert(device);
    *out = c;
    return MI_OK;
}

inline void mi_ctx_destroy(mi_ctx* c)
{
    if (!c) return;
    auto& w = stub::world();
    {
        std::lock_guard<std::mutex> lk(w.mu);
        --w.live_ctx[c->device];
        if (c->has_pipe) std::fprintf(stderr, "STUB: context of device %d destroyed before its pipe\n", c->device);
    }
    delete c;
}

inline mi_status mi_ctx_set_option(mi_ctx*, const char*, int) { return MI_OK; }

inline mi_status mi_thread_bind_near_device(int device, mi_numa_binding* out)
{
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    w.bound[std::this_thread::get_id()] = device;
    w.bind_log.emplace_back(std::this_thread::get_id(), device);
    if (std::getenv("MI_STUB_TRACE")) std::fprintf(stderr, "STUBTRACE bind device=%d\n", device);
    if (out) {
        out->node = device / 4;                          // two "sockets" of four devices
        out->cpus = 16;
        std::snprintf(out->why, sizeof out->why, "stub: GPU %d -> NUMA node %d, bound to 16 CPUs", device, device / 4);
    }
    return MI_OK;
}

inline mi_status mi_host_register(void* ptr, size_t bytes)
{
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    if (!ptr || !bytes || w.registered.count(ptr)) return MI_ERR_BAD_ARG;
    if (w.devices <= 0) return MI_ERR_NO_DEVICE;
    w.registered[ptr] = bytes;
    return MI_OK;
}

inline mi_status mi_host_unregister(void* ptr)
{
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    return w.registered.erase(ptr) ? MI_OK : MI_ERR_BAD_ARG;
}

inline mi_status mi_pipe_create(mi_ctx* c, const mi_pipe_config* cfg, mi_pipe** out)
{
    if (!c || !cfg || !out) return MI_ERR_BAD_ARG;
    if (c->has_pipe) { c->last_err = "a pipe already exists on this context"; return MI_ERR_BUSY; }
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->depth < 2 || cfg->depth > 16) { c->last_err = "bad pipe config"; return MI_ERR_BAD_ARG; }
    auto* p = new mi_pipe;
    p->ctx = c;
    p->cfg = *cfg;
    p->rng ^= (uint64_t)(c->device + 1) * 0xD1B54A32D192ED03ull;
    c->has_pipe = true;
    auto& w = stub::world();
    std::lock_guard<std::mutex> lk(w.mu);
    if (++w.live_pipes > w.peak_pipes) w.peak_pipes = w.live_pipes;
    *out = p;
    return MI_OK;
}

inline void mi_pipe_destroy(mi_pipe* p)
{
    if (!p) return;
    auto& w = stub::world();
    {
        std::lock_guard<std::mutex> lk(w.mu);
        --w.live_pipes;
        if (!p->pending.empty()) std::fprintf(stderr, "STUB: pipe of device %d destroyed with %zu frames never waited for\n", p->ctx->device, p->pending.size());
    }
    p->ctx->has_pipe = false;
    delete p;
}

inline int mi_pipe_pending(const mi_pipe* p) { return p ? (int)p->pending.size() : 0; }
inline int mi_pipe_depth(const mi_pipe* p) { return p ? p->cfg.depth : 0; }

inline mi_status mi_pipe_submit(mi_pipe* p, const uint8_t* in, uint8_t* out, uint64_t tag)
{
    if (!p || !in || !out) return MI_ERR_BAD_ARG;
    auto& w = stub::world();
    bool fail, slow = false;
    unsigned max_us;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        fail = w.submit_fails.count(p->ctx->device) != 0;
        if (fail) ++w.submit_failures;
        for (int d : w.slow_devices) slow = slow || d == p->ctx->device;
        max_us = w.max_delay_us;
    }
    if (fail) { p->ctx->last_err = "stub: injected submit failure on device " + std::to_string(p->ctx->device); return MI_ERR_HIP; }
    if ((int)p->pending.size() >= p->cfg.depth) { p->ctx->last_err = "pipe full"; return MI_ERR_BUSY; }
    p->rng ^= p->rng << 13; p->rng ^= p->rng >> 7; p->rng ^= p->rng << 17;
    unsigned us = max_us ? (unsigned)(p->rng % max_us) : 0;
    if (slow) us = us * 4 + 2000;
    // one "engine" per pipe: a frame starts when the one before it is done, so completion is in submission order per pipe
    const auto now = std::chrono::steady_clock::now();
    const auto start = p->engine_free > now ? p->engine_free : now;
    p->engine_free = start + std::chrono::microseconds(us);
    p->pending.push_back({in, out, tag, p->engine_free});
    return MI_OK;
}

inline mi_status mi_pipe_wait(mi_pipe* p, uint64_t* tag, uint8_t** out_frame)
{
    if (!p) return MI_ERR_BAD_ARG;
    if (p->pending.empty()) { p->ctx->last_err = "nothing pending"; return MI_ERR_BAD_ARG; }
    mi_pipe::Slot s = p->pending.front();
    p->pending.pop_front();                              // one wait, one frame gone, whatever the status
    std::this_thread::sleep_until(s.ready);
    if (tag) *tag = s.tag;
    if (out_frame) *out_frame = s.out;
    auto& w = stub::world();
    bool fail;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        ++p->waits;
        fail = w.wait_fails.count(p->ctx->device) && w.wait_fail_every > 0 && p->waits % (uint64_t)w.wait_fail_every == 0;
        if (fail) ++w.wait_failures;
        w.completions.push_back({s.tag, p->ctx->device});
        ++w.frames_by_device[p->ctx->device];
    }
    if (fail) { p->ctx->last_err = "stub: injected wait failure on device " + std::to_string(p->ctx->device); return MI_ERR_HIP; }
    stub::stamp(s.out, s.in, (size_t)p->cfg.width * p->cfg.height * 3 / 2, s.tag, p->ctx->device);
    return MI_OK;
}

}  // extern "C"
#endif
