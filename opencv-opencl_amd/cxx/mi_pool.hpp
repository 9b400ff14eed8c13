// mi_pool.hpp -- frame-parallel worker pool over N GPUs with IN-ORDER completion.
//
// Host-side mirror of the reference's scheduling layer:
//   GAsyncQueue + N worker threads popping frames, each running the luma op and pushing the result
//   downstream                                              OpenCVequalHist.cpp:102-196, :397-402
//   --workers 1..8                                          OpenCVequalHist.cpp:274
// Differences, on purpose: worker w is bound to GPU (w mod device_count) and owns one mi_ctx there
// (the shape of the per-worker OpenCL objects, OpenCLequalHist.cpp:142-152); frames are sharded
// frame k -> worker k mod N; results are delivered in frame order (the reference pushes them in
// completion order, which can reorder frames); and a worker does not run its frames one blocking
// write / task / read at a time (OpenCLequalHist.cpp:356-365): it drives an mi_pipe, so the upload of
// frame k+2, the kernels of frame k+1 and the download of frame k overlap and ONE worker per GPU keeps
// the PCIe link busy in both directions.  No collective: frames are independent.
// Placement: before it creates its context, each worker binds itself to the CPUs of its GPU's NUMA node
// (mi_thread_bind_near_device), so the pinned staging it allocates and the UV half it writes stay next to
// that GPU's PCIe root complex -- the reference places nothing (OpenCVequalHist.cpp:397-402).  `numa_bind = false`
// (or MI_LUMAEQ_NUMA_BIND=0) leaves the threads where the scheduler puts them.
#ifndef MI_POOL_HPP_
#define MI_POOL_HPP_

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mi_cv.hpp"

namespace micv {

struct PoolStats {                       // the counters of OpenCVequalHist.cpp:20-30 / :200-234
    std::atomic<uint64_t> frames_in{0}, frames_out{0}, processing_errors{0};
    // where the workers' time goes, summed over all workers, in nanoseconds (the reference prints per-stage averages the same way:
    // clahevideo.cpp:54-84 "CLAHE / memory / total")
    std::atomic<uint64_t> ns_submit{0}, ns_wait{0}, ns_deliver{0}, ns_idle{0};
};

struct FrameJob {
    uint64_t index = 0;                  // assigned by submit(), strictly increasing
    const unsigned char* in = nullptr;   // tightly packed NV12, W*H*3/2 bytes, caller-owned until delivered
    unsigned char* out = nullptr;        // caller-owned output frame (may equal `in`)
    bool ok = false;
    std::string error;
};

class FramePool {
public:
    enum Op { EQUALIZE, CLAHE_OP, CHANNELS_EQ };   // CHANNELS_EQ: NV12 -> BGR -> equalizeHist on B, G, R -> NV12 (BASELINE config 5 read literally; ignores uv)
    using Sink = std::function<void(const FrameJob&)>;   // called in frame order, from a pool thread

    // depth: frames a worker keeps in flight on its GPU (2..16; 0 = by frame size: 4 at 4K, 6 at 1080p and below);
    // uv_policy: MI_PIPE_UV_AUTO / _HOST / _DEVICE (mi_lumaeq.h)
    FramePool(int workers, int width, int height, Op op, UVMode uv, Sink sink,
              double clip = 2.0, Size tiles = Size(8, 8), size_t max_queue = 16, int depth = 0, int uv_policy = MI_PIPE_UV_AUTO,
              bool numa_bind = true, int max_workers_per_gpu = 2)
        : width_(width), height_(height), op_(op), uv_(uv), clip_(clip), tiles_(tiles), sink_(std::move(sink)), max_queue_(max_queue),
          depth_(depth <= 0 ? ((size_t)width * height * 3 / 2 >= ((size_t)8 << 20) ? 4 : 6)      // a worker fed by the submitting thread: four 4K
                            : (depth < 2 ? 2 : (depth > 16 ? 16 : depth))),                      // frames, six of 1080p and below (docs/experiments.md R4.9)
          uv_policy_(uv_policy), numa_bind_(numa_bind)
    {
        if (workers < 1) workers = 1;
        if (workers > 64) workers = 64;
        const int ndev = getDeviceCount();
        if (ndev <= 0) MI_CV_ERROR(GpuNotSupported, "no HIP device (this backend has no CPU fallback)");
        // The reference's --workers N buys CPU parallelism (OpenCVequalHist.cpp:397-402).  Here a worker only feeds a GPU, and ONE
        // worker already keeps a GPU's download engine ~95 % busy (5.4 k 4K frames/s); two are no slower; three to eight on one GPU
        // run at 4.9-5.05 k (profiles/r03_c_nv12_stream_crowd.txt).  So at most `max_workers_per_gpu` workers are started per GPU;
        // requested() still reports what was asked for.
        requested_ = workers;
        workers = workers_started(workers, ndev, max_workers_per_gpu);
        queues_.resize(workers);
        placement_.resize(workers);
        for (int w = 0; w < workers; ++w) threads_.emplace_back([this, w, ndev] { run(w, device_of_worker(w, ndev)); });
        // wait until every worker has created its context and sized its staging/scratch for W x H, so the first real
        // frame does not pay ~100 ms of one-time allocation (it would blow a 16.7 ms frame budget)
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return ready_ == (int)threads_.size(); });
    }
    ~FramePool() { finish(); }

    // The sharding rule, in one place (frame k -> GPU k mod N): frame k -> worker k mod workers -> GPU worker mod GPUs.
    // Static, so a caller can place per-frame resources (nv12_stream first-touches ring slots next to the GPU they feed).
    static int workers_started(int requested, int ndev, int max_workers_per_gpu)
    {
        if (requested < 1) requested = 1;
        if (requested > 64) requested = 64;
        return (max_workers_per_gpu >= 1 && requested > ndev * max_workers_per_gpu) ? ndev * max_workers_per_gpu : requested;
    }
    static int device_of_worker(int worker, int ndev) { return worker % ndev; }
    static int device_of_frame(uint64_t frame, int workers, int ndev) { return device_of_worker((int)(frame % (uint64_t)workers), ndev); }

    int workers() const { return (int)threads_.size(); }        // workers actually started (<= max_workers_per_gpu per GPU)
    int requested() const { return requested_; }
    // one line per worker: which GPU, which NUMA node, how many CPUs it was bound to (valid once the constructor has returned)
    std::vector<std::string> placement() const { std::lock_guard<std::mutex> lk(mu_); return placement_; }
    const PoolStats& stats() const { return stats_; }
    size_t queue_depth() const { std::lock_guard<std::mutex> lk(mu_); size_t n = 0; for (auto& q : queues_) n += q.size(); return n; }

    // Blocks while the target worker's queue is full (back-pressure, like the bounded appsink queue).
    uint64_t submit(const unsigned char* in, unsigned char* out)
    {
        std::unique_lock<std::mutex> lk(mu_);
        const uint64_t idx = next_index_++;
        auto& q = queues_[idx % queues_.size()];             // frame k -> worker k mod N (device_of_frame())
        cv_space_.wait(lk, [&] { return q.size() < max_queue_ || stop_; });
        FrameJob j; j.index = idx; j.in = in; j.out = out;
        q.push_back(j);
        stats_.frames_in.fetch_add(1, std::memory_order_relaxed);
        cv_work_.notify_all();
        return idx;
    }

    // Waits until every submitted frame has been delivered, then stops the workers.
    void finish()
    {
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (stop_) { lk.unlock(); join(); return; }
            cv_done_.wait(lk, [&] { return delivered_ == next_index_; });
            stop_ = true;
            cv_work_.notify_all();
            cv_space_.notify_all();
        }
        join();
    }

private:
    void join() { for (auto& t : threads_) if (t.joinable()) t.join(); }

    void run(int w, int device)
    {
        {   // placement first: the context created below allocates its pinned staging from this thread
            mi_numa_binding nb{};
            std::string line = "worker " + std::to_string(w) + " -> GPU " + std::to_string(device) + ": ";
            if (!numa_bind_) line += "NUMA binding off";
            else if (mi_thread_bind_near_device(device, &nb) == MI_OK) line += nb.why;
            else line += std::string("not bound (") + nb.why + ")";
            std::lock_guard<std::mutex> lk(mu_);
            placement_[w] = line;
        }
        setDevice(device);
        mi_ctx* c = nullptr;
        mi_pipe* pipe = nullptr;
        std::string pipe_error;
        try {                                                   // context + pipe: allocations and warm-up for this frame size happen here,
            // not under the first real frame (~100 ms would blow a 16.7 ms budget).  The context is the WORKER'S OWN, not the thread's
            // default one (detail::thread_ctx()): while frames are pending in a context's pipe its other entry points answer
            // MI_ERR_BUSY, and a sink that calls micv::equalizeHist / CLAHE on this thread must keep working.
            const mi_status cst = mi_ctx_create(device, &c);
            if (cst != MI_OK) MI_CV_ERROR(cst == MI_ERR_NO_DEVICE ? GpuNotSupported : GpuApiCallError,
                                          std::string("mi_ctx_create(device=") + std::to_string(device) + ") failed: " + mi_status_str(cst));
            mi_pipe_config cfg{};
            cfg.width = width_; cfg.height = height_;
            cfg.op = op_ == EQUALIZE ? MI_OP_EQUALIZE : (op_ == CLAHE_OP ? MI_OP_CLAHE : MI_OP_CHANNELS);
            cfg.uv_mode = (mi_uv_mode)uv_; cfg.clip_limit = clip_; cfg.tiles_x = tiles_.width; cfg.tiles_y = tiles_.height;
            cfg.depth = depth_; cfg.uv_policy = uv_policy_;
            detail::check(c, mi_pipe_create(c, &cfg, &pipe), "mi_pipe_create");
        } catch (const std::exception& e) {
            pipe_error = e.what();                              // reported per frame; the pool still comes up so submit()/finish() do not hang
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++ready_;
            cv_done_.notify_all();
        }
        std::deque<FrameJob> inflight;                          // submitted to the pipe, oldest first
        using clk = std::chrono::steady_clock;
        auto since = [](clk::time_point t) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t).count(); };
        for (;;) {
            FrameJob j;
            bool have = false;
            {
                const auto t_idle = clk::now();
                std::unique_lock<std::mutex> lk(mu_);
                if (inflight.empty()) cv_work_.wait(lk, [&] { return !queues_[w].empty() || stop_; });
                if (!queues_[w].empty() && (int)inflight.size() < depth_) {
                    j = queues_[w].front();
                    queues_[w].pop_front();
                    have = true;
                    cv_space_.notify_all();
                } else if (inflight.empty()) {
                    break;                                      // stopped and drained
                }
                stats_.ns_idle.fetch_add(since(t_idle), std::memory_order_relaxed);
            }
            if (have) {                                         // keep the pipe full before waiting for anything
                const auto t_sub = clk::now();
                mi_status st = pipe ? mi_pipe_submit(pipe, j.in, j.out, j.index) : MI_ERR_HIP;
                stats_.ns_submit.fetch_add(since(t_sub), std::memory_order_relaxed);
                if (st == MI_OK) { inflight.push_back(j); continue; }
                j.ok = false;                                   // per-frame drop-and-count, OpenCVequalHist.cpp:189-193
                j.error = pipe ? std::string("mi_pipe_submit: ") + mi_status_str(st) + " (" + mi_ctx_last_error_msg(c) + ")" : pipe_error;
                stats_.processing_errors.fetch_add(1, std::memory_order_relaxed);
                deliver(j);
                continue;
            }
            FrameJob d = inflight.front();                      // pipe full, or nothing new to submit: complete the oldest frame
            inflight.pop_front();
            uint64_t tag = 0;
            const auto t_wait = clk::now();
            const mi_status st = mi_pipe_wait(pipe, &tag, nullptr);
            stats_.ns_wait.fetch_add(since(t_wait), std::memory_order_relaxed);
            d.ok = st == MI_OK && tag == d.index;
            if (!d.ok) {
                d.error = std::string("mi_pipe_wait: ") + mi_status_str(st) + " (" + mi_ctx_last_error_msg(c) + ")";
                stats_.processing_errors.fetch_add(1, std::memory_order_relaxed);
            }
            const auto t_del = clk::now();
            deliver(d);
            stats_.ns_deliver.fetch_add(since(t_del), std::memory_order_relaxed);
        }
        if (pipe) mi_pipe_destroy(pipe);
        if (c) mi_ctx_destroy(c);
    }

    // re-sequencer: hold completed frames until all earlier ones have been delivered
    void deliver(const FrameJob& j)
    {
        std::unique_lock<std::mutex> lk(mu_);
        done_.emplace(j.index, j);
        while (!delivering_) {
            auto it = done_.find(delivered_);
            if (it == done_.end()) break;
            FrameJob next = it->second;
            done_.erase(it);
            delivering_ = true;
            lk.unlock();
            if (sink_) sink_(next);
            lk.lock();
            delivering_ = false;
            ++delivered_;
            stats_.frames_out.fetch_add(1, std::memory_order_relaxed);
        }
        cv_done_.notify_all();
    }

    int width_, height_;
    Op op_;
    UVMode uv_;
    double clip_;
    Size tiles_;
    Sink sink_;
    size_t max_queue_;
    int depth_, uv_policy_;
    bool numa_bind_;
    int requested_ = 0;
    std::vector<std::string> placement_;
    mutable std::mutex mu_;
    std::condition_variable cv_work_, cv_space_, cv_done_;
    std::vector<std::deque<FrameJob>> queues_;
    std::map<uint64_t, FrameJob> done_;
    std::vector<std::thread> threads_;
    uint64_t next_index_ = 0, delivered_ = 0;
    bool stop_ = false, delivering_ = false;
    int ready_ = 0;
    PoolStats stats_;
};

}  // namespace micv
#endif
