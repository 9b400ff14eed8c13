// mi_pool.hpp -- frame-parallel worker pool over N GPUs with IN-ORDER completion.
//
// Host-side mirror of the reference's scheduling layer:
//   GAsyncQueue + N worker threads popping frames, each running the luma op and pushing the result
//   downstream                                              OpenCVequalHist.cpp:102-196, :397-402
//   --workers 1..8                                          OpenCVequalHist.cpp:274
// Differences, on purpose: worker w is bound to GPU (w mod device_count) and owns one mi_ctx there
// (the shape of the per-worker OpenCL objects, OpenCLequalHist.cpp:142-152); frames are sharded
// frame k -> worker k mod N; and results are delivered in frame order (the reference pushes them in
// completion order, which can reorder frames).  No collective: frames are independent.
#ifndef MI_POOL_HPP_
#define MI_POOL_HPP_

#include <atomic>
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

    FramePool(int workers, int width, int height, Op op, UVMode uv, Sink sink,
              double clip = 2.0, Size tiles = Size(8, 8), size_t max_queue = 16)
        : width_(width), height_(height), op_(op), uv_(uv), clip_(clip), tiles_(tiles), sink_(std::move(sink)), max_queue_(max_queue)
    {
        if (workers < 1) workers = 1;
        if (workers > 64) workers = 64;
        const int ndev = getDeviceCount();
        if (ndev <= 0) MI_CV_ERROR(GpuNotSupported, "no HIP device (this backend has no CPU fallback)");
        queues_.resize(workers);
        for (int w = 0; w < workers; ++w) threads_.emplace_back([this, w, ndev] { run(w, w % ndev); });
        // wait until every worker has created its context and sized its staging/scratch for W x H, so the first real
        // frame does not pay ~100 ms of one-time allocation (it would blow a 16.7 ms frame budget)
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return ready_ == (int)threads_.size(); });
    }
    ~FramePool() { finish(); }

    int workers() const { return (int)threads_.size(); }
    const PoolStats& stats() const { return stats_; }
    size_t queue_depth() const { std::lock_guard<std::mutex> lk(mu_); size_t n = 0; for (auto& q : queues_) n += q.size(); return n; }

    // Blocks while the target worker's queue is full (back-pressure, like the bounded appsink queue).
    uint64_t submit(const unsigned char* in, unsigned char* out)
    {
        std::unique_lock<std::mutex> lk(mu_);
        const uint64_t idx = next_index_++;
        auto& q = queues_[idx % queues_.size()];             // frame k -> worker k mod N
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
        setDevice(device);
        try {                                                   // warm-up: context + allocations for this frame size
            const size_t fb = (size_t)width_ * height_ + (size_t)width_ * height_ / 2;
            std::vector<unsigned char> tmp(fb, 128);
            process(tmp.data(), tmp.data());
        } catch (const std::exception&) {
            // reported per frame later; the pool still comes up so submit()/finish() do not hang
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++ready_;
            cv_done_.notify_all();
        }
        for (;;) {
            FrameJob j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return !queues_[w].empty() || stop_; });
                if (queues_[w].empty()) return;
                j = queues_[w].front();
                queues_[w].pop_front();
                cv_space_.notify_all();
            }
            try {
                process(j.in, j.out);
                j.ok = true;
            } catch (const std::exception& e) {             // per-frame drop-and-count, OpenCVequalHist.cpp:189-193
                j.ok = false; j.error = e.what();
                stats_.processing_errors.fetch_add(1, std::memory_order_relaxed);
            }
            deliver(j);
        }
    }

    void process(const unsigned char* in, unsigned char* out)
    {
        if (op_ == EQUALIZE) equalizeHistNV12(in, out, width_, height_, uv_);
        else if (op_ == CLAHE_OP) claheNV12(in, out, width_, height_, uv_, clip_, tiles_);
        else equalizeHistChannelsNV12(in, out, width_, height_);
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
