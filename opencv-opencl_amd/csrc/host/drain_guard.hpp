// drain_guard.hpp -- "never return while a DMA on caller memory is in flight".
//
// Every entry point that enqueues a copy from / into memory the CALLER owns (a cv::Mat's pixels, a frame of a GstBufferPool)
// promises that the memory is free to be reused or released when the call returns -- on the error returns as well
// (the reference's accelerator path has no such care: OpenCLequalHist.cpp:367 swallows errors between a write and a read).
// A DrainOnExit sits on the stack of such a function: streams are handed to it BEFORE the first copy that touches caller
// memory is enqueued on them; every exit path that has not called done() -- an early `return st`, a HIPCHK that fired, an
// exception -- synchronises those streams from the destructor.  The success path calls done() once it has itself waited for
// the last copy.
//
// Stand-alone on purpose (no HIP header): the synchronise call is injected, so tests/cxx/test_drain_guard.cpp can exercise
// the exit paths on a machine without a GPU.
#ifndef MI_DRAIN_GUARD_HPP_
#define MI_DRAIN_GUARD_HPP_

namespace mi_host {

template <class SyncFn, int kMax = 4>
class DrainOnExit {
public:
    explicit DrainOnExit(SyncFn sync, unsigned long long* drains = nullptr) : sync_(sync), drains_(drains) {}
    DrainOnExit(const DrainOnExit&) = delete;
    DrainOnExit& operator=(const DrainOnExit&) = delete;
    // a copy on caller memory is about to be enqueued on `stream`
    void watch(void* stream)
    {
        for (int i = 0; i < n_; ++i) if (streams_[i] == stream) return;
        if (n_ < kMax) streams_[n_++] = stream;
        else all_ = true;                            // no slot left (never in this library: at most three streams): drain the device
    }
    // the caller has waited for everything it enqueued: nothing left to drain
    void done() { n_ = 0; all_ = false; }
    bool armed() const { return n_ > 0; }
    ~DrainOnExit()
    {
        if (n_ == 0) return;
        if (drains_) ++*drains_;
        if (all_) { sync_(nullptr); return; }        // sync(nullptr) = "everything on the device"
        for (int i = 0; i < n_; ++i) sync_(streams_[i]);
    }

private:
    SyncFn sync_;
    unsigned long long* drains_;
    void* streams_[kMax] = {};
    int n_ = 0;
    bool all_ = false;
};

}  // namespace mi_host
#endif
