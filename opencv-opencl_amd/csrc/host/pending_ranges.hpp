// pending_ranges.hpp -- host ranges that a queued, not yet awaited DMA of this library still reads or writes.
//
// A pipe DMAs frames the caller pinned (mi_host_register) as they are, asynchronously: between mi_pipe_submit and the
// mi_pipe_wait that retires the frame the copy engine owns the frame's pages.  If the caller unregisters the buffer in
// that window the runtime unpins pages with a transfer in flight -- a GPU access to an ordinary heap address, i.e. the
// "Memory access fault by GPU" that ends the process (the reference's accelerator path has the same window and no guard:
// OpenCLequalHist.cpp:356-367 enqueues on caller buffers and swallows every error).  mi_host_unregister therefore asks this
// table first and answers MI_ERR_BUSY while any pending transfer overlaps the range.
//
// Order that closes the race with a concurrent mi_host_unregister: a submit path add()s its ranges BEFORE it decides
// whether they are pinned, and drops them again if the frame ends up staged.  The unregister path (pin_registry.hpp, which holds
// the whole protocol) looks at this table in its FIRST critical section, moves the range to its "being unpinned" list there, and
// calls hipHostUnregister WITHOUT the registry's lock; a judge never calls a range on that list pinned and distrusts a runtime
// answer that straddles an (un)registration (generation re-check).  Either the unregister sees the pending entry (BUSY), or its
// first critical section came first and the submit path finds the range "being unpinned" / gone and stages the frame.
//
// Stand-alone on purpose (no HIP header): tests/cxx/test_host_helpers.cpp exercises it on a machine without a GPU.
#ifndef MI_PENDING_RANGES_HPP_
#define MI_PENDING_RANGES_HPP_

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

namespace mi_host {

class PendingRanges {
public:
    // a transfer on [p, p + bytes) is about to be queued for (owner, tag): owner = the pipe, tag = its slot
    void add(const void* owner, uint64_t tag, const void* p, size_t bytes)
    {
        if (!p || bytes == 0) return;
        std::lock_guard<std::mutex> lk(mu_);
        v_.push_back(Entry{owner, tag, (uintptr_t)p, (uintptr_t)p + bytes});
    }
    // the transfers of (owner, tag) have been waited for (or were never queued)
    void retire(const void* owner, uint64_t tag)
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t i = v_.size(); i-- > 0;)
            if (v_[i].owner == owner && v_[i].tag == tag) { v_[i] = v_.back(); v_.pop_back(); }
    }
    // the owner has drained its streams (pipe destruction): nothing of its is in flight any more
    void retire_all(const void* owner)
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t i = v_.size(); i-- > 0;)
            if (v_[i].owner == owner) { v_[i] = v_.back(); v_.pop_back(); }
    }
    bool overlaps(uintptr_t lo, uintptr_t hi) const
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (const Entry& e : v_) if (e.lo < hi && lo < e.hi) return true;
        return false;
    }
    size_t size() const { std::lock_guard<std::mutex> lk(mu_); return v_.size(); }

private:
    struct Entry { const void* owner; uint64_t tag; uintptr_t lo, hi; };
    mutable std::mutex mu_;
    std::vector<Entry> v_;
};

}  // namespace mi_host
#endif
