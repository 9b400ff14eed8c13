// pin_registry.hpp -- which caller memory may be handed to the copy engines as it is.
//
// The library DMAs a host plane directly only when it is pinned: registered through mi_host_register (this registry), or pinned by
// the caller, which the HIP runtime is asked about.  Everything else is packed through staging the library owns.  Three things can
// happen to a range at once -- a worker judging it before a submit, another worker judging a different range, and somebody
// unregistering it -- and round 4 made that safe with ONE mutex held across the runtime calls, including hipHostUnregister, which
// waits for the device: a single unregister stalled every submitter of every GPU.  This version keeps the same
// guarantees and holds the lock only over the registry's own vectors:
//
//   judge     lock: registered -> pinned; being unpinned -> NOT pinned; remembered "unknown to the runtime" -> not pinned.  unlock.
//             ask the runtime (no lock held).  If it says pinned: lock again, and accept the answer only if no registration or
//             unregistration started or ended meanwhile (generation unchanged) and the range is not being unpinned; otherwise the
//             plane is staged this once -- slower, never wrong.
//   remove    lock: find; a queued DMA still uses it -> busy; move the range to the "being unpinned" list; generation++.  unlock.
//             unpin through the runtime (no lock held -- this is the call that may wait for the device).
//             lock: drop it from "being unpinned"; if the runtime REFUSED and the pages are still pinned, put it back among the
//             registered ranges (the caller can retry and is still owed an MI_OK before freeing the memory -- ADVICE r4: it used to be
//             forgotten); if the runtime refused because it does not know the pages as pinned ANY MORE (the caller or a teardown
//             unpinned them behind the library's back), the entry is dropped: put back, it could never be removed again, and once the
//             address was freed and reused the registry would answer "pinned" for memory nobody pinned; generation++.
//
// Why a judge can never bless a range whose pages are about to be unpinned: a pipe add()s a frame's ranges to the pending-DMA table
// BEFORE it judges them (pending_ranges.hpp).  A remove() whose first critical section comes after that add() answers busy; one
// whose first critical section came before it has the range on the "being unpinned" list, or has finished and bumped the
// generation twice, by the time the judge looks again.
//
// The reference has no such thing: its accelerator worker maps, enqueues on and unmaps caller buffers with every error swallowed
// (OpenCLequalHist.cpp:307-367).  Stand-alone on purpose (no HIP header; the runtime is two callables): tests/cxx/test_host_helpers.cpp
// runs it on a machine without a GPU, plain and under ThreadSanitizer.
#ifndef MI_PIN_REGISTRY_HPP_
#define MI_PIN_REGISTRY_HPP_

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

#include "pending_ranges.hpp"

namespace mi_host {

// Per-context memory of ranges the runtime was asked about and did NOT know as pinned.  Only the negative verdict is remembered:
// "not pinned" is always safe (the plane is packed through the library's own staging), while a remembered "pinned" could outlive
// the caller's own hipHostUnregister and hand the runtime pageable memory.  Entries expire with every (un)registration and after
// kNegLife look-ups, so memory the caller pins later on is noticed again.
struct PinnedNegCache {
    static constexpr int kSlots = 8;
    static constexpr uint32_t kNegLife = 4096;
    struct Entry { uintptr_t lo = 0; size_t bytes = 0; uint32_t left = 0; } e[kSlots];
    uint64_t generation = 0;
    int next = 0;
    bool hit(uintptr_t lo, size_t bytes, uint64_t g)
    {
        if (g != generation) { for (auto& x : e) x.left = 0; generation = g; return false; }
        for (auto& x : e) if (x.left && x.lo == lo && x.bytes == bytes) { --x.left; return true; }
        return false;
    }
    void remember(uintptr_t lo, size_t bytes) { e[next] = Entry{lo, bytes, kNegLife}; next = (next + 1) % kSlots; }
};

class PinRegistry {
public:
    enum Removal { REMOVED = 0, NOT_REGISTERED = 1, BUSY = 2, RUNTIME_REFUSED = 3, ALREADY_UNPINNED = 4 };
    // what the unpin callable of remove() answers (a plain bool works too: false = REFUSED, true = UNPINNED)
    enum Unpin { REFUSED = 0, UNPINNED = 1, NOT_PINNED_ANY_MORE = 2 };

    // [p, p + bytes) was pinned through the runtime on the library's behalf (mi_host_register, after hipHostRegister succeeded)
    void add(const void* p, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(mu_);
        pinned_.push_back({(uintptr_t)p, (uintptr_t)p + bytes});
        ++generation_;
    }

    // May the copy engines be given [p, p + bytes) as it is?  ask_runtime(p, bytes) -> bool is called WITHOUT the lock.
    template <class AskRuntime>
    bool pinned(const void* p, size_t bytes, PinnedNegCache* neg, AskRuntime&& ask_runtime)
    {
        if (!p || bytes == 0) return false;
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        uint64_t g0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (overlaps(unpinning_, lo, hi)) return false;
            for (const auto& r : pinned_) if (lo >= r.lo && hi <= r.hi) return true;
            g0 = generation_;
            if (neg && neg->hit(lo, bytes, g0)) return false;
        }
        const bool says_pinned = ask_runtime(p, bytes);
        std::lock_guard<std::mutex> lk(mu_);
        if (!says_pinned) {
            if (neg && generation_ == g0) neg->remember(lo, bytes);
            return false;
        }
        // the runtime's "pinned" is only as good as the moment it was given: anything (un)registered since -> stage this once
        return generation_ == g0 && !overlaps(unpinning_, lo, hi);
    }

    // mi_host_unregister.  unpin(ptr) -> Unpin (or bool) is called WITHOUT the lock; it may wait for the device.
    template <class UnpinFn>
    Removal remove(void* ptr, const PendingRanges& pending, UnpinFn&& unpin)
    {
        Range r{};
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = std::find_if(pinned_.begin(), pinned_.end(), [&](const Range& x) { return x.lo == (uintptr_t)ptr; });
            if (it == pinned_.end()) {
                // a second thread is unpinning this very range right now: BUSY -- and once that thread is through, the retry the header
                // asks for finds nothing and gets NOT_REGISTERED, which then means "already removed".  Interior pointers of such a range
                // are NOT_REGISTERED at once, as they are for any registered range.
                for (const Range& x : unpinning_) if (x.lo == (uintptr_t)ptr) return BUSY;
                return NOT_REGISTERED;
            }
            // a pipe still has a transfer queued on this buffer (submitted, not yet retired by mi_pipe_wait): unpinning it now would
            // leave the copy engine with an ordinary heap address.  The caller waits for its frames (or destroys the pipe) and asks again.
            if (pending.overlaps(it->lo, it->hi)) return BUSY;
            r = *it;
            pinned_.erase(it);
            unpinning_.push_back(r);
            ++generation_;
        }
        const int verdict = (int)unpin(ptr);
        std::lock_guard<std::mutex> lk(mu_);
        const auto it = std::find_if(unpinning_.begin(), unpinning_.end(), [&](const Range& x) { return x.lo == r.lo && x.hi == r.hi; });
        if (it != unpinning_.end()) unpinning_.erase(it);    // (always there: a second remover of the same range was told BUSY above)
        if (verdict == REFUSED) pinned_.push_back(r);        // still pinned: keep it, the caller may retry
        ++generation_;
        return verdict == UNPINNED ? REMOVED : verdict == REFUSED ? RUNTIME_REFUSED : ALREADY_UNPINNED;
    }

    size_t size() const { std::lock_guard<std::mutex> lk(mu_); return pinned_.size(); }
    uint64_t generation() const { std::lock_guard<std::mutex> lk(mu_); return generation_; }

private:
    struct Range { uintptr_t lo, hi; };
    static bool overlaps(const std::vector<Range>& v, uintptr_t lo, uintptr_t hi)
    {
        for (const Range& r : v) if (r.lo < hi && lo < r.hi) return true;
        return false;
    }
    mutable std::mutex mu_;
    std::vector<Range> pinned_, unpinning_;
    uint64_t generation_ = 1;
};

}  // namespace mi_host
#endif
