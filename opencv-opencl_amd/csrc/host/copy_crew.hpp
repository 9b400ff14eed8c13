// copy_crew.hpp -- two threads for the one host memcpy the cv::Mat boundary cannot avoid.
//
// The reference hands the op ordinary (pageable) Mats (OpenCVequalHist.cpp:141-145: a Mat over a mapped GstBuffer, a cloned
// Mat next to it).  The library never gives the HIP runtime memory it did not pin itself (DESIGN.md "host memory"), so such a
// plane is packed through pinned staging by the CPU, chunk by chunk, overlapped with the DMA of the chunk before.  One core
// moves ~30 GB/s, the link ~53 GB/s: a single copying thread, not PCIe, bounds a synchronous 4K call (0.59 ms against 0.35 ms
// from pinned Mats).  A CopyCrew is the calling thread plus ONE helper that sleeps between calls:
//   begin()      wake the helper (it then spins for work until end(), so picking up a job costs no wake-up latency)
//   copy_rows()  the caller copies the first half of the rows, the helper the second; if the helper has not claimed its half
//                by the time the caller is done with its own -- still waking up, descheduled -- the caller copies that too
//   end()        the helper goes back to sleep -- after lingering for kLingerUs: a caller that comes back within that time (a
//                streaming worker copying frame after frame, a loop of synchronous calls) finds it still spinning and pays no
//                wake-up latency (~50 us of a ~250 us copy otherwise); an idle context costs nothing
// Nothing depends on the helper making progress: it only ever takes work the caller would otherwise do itself.
// Stand-alone (no HIP): tests/cxx/test_copy_crew.cpp runs it on the CPU.
#ifndef MI_COPY_CREW_HPP_
#define MI_COPY_CREW_HPP_

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>

namespace mi_host {

inline void copy_rows_plain(uint8_t* dst, size_t dstep, const uint8_t* src, size_t sstep, size_t width, size_t rows)
{
    if (dstep == width && sstep == width) { memcpy(dst, src, width * rows); return; }
    for (size_t y = 0; y < rows; ++y) memcpy(dst + y * dstep, src + y * sstep, width);
}

class CopyCrew {
public:
    static constexpr size_t kMinBytes = 256u << 10;      // below this a second thread costs more than it saves
    static constexpr int kLingerUs = 300;                // how long the helper keeps spinning after end() before it sleeps

    CopyCrew() = default;
    CopyCrew(const CopyCrew&) = delete;
    CopyCrew& operator=(const CopyCrew&) = delete;
    ~CopyCrew() { stop(); }

    void begin()
    {
        if (!th_.joinable() && !no_thread_) {
            try { th_ = std::thread([this] { loop(); }); }
            catch (...) { no_thread_ = true; }            // no second thread to be had: the caller copies alone, as before
        }
        { std::lock_guard<std::mutex> lk(mu_); state_.store(1, std::memory_order_release); }
        cv_.notify_one();
    }
    void end() { state_.store(0, std::memory_order_release); }

    void copy_rows(uint8_t* dst, size_t dstep, const uint8_t* src, size_t sstep, size_t width, size_t rows)
    {
        const size_t bytes = width * rows;
        if (bytes < kMinBytes || state_.load(std::memory_order_acquire) != 1) { copy_rows_plain(dst, dstep, src, sstep, width, rows); return; }
        Job j;
        if (dstep == width && sstep == width) {          // one run of bytes: cut it at a page boundary near the middle
            const size_t cut = (bytes / 2) & ~(size_t)4095;
            j = Job{dst + cut, bytes - cut, src + cut, bytes - cut, bytes - cut, 1};
            publish(j);
            memcpy(dst, src, cut);
        } else {
            const size_t r0 = rows / 2;
            j = Job{dst + r0 * dstep, dstep, src + r0 * sstep, sstep, width, rows - r0};
            publish(j);
            copy_rows_plain(dst, dstep, src, sstep, width, r0);
        }
        const uint64_t seq = job_seq_.load(std::memory_order_relaxed);
        uint64_t expect = seq - 1;
        if (claimed_.compare_exchange_strong(expect, seq, std::memory_order_acq_rel)) {
            copy_rows_plain(j.dst, j.dstep, j.src, j.sstep, j.width, j.rows);      // the helper never showed up: do its half as well
            ++alone_;
            return;
        }
        while (done_.load(std::memory_order_acquire) != seq) cpu_relax();          // the helper holds the half: it is copying right now
        ++shared_;
    }

    unsigned long long shared_jobs() const { return shared_; }   // copies both threads took part in
    unsigned long long alone_jobs() const { return alone_; }     // copies the caller ended up doing alone although the crew was up

    void stop()
    {
        if (!th_.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu_); state_.store(2, std::memory_order_release); }
        cv_.notify_one();
        th_.join();
    }

private:
    struct Job { uint8_t* dst; size_t dstep; const uint8_t* src; size_t sstep; size_t width; size_t rows; };

    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

    // jobs are numbered from 1; job n may be claimed (by either thread) iff claimed_ == n - 1
    void publish(const Job& j)
    {
        job_ = j;
        job_seq_.store(job_seq_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
    }

    void loop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return state_.load(std::memory_order_acquire) != 0; });
            if (state_.load(std::memory_order_acquire) == 2) return;
            lk.unlock();
            auto idle_since = std::chrono::steady_clock::now();
            for (;;) {
                const int st = state_.load(std::memory_order_acquire);
                if (st == 2) break;
                if (st == 0) {                                       // between calls: linger, then sleep
                    if (std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(kLingerUs)) break;
                    cpu_relax();
                    continue;
                }
                idle_since = std::chrono::steady_clock::now();
                const uint64_t seq = job_seq_.load(std::memory_order_acquire);
                uint64_t expect = seq - 1;
                if (seq != 0 && claimed_.load(std::memory_order_relaxed) == expect
                    && claimed_.compare_exchange_strong(expect, seq, std::memory_order_acq_rel)) {
                    // the claim succeeded for job `seq`: the caller cannot publish the next job before done_ == seq, so job_ is stable
                    const Job j = job_;
                    copy_rows_plain(j.dst, j.dstep, j.src, j.sstep, j.width, j.rows);
                    done_.store(seq, std::memory_order_release);
                } else {
                    cpu_relax();
                }
            }
            lk.lock();
        }
    }

    std::thread th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<int> state_{0};                          // 0 between calls (helper lingering, then asleep), 1 inside a call (helper spinning for jobs), 2 quit
    Job job_{};
    std::atomic<uint64_t> job_seq_{0}, claimed_{0}, done_{0};
    unsigned long long shared_ = 0, alone_ = 0;
    bool no_thread_ = false;
};

}  // namespace mi_host
#endif
