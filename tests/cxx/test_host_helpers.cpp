// test_host_helpers.cpp -- CPU-only unit tests of the two stand-alone host helpers of libmi_lumaeq:
//   host/drain_guard.hpp  "never return while a DMA on caller memory is in flight": every exit path of a function that has not
//                         itself waited for its streams synchronises them (stubbed synchronise call, stubbed failures)
//   host/copy_crew.hpp    the calling thread + one helper copying a plane into / out of pinned staging
// No GPU, no HIP: both headers are written against injected / standard facilities so that their exit paths can be checked here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../opencv-opencl_amd/csrc/host/copy_crew.hpp"
#include "../../opencv-opencl_amd/csrc/host/drain_guard.hpp"

static int g_fail = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

// ---- drain guard ---------------------------------------------------------------------------------------
struct SyncLog {
    std::vector<void*>* log;
    void operator()(void* s) const { log->push_back(s); }
};
using Drain = mi_host::DrainOnExit<SyncLog>;

enum Outcome { OK = 0, FAIL_BEFORE_COPY, FAIL_AFTER_H2D, FAIL_AFTER_KERNEL, FAIL_AFTER_D2H, THROW_AFTER_H2D };

// the shape of host_op(): allocate, H2D on caller memory, kernels, D2H on caller memory, synchronise
static int host_form_like(Outcome o, std::vector<void*>* log, unsigned long long* drains, void* stream)
{
    if (o == FAIL_BEFORE_COPY) return 4;                  // e.g. staging allocation failed: nothing queued, nothing to drain
    Drain drain(SyncLog{log}, drains);
    drain.watch(stream);
    /* hipMemcpyAsync(d_in, src, ...) */
    if (o == FAIL_AFTER_H2D) return 3;
    if (o == THROW_AFTER_H2D) throw std::runtime_error("bad_alloc in a helper");
    /* kernels */
    if (o == FAIL_AFTER_KERNEL) return 3;
    /* hipMemcpyAsync(dst, d_out, ...) */
    if (o == FAIL_AFTER_D2H) return 3;
    /* hipStreamSynchronize(stream) succeeded */
    drain.done();
    return 0;
}

static void test_drain_guard()
{
    int s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
    for (Outcome o : {OK, FAIL_BEFORE_COPY}) {
        std::vector<void*> log; unsigned long long drains = 0;
        host_form_like(o, &log, &drains, &s1);
        CHECK(log.empty() && drains == 0);                // success waited itself; the early failure queued nothing
    }
    for (Outcome o : {FAIL_AFTER_H2D, FAIL_AFTER_KERNEL, FAIL_AFTER_D2H}) {
        std::vector<void*> log; unsigned long long drains = 0;
        CHECK(host_form_like(o, &log, &drains, &s1) == 3);
        CHECK(log.size() == 1 && log[0] == &s1 && drains == 1);   // the stream was synchronised before the error reached the caller
    }
    {
        std::vector<void*> log; unsigned long long drains = 0;
        bool thrown = false;
        try { host_form_like(THROW_AFTER_H2D, &log, &drains, &s1); } catch (const std::exception&) { thrown = true; }
        CHECK(thrown && log.size() == 1 && drains == 1);  // unwinding drains as well
    }
    {   // the pipe's three streams, each once, in the order they were handed over; duplicates are not synchronised twice
        std::vector<void*> log;
        {
            Drain d(SyncLog{&log});
            d.watch(&s1); d.watch(&s2); d.watch(&s3); d.watch(&s2);
            CHECK(d.armed());
        }
        CHECK(log.size() == 3 && log[0] == &s1 && log[1] == &s2 && log[2] == &s3);
    }
    {   // more streams than slots: the whole device is drained (sync(nullptr))
        std::vector<void*> log;
        {
            Drain d(SyncLog{&log});
            d.watch(&s1); d.watch(&s2); d.watch(&s3); d.watch(&s4); d.watch(&s5);
        }
        CHECK(log.size() == 1 && log[0] == nullptr);
    }
    {   // done() then a new watch(): armed again (a function with two phases)
        std::vector<void*> log;
        {
            Drain d(SyncLog{&log});
            d.watch(&s1); d.done(); CHECK(!d.armed());
            d.watch(&s2);
        }
        CHECK(log.size() == 1 && log[0] == &s2);
    }
}

// ---- copy crew -----------------------------------------------------------------------------------------
static void fill(std::vector<uint8_t>& v, unsigned seed)
{
    unsigned x = seed * 2654435761u + 12345u;
    for (auto& b : v) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); }
}

static void test_copy_crew()
{
    mi_host::CopyCrew crew;
    // before begin(): plain copy, no thread
    {
        std::vector<uint8_t> a(1 << 20), b(1 << 20, 0);
        fill(a, 1);
        crew.copy_rows(b.data(), 1 << 20, a.data(), 1 << 20, 1 << 20, 1);
        CHECK(a == b && crew.shared_jobs() == 0 && crew.alone_jobs() == 0);
    }
    // contiguous planes and strided views of many sizes, 200 calls back to back; bytes outside the view must stay untouched
    unsigned long long jobs = 0;
    for (int rep = 0; rep < 200; ++rep) {
        const size_t width = 64 + (size_t)(rep * 37) % 4000, rows = 1 + (size_t)(rep * 53) % 700;
        const bool strided = rep % 3 != 0;
        const size_t sstep = strided ? width + 17 : width, dstep = strided ? width + 96 : width;
        std::vector<uint8_t> src(sstep * rows + 64), dst(dstep * rows + 64, 0xAB), ref;
        fill(src, 100 + rep);
        ref = dst;
        for (size_t y = 0; y < rows; ++y) memcpy(ref.data() + y * dstep, src.data() + y * sstep, width);
        crew.begin();
        crew.copy_rows(dst.data(), dstep, src.data(), sstep, width, rows);
        crew.end();
        CHECK(dst == ref);
        if (width * rows >= mi_host::CopyCrew::kMinBytes) ++jobs;
    }
    CHECK(crew.shared_jobs() + crew.alone_jobs() == jobs);   // every large copy was either shared or finished by the caller alone
    // several copies inside one begin()/end() bracket (the chunk loop of host_op), large enough for the helper to be awake
    {
        const size_t n = 8u << 20;
        std::vector<uint8_t> a(n), b(n, 0);
        fill(a, 7);
        const unsigned long long shared0 = crew.shared_jobs();
        crew.begin();
        for (size_t off = 0; off < n; off += 1u << 20) crew.copy_rows(b.data() + off, 1u << 20, a.data() + off, 1u << 20, 1u << 20, 1);
        crew.end();
        CHECK(a == b);
        printf("copy crew: %llu of 8 chunk copies shared with the helper, %llu done by the caller alone in total\n",
               crew.shared_jobs() - shared0, crew.alone_jobs());
    }
    // stop() is idempotent and a stopped crew still copies (alone)
    crew.stop(); crew.stop();
    {
        std::vector<uint8_t> a(1 << 20), b(1 << 20, 0);
        fill(a, 9);
        crew.copy_rows(b.data(), 1 << 20, a.data(), 1 << 20, 1 << 20, 1);
        CHECK(a == b);
    }
}

int main()
{
    test_drain_guard();
    test_copy_crew();
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("host helpers ok\n");
    return 0;
}
