// test_host_helpers.cpp -- CPU-only unit tests of the two stand-alone host helpers of libmi_lumaeq:
//   host/drain_guard.hpp  "never return while a DMA on caller memory is in flight": every exit path of a function that has not
//                         itself waited for its streams synchronises them (stubbed synchronise call, stubbed failures)
//   host/copy_crew.hpp    the calling thread + one helper copying a plane into / out of pinned staging
//   host/numa_affinity.hpp  GPU PCI address -> NUMA node -> CPU list -> thread affinity, against a FAKE sysfs tree
//   host/pending_ranges.hpp  caller memory a pipe's queued DMA still owns: what mi_host_unregister consults before it unpins
//   host/pin_registry.hpp  which caller memory may be DMA'd as it is; an unregister that waits for the device must not stall judges
// No GPU, no HIP: both headers are written against injected / standard facilities so that their exit paths can be checked here.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../opencv-opencl_amd/csrc/host/copy_crew.hpp"
#include "../../opencv-opencl_amd/csrc/host/drain_guard.hpp"
#include "../../opencv-opencl_amd/csrc/host/numa_affinity.hpp"
#include "../../opencv-opencl_amd/csrc/host/pending_ranges.hpp"
#include "../../opencv-opencl_amd/csrc/host/pin_registry.hpp"
#include "../../opencv-opencl_amd/csrc/host/wide_hint.hpp"

#include <thread>

#include <sys/stat.h>
#include <unistd.h>

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

// ---- NUMA placement ------------------------------------------------------------------------------------
static void mkdirs(const std::string& path)
{
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}
static void put(const std::string& path, const std::string& text)
{
    mkdirs(path.substr(0, path.rfind('/')));
    FILE* f = fopen(path.c_str(), "w");
    if (f) { fputs(text.c_str(), f); fclose(f); }
}

static void test_numa_affinity()
{
    using namespace mi_host;
    CHECK((parse_cpulist("0-3,8,10-11\n") == std::vector<int>{0, 1, 2, 3, 8, 10, 11}));
    CHECK((parse_cpulist("5") == std::vector<int>{5}));
    CHECK(parse_cpulist("").empty() && parse_cpulist("\n").empty());
    CHECK((parse_cpulist("0-1,x") == std::vector<int>{0, 1}));         // malformed tail: what was parsed before it
    CHECK(parse_cpulist("7-3").empty());
    CHECK(normalize_bdf("0000:C1:00.0\n") == "0000:c1:00.0" && normalize_bdf("c1:00.0") == "0000:c1:00.0");

    // a fake two-socket machine: the CPUs this process may use are split over "node 0" and "node 1"
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    CHECK(sched_getaffinity(0, sizeof allowed, &allowed) == 0);
    std::vector<int> mine;
    for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) mine.push_back(c);
    CHECK(!mine.empty());
    char tmpl[] = "/tmp/mi_fake_sysfs_XXXXXX";
    const char* root_c = mkdtemp(tmpl);
    CHECK(root_c != nullptr);
    if (!root_c) return;
    const std::string root = root_c;
    auto list = [](const std::vector<int>& v) { std::string s; for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string(v[i]); return s + "\n"; };
    const std::vector<int> n0(mine.begin(), mine.begin() + (mine.size() + 1) / 2), n1(mine.begin() + (mine.size() + 1) / 2, mine.end());
    put(root + "/devices/system/node/node0/cpulist", list(n0));
    put(root + "/devices/system/node/node1/cpulist", n1.empty() ? "\n" : list(n1));
    put(root + "/devices/system/node/node2/cpulist", "100000-100003\n");          // a node whose CPUs this process cannot use
    put(root + "/bus/pci/devices/0000:c1:00.0/numa_node", "0\n");
    put(root + "/bus/pci/devices/0000:e3:00.0/numa_node", "1\n");
    put(root + "/bus/pci/devices/0000:05:00.0/numa_node", "-1\n");                // single-node platforms say -1
    put(root + "/bus/pci/devices/0000:07:00.0/numa_node", "2\n");
    CHECK(numa_node_of_pci("0000:C1:00.0", root) == 0 && numa_node_of_pci("e3:00.0", root) == 1);
    CHECK(numa_node_of_pci("0000:05:00.0", root) == -1 && numa_node_of_pci("0000:99:00.0", root) == -1);   // unknown device: no file
    CHECK(cpus_of_node(0, root) == n0 && cpus_of_node(5, root).empty());

    NumaBinding b = bind_thread_near_pci("0000:c1:00.0", root);
    CHECK(b.node == 0 && b.cpus == (int)n0.size());
    cpu_set_t now;
    CPU_ZERO(&now);
    CHECK(sched_getaffinity(0, sizeof now, &now) == 0);
    for (int c : mine) CHECK((CPU_ISSET(c, &now) != 0) == (std::find(n0.begin(), n0.end(), c) != n0.end()));
    // a thread started AFTER the binding inherits it (the library's helper thread is created from the bound worker)
    int inherited = -1;
    std::thread([&] { cpu_set_t t; CPU_ZERO(&t); sched_getaffinity(0, sizeof t, &t); inherited = CPU_COUNT(&t); }).join();
    CHECK(inherited == (int)n0.size());
    CHECK(sched_setaffinity(0, sizeof allowed, &allowed) == 0);                     // back to the full set for the cases below
    b = bind_thread_near_pci("0000:05:00.0", root);
    CHECK(b.node == -1 && b.cpus == 0 && b.why.find("not bound") != std::string::npos);
    b = bind_thread_near_pci("0000:07:00.0", root);
    CHECK(b.node == 2 && b.cpus == 0 && b.why.find("none of its CPUs") != std::string::npos);
    b = bind_thread_near_pci("0000:c1:00.0", root, /*apply=*/false);                // "disabled": reports, does not bind
    CHECK(b.node == 0 && b.cpus == (int)n0.size());
    CPU_ZERO(&now);
    CHECK(sched_getaffinity(0, sizeof now, &now) == 0 && CPU_COUNT(&now) == (int)mine.size());
    if (!n1.empty()) {
        b = bind_thread_near_pci("0000:e3:00.0", root);
        CHECK(b.node == 1 && b.cpus == (int)n1.size());
        // ... and straight on to a GPU on the OTHER node, without anybody widening the mask in between: the thread's mask from before
        // its first binding is the base, not the one it has now (round 3: "none of its CPUs is available", stuck on the wrong socket)
        b = bind_thread_near_pci("0000:c1:00.0", root);
        CHECK(b.node == 0 && b.cpus == (int)n0.size());
        CPU_ZERO(&now);
        CHECK(sched_getaffinity(0, sizeof now, &now) == 0 && CPU_COUNT(&now) == (int)n0.size());
        for (int c : n0) CHECK(CPU_ISSET(c, &now) != 0);
        CHECK(unbind_thread());                                                      // back where the thread came from
        CPU_ZERO(&now);
        CHECK(sched_getaffinity(0, sizeof now, &now) == 0 && CPU_COUNT(&now) == (int)mine.size());
        // another thread has its own "original" mask: bound, it narrows; unbind_thread() on a thread never bound says so
        bool fresh_unbind = true; int fresh_cpus = -1;
        std::thread([&] { fresh_unbind = unbind_thread(); fresh_cpus = bind_thread_near_pci("0000:e3:00.0", root).cpus; }).join();
        CHECK(!fresh_unbind && fresh_cpus == (int)n1.size());
    }
    {   // CpuSet: sized beyond the 1024 CPUs a cpu_set_t holds
        CpuSet big(5000);
        big.add(4999); big.add(7); big.add(5000); big.add(-1);
        CHECK(big.has(4999) && big.has(7) && !big.has(5000) && !big.has(8) && big.bytes >= 5000 / 8);
        CpuSet copy(big);
        CHECK(copy.has(4999) && copy.ncpus == 5000);
        CpuSet cur(8);
        CHECK(cur.load_current() && cur.ncpus >= 1024);
        for (int c : mine) CHECK(cur.has(c));
    }
    const std::string rm = "rm -rf " + root;
    (void)!system(rm.c_str());
}

// ---- pending DMA ranges ------------------------------------------------------------------------------------
// The sequence of mi_pipe_submit / mi_pipe_wait / mi_host_unregister / pipe destruction, with the HIP calls left out: unregister
// must see BUSY exactly while a frame that is DMA'd from / into the caller's own (pinned) buffer is between submit and wait.
struct FakePipe { int dummy; };

static bool submit_like(mi_host::PendingRanges& t, FakePipe* p, uint64_t slot, const uint8_t* in, uint8_t* out, size_t bytes,
                        bool in_pinned, bool out_pinned, bool fail_after_enqueue, std::vector<void*>* synced)
{
    t.add(p, slot, in, bytes);                                 // entered BEFORE the ranges are judged
    t.add(p, slot, out, bytes);
    struct G { mi_host::PendingRanges& t; FakePipe* p; uint64_t id; bool keep = false; ~G() { if (!keep) t.retire(p, id); } } g{t, p, slot};
    Drain drain(SyncLog{synced}, nullptr);
    drain.watch(p);
    if (fail_after_enqueue) return false;                      // the drain guard waits first (declared later), then the ranges leave
    drain.done();
    g.keep = in_pinned || out_pinned;
    return true;
}

static void test_pending_ranges()
{
    mi_host::PendingRanges t;
    FakePipe a{}, b{};
    std::vector<uint8_t> f0(4096), f1(4096), o0(4096), o1(4096);
    auto busy = [&](const std::vector<uint8_t>& v) { return t.overlaps((uintptr_t)v.data(), (uintptr_t)v.data() + v.size()); };
    std::vector<void*> synced;
    CHECK(!busy(f0) && t.size() == 0);
    CHECK(submit_like(t, &a, 0, f0.data(), o0.data(), 4096, true, true, false, &synced));        // pinned frame in flight
    CHECK(busy(f0) && busy(o0) && !busy(f1) && t.size() == 2);
    CHECK(t.overlaps((uintptr_t)f0.data() + 4095, (uintptr_t)f0.data() + 4097));                  // partial overlap counts
    CHECK(!t.overlaps((uintptr_t)f0.data() + 4096, (uintptr_t)f0.data() + 4097));                 // half-open: the byte after the end does not
    CHECK(submit_like(t, &a, 1, f1.data(), o1.data(), 4096, false, false, false, &synced));      // staged frame: the DMA runs on the slot's staging
    CHECK(!busy(f1) && !busy(o1) && t.size() == 2);
    CHECK(submit_like(t, &b, 0, f1.data(), o1.data(), 4096, true, false, false, &synced));       // another pipe, same slot number, input pinned only
    CHECK(busy(f1) && busy(o1));                                                                  // (both ranges stay: one entry per frame side)
    t.retire(&a, 0);                                                                              // mi_pipe_wait of pipe a's frame
    CHECK(!busy(f0) && !busy(o0) && busy(f1));                                                    // ... does not touch pipe b's slot 0
    CHECK(synced.empty());
    CHECK(!submit_like(t, &a, 2, f0.data(), o0.data(), 4096, true, true, true, &synced));        // a failed submit: drained, nothing left pending
    CHECK(synced.size() == 1 && !busy(f0) && !busy(o0));
    t.retire_all(&b);                                                                             // pipe b destroyed with its frame never waited for
    CHECK(!busy(f1) && t.size() == 0);
    t.add(&a, 0, nullptr, 100); t.add(&a, 0, f0.data(), 0);                                       // nothing to guard
    CHECK(t.size() == 0);
    // two threads submitting and retiring while a third keeps asking: no entry is lost or left behind
    std::thread th[2];
    for (int k = 0; k < 2; ++k)
        th[k] = std::thread([&, k] {
            FakePipe* p = k ? &b : &a;
            for (int i = 0; i < 20000; ++i) { t.add(p, (uint64_t)(i & 3), f0.data() + 64 * k, 64); t.retire(p, (uint64_t)(i & 3)); }
        });
    size_t seen = 0;
    for (int i = 0; i < 20000; ++i) seen += busy(f0) ? 1 : 0;
    for (auto& x : th) x.join();
    CHECK(t.size() == 0 && !busy(f0));
    (void)seen;
}

// ---- pin registry ---------------------------------------------------------------------------------------------------
// The runtime is two callables here: `ask` (what hipPointerGetAttributes + hipMemGetAddressRange would say) and `unpin`
// (hipHostUnregister, which may wait for the device).
static void test_pin_registry()
{
    using mi_host::PinRegistry;
    PinRegistry reg;
    mi_host::PendingRanges pending;
    mi_host::PinnedNegCache neg;
    std::vector<unsigned char> a(4096), b(4096), c(4096);
    std::atomic<int> asked{0};
    std::atomic<bool> runtime_knows_c{false};
    auto ask = [&](const void* p, size_t) { ++asked; return runtime_knows_c.load() && p >= c.data() && p < c.data() + c.size(); };
    auto unpin_ok = [](void*) { return true; };

    // registered ranges are pinned without asking the runtime; sub-ranges too; a range that sticks out is not
    reg.add(a.data(), a.size());
    CHECK(reg.pinned(a.data(), a.size(), &neg, ask) && reg.pinned(a.data() + 100, 1000, &neg, ask) && asked == 0);
    CHECK(!reg.pinned(a.data() + 100, a.size(), &neg, ask) && asked == 1);
    CHECK(!reg.pinned(nullptr, 10, &neg, ask) && !reg.pinned(a.data(), 0, &neg, ask));
    // unknown memory: the runtime is asked once, the negative verdict is remembered (per context) until something is (un)registered
    CHECK(!reg.pinned(b.data(), b.size(), &neg, ask) && asked == 2);
    CHECK(!reg.pinned(b.data(), b.size(), &neg, ask) && asked == 2);
    reg.add(b.data(), 16);                                   // generation moves: remembered verdicts are dropped
    CHECK(!reg.pinned(b.data(), b.size(), &neg, ask) && asked == 3);
    // memory the caller pinned itself: the runtime's word counts, and is never cached
    runtime_knows_c = true;
    CHECK(reg.pinned(c.data(), c.size(), &neg, ask) && reg.pinned(c.data(), c.size(), &neg, ask) && asked == 5);
    // removal: unknown pointer, interior pointer, pending DMA, success
    CHECK(reg.remove(c.data(), pending, unpin_ok) == PinRegistry::NOT_REGISTERED);
    CHECK(reg.remove(a.data() + 1, pending, unpin_ok) == PinRegistry::NOT_REGISTERED);
    int owner = 0;
    pending.add(&owner, 7, a.data() + 512, 64);
    CHECK(reg.remove(a.data(), pending, unpin_ok) == PinRegistry::BUSY && reg.pinned(a.data(), a.size(), &neg, ask));
    pending.retire(&owner, 7);
    // the runtime refuses: the range STAYS registered and can be removed later (it used to be forgotten: ADVICE r4)
    CHECK(reg.remove(a.data(), pending, [](void*) { return false; }) == PinRegistry::RUNTIME_REFUSED);
    CHECK(reg.pinned(a.data(), a.size(), &neg, ask) && reg.size() == 2);
    CHECK(reg.remove(a.data(), pending, unpin_ok) == PinRegistry::REMOVED && reg.size() == 1);
    runtime_knows_c = false;
    CHECK(!reg.pinned(a.data(), a.size(), &neg, ask));
    CHECK(reg.remove(a.data(), pending, unpin_ok) == PinRegistry::NOT_REGISTERED);
    // "refused forever" (ADVICE r5): the runtime refuses because it does not know the pages as pinned any more (the caller unpinned
    // them behind the library's back).  Put back, the entry could never be removed, and after the memory was freed and the address
    // reused the registry would answer "pinned" without asking.  It is dropped instead; a refusal with the pages STILL pinned keeps it.
    reg.add(a.data(), a.size());
    for (int k = 0; k < 3; ++k)
        CHECK(reg.remove(a.data(), pending, [](void*) { return PinRegistry::REFUSED; }) == PinRegistry::RUNTIME_REFUSED && reg.size() == 2);
    CHECK(reg.pinned(a.data(), a.size(), &neg, ask));
    CHECK(reg.remove(a.data(), pending, [](void*) { return PinRegistry::NOT_PINNED_ANY_MORE; }) == PinRegistry::ALREADY_UNPINNED);
    CHECK(reg.size() == 1);
    {
        const int before = asked;
        CHECK(!reg.pinned(a.data(), a.size(), &neg, ask) && asked == before + 1);      // the runtime is asked again: nothing is assumed
    }
    CHECK(reg.remove(a.data(), pending, unpin_ok) == PinRegistry::NOT_REGISTERED);

    // an unregister that waits for the device (unpin really sleeps 400 ms) must not stall anybody who judges a range meanwhile, the
    // range being unpinned must not be judged pinned in that window -- not even if the runtime still says so -- a second remover of
    // the same pointer is told BUSY, and one of an interior pointer NOT_REGISTERED.  If the registry held its lock across the unpin
    // call again, the first verdict below would return only after those 400 ms: a clean failure of the time check, not a deadlock.
    reg.add(a.data(), a.size());
    std::atomic<bool> in_unpin{false};
    std::thread remover([&] {
        CHECK(reg.remove(a.data(), pending, [&](void*) {
            in_unpin = true;
            std::this_thread::sleep_for(std::chrono::milliseconds(400));
            return true;
        }) == PinRegistry::REMOVED);
    });
    while (!in_unpin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    const auto t0 = std::chrono::steady_clock::now();
    mi_host::PinnedNegCache neg2;
    bool any_pinned = false;
    for (int k = 0; k < 1000; ++k) {
        any_pinned = any_pinned || reg.pinned(a.data(), a.size(), &neg2, [](const void*, size_t) { return true; });    // "still pinned" says the runtime
        any_pinned = any_pinned || reg.pinned(a.data() + 64, 128, nullptr, [](const void*, size_t) { return true; });
    }
    CHECK(reg.pinned(b.data(), 16, &neg2, ask));             // other registered ranges are judged as ever
    CHECK(reg.remove(a.data(), pending, unpin_ok) == PinRegistry::BUSY);
    CHECK(reg.remove(a.data() + 8, pending, unpin_ok) == PinRegistry::NOT_REGISTERED);     // interior pointer: never BUSY
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(!any_pinned);
    CHECK(ms < 200.0);                                       // 2000 verdicts + 2 removals well inside the 400 ms the unpin call sleeps
    remover.join();
    CHECK(!reg.pinned(a.data(), a.size(), &neg2, ask));
    CHECK(reg.remove(a.data(), pending, unpin_ok) == PinRegistry::NOT_REGISTERED);         // the BUSY caller's retry: "already removed"

    // a runtime answer that straddles an (un)registration is not trusted: the plane is staged this once
    {
        const uint64_t g = reg.generation();
        const bool verdict = reg.pinned(c.data(), c.size(), nullptr, [&](const void*, size_t) { reg.add(c.data() + 8, 8); return true; });
        CHECK(!verdict && reg.generation() == g + 1);
        CHECK(reg.pinned(c.data(), c.size(), nullptr, [](const void*, size_t) { return true; }));           // quiet again: trusted
        CHECK(reg.remove(c.data() + 8, pending, unpin_ok) == PinRegistry::REMOVED);
    }

    // hammer: judges, registrations and removals of disjoint and shared ranges from several threads (meant for ThreadSanitizer)
    {
        std::vector<std::vector<unsigned char>> bufs(8, std::vector<unsigned char>(1024));
        std::atomic<bool> stop{false};
        std::atomic<long> verdicts{0};
        std::vector<std::thread> ts;
        for (int t = 0; t < 3; ++t)
            ts.emplace_back([&, t] {
                mi_host::PinnedNegCache n;
                while (!stop) {
                    for (auto& v : bufs) verdicts += reg.pinned(v.data(), v.size(), &n, [](const void*, size_t) { return false; });
                    (void)t;
                }
            });
        for (int round = 0; round < 200; ++round)
            for (auto& v : bufs) {
                reg.add(v.data(), v.size());
                CHECK(reg.remove(v.data(), pending, [](void*) { std::this_thread::yield(); return true; }) == PinRegistry::REMOVED);
            }
        stop = true;
        for (auto& t : ts) t.join();
        CHECK(reg.size() == 1);                              // b's 16 bytes from above
        CHECK(reg.remove(b.data(), pending, unpin_ok) == PinRegistry::REMOVED && reg.size() == 0);
    }
}

// ---- the host half of the 16-bit CLAHE's wide-content hint (host/wide_hint.hpp)
static void test_wide_hint()
{
    using mi_host::mid_kernel_wanted;
    CHECK(mid_kernel_wanted(2, 0, 0, 8) && mid_kernel_wanted(2, 5000, 0, 8));            // always
    CHECK(!mid_kernel_wanted(0, 1000, 1000, 8));                                         // never
    CHECK(!mid_kernel_wanted(1, 1000, 0, 8));                                            // a fresh context: "seen" = 0 lies 1000 calls back
    CHECK(mid_kernel_wanted(1, 1010, 1010, 8) && mid_kernel_wanted(1, 1018, 1010, 8));   // seen in this call ... eight executed calls ago
    CHECK(!mid_kernel_wanted(1, 1019, 1010, 8));                                         // nine: off again
    CHECK(mid_kernel_wanted(1, 1010, 1011, 8) && mid_kernel_wanted(1, 1010, 1030, 8));   // "seen" AHEAD of "executed" (the call in flight, a caller
                                                                                         // enqueueing ahead): recent, not 4 billion calls ago
    CHECK(mid_kernel_wanted(1, 3, 0xfffffffeu, 8) && !mid_kernel_wanted(1, 20, 0xfffffffeu, 8));      // across the wrap of the sequence numbers
}

int main()
{
    test_wide_hint();
    test_pin_registry();
    test_drain_guard();
    test_pending_ranges();
    test_copy_crew();
    test_numa_affinity();
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("host helpers ok\n");
    return 0;
}
