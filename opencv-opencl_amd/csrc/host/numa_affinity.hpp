// numa_affinity.hpp -- put a GPU's host worker on the CPUs next to that GPU.
//
// The frame-sharded stream runs one host worker per GPU (the reference's worker pool, OpenCVequalHist.cpp:397-402, has no
// placement at all).  On an 8-GPU node each worker moves ~100 MB of frames per millisecond through pinned staging and a UV
// memset / memcpy of its own; if the thread runs -- and its staging buffers are first touched -- on the other socket, every one of
// those bytes crosses the inter-socket link on its way to the GPU's PCIe root complex.  So, BEFORE a worker creates its context
// (and with it the pinned buffers):  GPU -> PCI address (hipDeviceGetPCIBusId) -> /sys/bus/pci/devices/<bdf>/numa_node ->
// /sys/devices/system/node/node<N>/cpulist -> sched_setaffinity(that list, intersected with the CPUs the process may use).
// In-process on purpose: no numactl / taskset wrapper has to re-exec anything.
//
// Stand-alone (no HIP; the sysfs root is a parameter) so that tests/cxx/test_host_helpers.cpp can run it against a fake tree.
#ifndef MI_NUMA_AFFINITY_HPP_
#define MI_NUMA_AFFINITY_HPP_

#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mi_host {

inline bool read_small_file(const std::string& path, std::string* out)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    *out = buf;
    return true;
}

// "0-15,128-143" -> {0..15, 128..143}; malformed input yields what was parsed up to the error
inline std::vector<int> parse_cpulist(const std::string& s)
{
    std::vector<int> cpus;
    const char* p = s.c_str();
    while (*p) {
        while (*p && !isdigit((unsigned char)*p)) { if (*p != ',' && !isspace((unsigned char)*p)) return cpus; ++p; }
        if (!*p) break;
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        long b = a;
        p = e;
        if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1) return cpus; p = e; }
        if (a < 0 || b < a || b - a > (1 << 20)) return cpus;
        for (long c = a; c <= b; ++c) cpus.push_back((int)c);
    }
    return cpus;
}

// hipDeviceGetPCIBusId gives "0000:c1:00.0" (any case); sysfs names are lower case
inline std::string normalize_bdf(std::string bdf)
{
    while (!bdf.empty() && isspace((unsigned char)bdf.back())) bdf.pop_back();
    for (auto& ch : bdf) ch = (char)tolower((unsigned char)ch);
    if (bdf.size() == 7) bdf = "0000:" + bdf;                      // "c1:00.0" -> "0000:c1:00.0"
    return bdf;
}

// NUMA node of a PCI device, -1 when the platform does not say (single-node machines report -1)
inline int numa_node_of_pci(const std::string& bdf, const std::string& sysfs_root = "/sys")
{
    std::string txt;
    if (!read_small_file(sysfs_root + "/bus/pci/devices/" + normalize_bdf(bdf) + "/numa_node", &txt)) return -1;
    char* e = nullptr;
    const long v = strtol(txt.c_str(), &e, 10);
    return e == txt.c_str() ? -1 : (int)v;
}

inline std::vector<int> cpus_of_node(int node, const std::string& sysfs_root = "/sys")
{
    std::string txt;
    if (node < 0 || !read_small_file(sysfs_root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &txt)) return {};
    return parse_cpulist(txt);
}

struct NumaBinding {
    int node = -1;          // NUMA node of the device, -1 unknown
    int cpus = 0;           // CPUs the calling thread was bound to (0: not bound)
    std::string why;        // human-readable outcome for banners
};

// A dynamically sized CPU set (CPU_ALLOC): the fixed cpu_set_t holds 1024 CPUs, and sched_getaffinity fails with EINVAL on a
// host that has more.  Grows until the kernel accepts it.
struct CpuSet {
    cpu_set_t* set = nullptr;
    size_t bytes = 0;
    int ncpus = 0;
    explicit CpuSet(int n) { resize(n); }
    CpuSet(const CpuSet& o) { resize(o.ncpus); if (set && o.set) memcpy(set, o.set, bytes); }
    CpuSet& operator=(const CpuSet&) = delete;
    ~CpuSet() { if (set) CPU_FREE(set); }
    void resize(int n)
    {
        if (set) CPU_FREE(set);
        ncpus = n; bytes = CPU_ALLOC_SIZE(n); set = CPU_ALLOC(n);
        if (set) CPU_ZERO_S(bytes, set);
    }
    bool has(int c) const { return set && c >= 0 && c < ncpus && CPU_ISSET_S(c, bytes, set); }
    void add(int c) { if (set && c >= 0 && c < ncpus) CPU_SET_S(c, bytes, set); }
    // the calling thread's current affinity mask; false when the kernel refuses every size up to 2^20 CPUs
    bool load_current()
    {
        for (int n = ncpus < 1024 ? 1024 : ncpus; n <= (1 << 20); n *= 2) {
            if (n != ncpus) resize(n);
            if (!set) return false;
            if (sched_getaffinity(0, bytes, set) == 0) return true;
        }
        return false;
    }
};

// The mask a thread had BEFORE this library first bound it: every later binding of the same thread starts from it, so a thread that
// was put next to GPU A can be moved next to GPU B on another node (intersecting with the CURRENT mask, as round 3 did, left such a
// thread with "none of its CPUs is available" and pinned to the wrong socket).
inline CpuSet*& thread_original_mask()
{
    static thread_local struct Holder { CpuSet* p = nullptr; ~Holder() { delete p; } } h;
    return h.p;
}

// Binds the CALLING THREAD to the CPUs of the device's node that the thread was allowed to run on before its first binding.  Never
// fails hard: an unknown node, an empty intersection or a refused sched_setaffinity leave the thread where it was and say so in `why`.
inline NumaBinding bind_thread_near_pci(const std::string& bdf, const std::string& sysfs_root = "/sys", bool apply = true)
{
    NumaBinding r;
    r.node = numa_node_of_pci(bdf, sysfs_root);
    if (r.node < 0) { r.why = "device " + normalize_bdf(bdf) + ": no NUMA node reported, thread not bound"; return r; }
    const std::vector<int> node_cpus = cpus_of_node(r.node, sysfs_root);
    CpuSet*& original = thread_original_mask();
    CpuSet current(1024);
    if (!original && !current.load_current()) { r.why = "sched_getaffinity failed, thread not bound"; return r; }
    const CpuSet& base = original ? *original : current;           // where this thread may run at all
    CpuSet want(base.ncpus);
    int n = 0;
    for (int c : node_cpus)
        if (base.has(c)) { want.add(c); ++n; }
    if (n == 0) {
        r.why = "device " + normalize_bdf(bdf) + " is on NUMA node " + std::to_string(r.node) + " but none of its CPUs is available to this process, thread not bound";
        return r;
    }
    if (apply) {
        if (!want.set || sched_setaffinity(0, want.bytes, want.set) != 0) { r.why = "sched_setaffinity refused, thread not bound"; return r; }
        if (!original) original = new CpuSet(current);             // first binding of this thread: remember where it came from
    }                                                              // (a dry run, apply == false, changes and remembers nothing)
    r.cpus = n;
    r.why = "device " + normalize_bdf(bdf) + " -> NUMA node " + std::to_string(r.node) + ", thread bound to " + std::to_string(n) + " of its CPUs";
    return r;
}

// Gives the calling thread back the mask it had before its first binding (a pool thread that is done with its GPU); false when the
// thread was never bound through this header.
inline bool unbind_thread()
{
    CpuSet* original = thread_original_mask();
    return original && original->set && sched_setaffinity(0, original->bytes, original->set) == 0;
}

}  // namespace mi_host
#endif
