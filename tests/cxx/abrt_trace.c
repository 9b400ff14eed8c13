/* abrt_trace.c -- diagnostic for the GPU test session (loaded by tests/conftest.py, or LD_PRELOADed): on SIGABRT / SIGSEGV /
 * SIGBUS it writes the NATIVE call chain of the faulting thread to stderr AND to a file (glibc backtrace; libgcc_s is loaded at
 * install time, not inside the handler), then hands the signal to whoever was installed before it (Python's faulthandler prints
 * the Python stacks and lets the default action run).  A silent abort() from a runtime thread then shows which library called it.
 * The file matters: under pytest's default capture fd 2 is a temporary file that dies with the process, which is how the abort
 * of round 2 (gpurun_out/t_full.txt of that session: only faulthandler's lines, written to a saved copy of stderr) stayed silent.
 * File: $MI_ABRT_TRACE_FILE, else gpurun_out/abrt_trace.log under the current directory if that directory exists or can be made,
 * else /tmp/mi_abrt_trace.log.  It is created empty at install and only ever appended to by the handler. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

static struct sigaction g_prev[65];
static int g_fd = -1;

/* The locked-memory state of the process at the moment of death: the one abort site on record was a runtime path that pins the
 * caller's pages (hsa_amd_memory_lock_to_pool); a lock that failed against RLIMIT_MEMLOCK or the locked-page accounting would show
 * here.  Async-signal-safe: open / read / write only, fixed buffers. */
static void put_num(int fd, const char* label, unsigned long long v, int unlimited)
{
    char b[96]; int n = 0;
    while (label[n] && n < 60) { b[n] = label[n]; ++n; }
    if (unlimited) { memcpy(b + n, "unlimited", 9); n += 9; }
    else {
        char d[24]; int k = 0;
        do { d[k++] = (char)('0' + v % 10); v /= 10; } while (v && k < 24);
        while (k) b[n++] = d[--k];
    }
    b[n++] = '\n';
    (void)!write(fd, b, (size_t)n);
}

static void dump_memlock(int fd)
{
    static const char hdr[] = "[abrt_trace] locked / pinned memory of the process (RLIMIT_MEMLOCK soft, hard in bytes; /proc/self/status):\n";
    (void)!write(fd, hdr, sizeof hdr - 1);
    struct rlimit rl;
    if (getrlimit(RLIMIT_MEMLOCK, &rl) == 0) {
        put_num(fd, "  RLIMIT_MEMLOCK soft: ", (unsigned long long)rl.rlim_cur, rl.rlim_cur == RLIM_INFINITY);
        put_num(fd, "  RLIMIT_MEMLOCK hard: ", (unsigned long long)rl.rlim_max, rl.rlim_max == RLIM_INFINITY);
    }
    const int sfd = open("/proc/self/status", O_RDONLY | O_CLOEXEC);
    if (sfd < 0) return;
    static char buf[8192];
    ssize_t got = read(sfd, buf, sizeof buf - 1);
    close(sfd);
    if (got <= 0) return;
    buf[got] = 0;
    static const char* keys[] = {"VmLck:", "VmPin:", "VmRSS:", "VmHWM:", "Threads:"};
    for (unsigned k = 0; k < sizeof keys / sizeof keys[0]; ++k) {
        const char* p = strstr(buf, keys[k]);
        if (!p) continue;
        const char* e = strchr(p, '\n');
        (void)!write(fd, "  ", 2);
        (void)!write(fd, p, e ? (size_t)(e - p + 1) : strlen(p));
    }
}

static void on_fatal(int sig)
{
    static const char msg[] = "\n[abrt_trace] fatal signal, native backtrace of the faulting thread:\n";
    void* frames[96];
    const int n = backtrace(frames, 96);
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    dump_memlock(2);
    if (g_fd >= 0) {
        (void)!write(g_fd, msg, sizeof msg - 1);
        backtrace_symbols_fd(frames, n, g_fd);
        dump_memlock(g_fd);
        (void)fsync(g_fd);
    }
    if (sig > 0 && sig < 65) sigaction(sig, &g_prev[sig], 0);      /* the previous owner (faulthandler, or SIG_DFL) */
    else signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    void* warm[4];
    (void)backtrace(warm, 4);                 /* loads libgcc_s now, not inside the handler */
    const char* path = getenv("MI_ABRT_TRACE_FILE");
    if (!path || !*path) {
        (void)mkdir("gpurun_out", 0777);
        path = access("gpurun_out", W_OK) == 0 ? "gpurun_out/abrt_trace.log" : "/tmp/mi_abrt_trace.log";
    }
    g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND | O_CLOEXEC, 0666);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fatal;
    sa.sa_flags = SA_NODEFER | SA_RESETHAND;
    sigaction(SIGABRT, &sa, &g_prev[SIGABRT]);
    sigaction(SIGSEGV, &sa, &g_prev[SIGSEGV]);
    sigaction(SIGBUS, &sa, &g_prev[SIGBUS]);
}
