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
#include <sys/stat.h>
#include <unistd.h>

static struct sigaction g_prev[65];
static int g_fd = -1;

static void on_fatal(int sig)
{
    static const char msg[] = "\n[abrt_trace] fatal signal, native backtrace of the faulting thread:\n";
    void* frames[96];
    const int n = backtrace(frames, 96);
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    if (g_fd >= 0) {
        (void)!write(g_fd, msg, sizeof msg - 1);
        backtrace_symbols_fd(frames, n, g_fd);
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
