/* abrt_trace.c -- LD_PRELOAD helper for the fault-injection child process (tests/test_gpu_parity.py): on SIGABRT / SIGSEGV /
 * SIGBUS it writes the NATIVE call chain of the faulting thread to stderr (glibc backtrace, no allocation in the handler path
 * beyond what backtrace_symbols_fd needs) and then lets the default action run.  Diagnostic only. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig)
{
    static const char msg[] = "\n[abrt_trace] fatal signal, native backtrace of the faulting thread:\n";
    void* frames[96];
    (void)!write(2, msg, sizeof msg - 1);
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    void* warm[4];
    (void)backtrace(warm, 4);                 /* loads libgcc_s now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fatal;
    sa.sa_flags = SA_NODEFER | SA_RESETHAND;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
