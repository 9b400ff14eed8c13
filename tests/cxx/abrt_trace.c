/* abrt_trace.c -- diagnostic for the GPU test session (loaded by tests/conftest.py, or LD_PRELOADed): on SIGABRT / SIGSEGV /
 * SIGBUS it writes the NATIVE call chain of the faulting thread to stderr (glibc backtrace; libgcc_s is loaded at install time,
 * not inside the handler), then hands the signal to whoever was installed before it (Python's faulthandler prints the Python
 * stacks and lets the default action run).  A silent abort() from a runtime thread then shows which library called it. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[65];

static void on_fatal(int sig)
{
    static const char msg[] = "\n[abrt_trace] fatal signal, native backtrace of the faulting thread:\n";
    void* frames[96];
    (void)!write(2, msg, sizeof msg - 1);
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    if (sig > 0 && sig < 65) sigaction(sig, &g_prev[sig], 0);      /* the previous owner (faulthandler, or SIG_DFL) */
    else signal(sig, SIG_DFL);
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
    sigaction(SIGABRT, &sa, &g_prev[SIGABRT]);
    sigaction(SIGSEGV, &sa, &g_prev[SIGSEGV]);
    sigaction(SIGBUS, &sa, &g_prev[SIGBUS]);
}
