/* LD_PRELOAD helper for the GPU box (no debugger there): print the native call stack when the process aborts.
 *   gcc -shared -fPIC -o scripts/dbg/abort_trace.so scripts/dbg/abort_trace.c
 *   LD_PRELOAD=$PWD/scripts/dbg/abort_trace.so python -m pytest ... */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig)
{
    void *frames[64];
    const char msg[] = "\n== native stack at SIGABRT ==\n";
    (void)!write(2, msg, sizeof msg - 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, 0);
}
