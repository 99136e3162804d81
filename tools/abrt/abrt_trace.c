/* LD_PRELOAD aid: who called abort()? Prints the native backtrace and the name of the thread that raised SIGABRT (a runtime or
 * communicator thread has no Python frame, so Python's faulthandler shows nothing for it), then lets the default action run.
 *   gcc -O1 -g -shared -fPIC -o tools/abrt/libabrt_trace.so tools/abrt/abrt_trace.c
 *   LD_PRELOAD=$PWD/tools/abrt/libabrt_trace.so python ...          (tools/rccl_soak.sh TRACE=1 does this) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

static void on_abort(int sig) {
    void* frames[96];
    char name[32] = {0}, head[160];
    prctl(PR_GET_NAME, name, 0, 0, 0);
    int n = snprintf(head, sizeof head, "\n=== abrt_trace: signal %d in thread %ld (%s) of pid %d ===\n", sig, (long)syscall(SYS_gettid), name, (int)getpid());
    if (write(2, head, n) < 0) {}
    n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    if (write(2, "=== end ===\n", 12) < 0) {}
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, NULL);
}
