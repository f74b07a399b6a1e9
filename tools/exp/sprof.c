// A flat sampling profiler for the host stages (no perf / gdb in the image): SIGALRM every `period_us` of wall time (ITIMER_PROF only ticks with the kernel's 4 ms jiffies), the handler stores the interrupted
// program counter; tools/exp/sprof.py maps them to functions with `nm`.   gcc -O2 -shared -fPIC -o tools/exp/bin/libsprof.so tools/exp/sprof.c
#define _GNU_SOURCE
#include <signal.h>
#include <stdint.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>

#define CAP (1 << 20)
static uint64_t g_pc[CAP], g_ret[CAP];  // the interrupted pc, and the word on top of its stack (the return address while a leaf like memcpy / memset runs)
static volatile uint32_t g_n;

static void on_prof(int sig, siginfo_t *si, void *uc_) {
    (void)sig; (void)si;
    ucontext_t *uc = (ucontext_t *)uc_;
    uint32_t i = g_n;
    if (i < CAP) { g_pc[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP]; g_ret[i] = *(const uint64_t *)uc->uc_mcontext.gregs[REG_RSP]; g_n = i + 1; }
}

void sprof_start(int period_us) {
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGALRM, &sa, 0);
    g_n = 0;
    struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = period_us; it.it_value = it.it_interval;
    setitimer(ITIMER_REAL, &it, 0);
}

uint32_t sprof_stop(void) {
    struct itimerval it; memset(&it, 0, sizeof it);
    setitimer(ITIMER_REAL, &it, 0);
    return g_n;
}

const uint64_t *sprof_samples(void) { return g_pc; }
const uint64_t *sprof_returns(void) { return g_ret; }
