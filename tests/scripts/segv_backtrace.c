// Test aid (LD_PRELOAD): a native backtrace on SIGSEGV / SIGABRT / SIGBUS, for crashes inside the C libraries that
// Python's faulthandler can only place at the ctypes call.   gcc -shared -fPIC -O1 -o segv_backtrace.so segv_backtrace.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* info, void* ctx) {
  (void)ctx;
  char msg[128];
  int n = snprintf(msg, sizeof(msg), "\n=== native backtrace: signal %d, fault address %p ===\n", sig, info ? info->si_addr : 0);
  if (n > 0) (void)!write(2, msg, (size_t)n);
  void* frames[96];
  int depth = backtrace(frames, 96);
  backtrace_symbols_fd(frames, depth, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
}
