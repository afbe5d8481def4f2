// debug aid: print the C backtrace of the thread that raised SIGABRT / SIGSEGV (loaded with ctypes by scripts/dbg/*.py).
// The handler runs on an alternate stack so that a stack overflow (runaway recursion) still gets its trace.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
#include <stdlib.h>
static void handler(int sig) {
  void* frames[48];
  int n = backtrace(frames, 48);
  const char* msg = sig == SIGSEGV ? "\n==== SIGSEGV backtrace (innermost 48 frames) ====\n" : "\n==== SIGABRT backtrace ====\n";
  write(2, msg, strlen(msg));
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
void install_abrt_trace(void) {
  stack_t ss;
  ss.ss_sp = malloc(1 << 18);
  ss.ss_size = 1 << 18;
  ss.ss_flags = 0;
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sa.sa_flags = SA_ONSTACK;
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGSEGV, &sa, 0);
}
