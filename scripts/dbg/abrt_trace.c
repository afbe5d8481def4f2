// debug aid: print the C backtrace of the thread that raised SIGABRT (loaded with ctypes by scripts/dbg_abort.py)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
static void handler(int sig) {
  void* frames[64];
  int n = backtrace(frames, 64);
  const char* msg = "\n==== SIGABRT backtrace ====\n";
  write(2, msg, strlen(msg));
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
void install_abrt_trace(void) { signal(SIGABRT, handler); signal(SIGSEGV, handler); }
