// crash_trace.h -- diagnostic aid of the example drivers: on SIGSEGV / SIGABRT / SIGBUS print which thread died and its call stack
// (module + offset: readable without debug symbols) to stderr, then die by the same signal.  It masks nothing -- the exit status is
// still the signal -- but a crash that happens once in a few hundred runs leaves its address behind.  Background: round 3 saw a
// glibc heap abort and round 4 one SIGSEGV of examples/strong_hip AFTER its validated result and timing line had been printed (during
// process exit); every host path of this repository is clean under AddressSanitizer (tests/test_sanitized_host_paths.py) and 340
// further runs, with and without a debugger, exited normally.
#pragma once
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
// <unistd.h> declares the system call brk(), which collides with this repository's namespace brklib: the three functions needed
extern "C" long write(int, const void *, unsigned long);
extern "C" int getpid(void) noexcept;
extern "C" long syscall(long, ...) noexcept;

namespace brk_crash {
inline void say(const char *s) { (void)!write(2, s, strlen(s)); }
inline void handler(int sig) {
  char line[96];
  const long tid = (long)syscall(SYS_gettid), pid = (long)getpid();
  say("\n[crash_trace] signal ");
  say(sig == SIGSEGV ? "SIGSEGV" : (sig == SIGABRT ? "SIGABRT" : "SIGBUS"));
  // async-signal-safe formatting of two integers
  char *p = line;
  auto put = [&](long v) {
    char tmp[24];
    int n = 0;
    do tmp[n++] = (char)('0' + v % 10), v /= 10; while (v);
    while (n) *p++ = tmp[--n];
  };
  *p++ = ' ', *p++ = 'i', *p++ = 'n', *p++ = ' ', *p++ = 't', *p++ = 'h', *p++ = 'r', *p++ = 'e', *p++ = 'a', *p++ = 'd', *p++ = ' ';
  put(tid);
  *p++ = ' ', *p++ = '(', *p++ = 'm', *p++ = 'a', *p++ = 'i', *p++ = 'n', *p++ = ' ', *p++ = 't', *p++ = 'h', *p++ = 'r', *p++ = 'e', *p++ = 'a', *p++ = 'd', *p++ = ' ';
  put(pid);
  *p++ = ')', *p++ = ':', *p++ = '\n', *p = 0;
  say(line);
  void *frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  say("[crash_trace] end\n");
  signal(sig, SIG_DFL);
  raise(sig);
}
inline void install() {
  void *warm[2];
  (void)backtrace(warm, 2); // loads libgcc's unwinder now, not inside the handler
  for (int s : {SIGSEGV, SIGABRT, SIGBUS}) signal(s, handler);
}
} // namespace brk_crash
