// crash_trace.h -- diagnostic aid of the example drivers: on SIGSEGV / SIGABRT / SIGBUS print which thread died (id, name), the
// faulting address and program counter, the library that was mapped at that program counter when main() started and whether it is
// still mapped now, and the call stack (module + offset), then die by the same signal.  It masks nothing -- the exit status is still
// the signal -- but a crash that happens once in a few hundred runs leaves its address behind.  Background:
// examples/strong_hip has died during process exit, after its validated result and timing line, about once in a few hundred runs;
// every host path of this repository is clean under AddressSanitizer (tests/test_sanitized_host_paths.py).
#pragma once
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>

namespace brk_crash {
constexpr int kMapsMax = 1 << 17;
inline char g_maps[kMapsMax]; // /proc/self/maps as it was when main() started
inline int g_maps_len = 0;

inline void say(const char *s) { (void)!write(2, s, strlen(s)); }
inline void say_hex(unsigned long v) {
  char tmp[20] = "0x";
  int n = 2;
  bool started = false;
  for (int sh = 60; sh >= 0; sh -= 4) {
    const int d = (int)((v >> sh) & 15);
    if (d || started || sh == 0) tmp[n++] = (char)(d < 10 ? '0' + d : 'a' + d - 10), started = true;
  }
  tmp[n] = 0;
  say(tmp);
}
inline void say_dec(long v) {
  char tmp[24];
  int n = 0;
  do tmp[n++] = (char)('0' + v % 10), v /= 10; while (v);
  char out[24];
  int m = 0;
  while (n) out[m++] = tmp[--n];
  out[m] = 0;
  say(out);
}
inline int slurp(const char *path, char *buf, int cap) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return 0;
  int len = 0;
  for (;;) {
    const long r = read(fd, buf + len, (size_t)(cap - 1 - len));
    if (r <= 0) break;
    len += (int)r;
    if (len >= cap - 1) break;
  }
  close(fd);
  buf[len] = 0;
  return len;
}
// the line of a maps text whose [lo, hi) contains addr, printed to stderr; false when none does
inline bool say_mapping(const char *maps, int len, unsigned long addr) {
  int p = 0;
  while (p < len) {
    int e = p;
    while (e < len && maps[e] != '\n') ++e;
    unsigned long lo = 0, hi = 0;
    int q = p;
    auto hex = [&](unsigned long &v) {
      for (; q < e; ++q) {
        const char c = maps[q];
        const int d = c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1);
        if (d < 0) break;
        v = v * 16 + (unsigned long)d;
      }
    };
    hex(lo);
    ++q; // '-'
    hex(hi);
    if (addr >= lo && addr < hi) {
      (void)!write(2, maps + p, (size_t)(e - p));
      say("\n");
      return true;
    }
    p = e + 1;
  }
  return false;
}
inline void handler(int sig, siginfo_t *info, void *uctx) {
  const long tid = (long)syscall(SYS_gettid), pid = (long)getpid();
  const unsigned long pc = (unsigned long)((ucontext_t *)uctx)->uc_mcontext.gregs[REG_RIP];
  say("\n[crash_trace] signal ");
  say(sig == SIGSEGV ? "SIGSEGV" : (sig == SIGABRT ? "SIGABRT" : "SIGBUS"));
  say(" in thread ");
  say_dec(tid);
  say(tid == pid ? " (the main thread)" : " (NOT the main thread, which is ");
  if (tid != pid) say_dec(pid), say(")");
  {
    char path[64] = "/proc/self/task/", name[64];
    int n = (int)strlen(path);
    char tmp[24];
    int m = 0;
    long v = tid;
    do tmp[m++] = (char)('0' + v % 10), v /= 10; while (v);
    while (m) path[n++] = tmp[--m];
    memcpy(path + n, "/comm", 6);
    if (slurp(path, name, sizeof name) > 0) say(", thread name: "), say(name);
    else say("\n");
  }
  say("[crash_trace] fault address ");
  say_hex((unsigned long)info->si_addr);
  say(", program counter ");
  say_hex(pc);
  say("\n[crash_trace] mapped at that program counter when main() started: ");
  if (!say_mapping(g_maps, g_maps_len, pc)) say("nothing\n");
  static char now[kMapsMax];
  const int now_len = slurp("/proc/self/maps", now, kMapsMax);
  say("[crash_trace] mapped there now: ");
  if (!say_mapping(now, now_len, pc)) say("nothing (code that has been unmapped, or a wild jump)\n");
  say("[crash_trace] call stack:\n");
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
  g_maps_len = slurp("/proc/self/maps", g_maps, kMapsMax);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO;
  sigemptyset(&sa.sa_mask);
  const int sigs[3] = {SIGSEGV, SIGABRT, SIGBUS};
  for (int s : sigs) sigaction(s, &sa, nullptr);
}
} // namespace brk_crash
