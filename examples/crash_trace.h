// crash_trace.h -- diagnostic aid of the example drivers: on SIGSEGV / SIGABRT / SIGBUS print which thread died (id, name), the
// faulting address and program counter, the library that was mapped at that program counter when main() started and whether it is
// still mapped now, and the call stack (module + offset), then die by the same signal.  It masks nothing -- the exit status is still
// the signal -- but a crash that happens once in a few hundred runs leaves its address behind.  Background:
// examples/strong_hip has died during process exit, after its validated result and timing line, about once in a few hundred runs;
// every host path of this repository is clean under AddressSanitizer (tests/test_sanitized_host_paths.py).
// Round 5 additions: snapshot() may be called again (after GPU initialisation, just before main returns) so that libraries the
// runtime dlopens lazily are in the picture; the handler also prints the nearest mapping BELOW an unmapped program counter and the
// register context (RSP, RBP, the return address at [RSP]); census(label) prints every thread of the process with its id and name --
// BRK_CENSUS=1 in the environment makes the drivers do that at four points of main() (who owns tid = pid + 1).
#pragma once
#include <dirent.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>

namespace brk_crash {
constexpr int kMapsMax = 1 << 17;
inline char g_maps[kMapsMax]; // /proc/self/maps as it was when main() started
inline int g_maps_len = 0;
inline char g_maps_late[kMapsMax]; // ... and at the latest snapshot() (after GPU initialisation / before main returns)
inline int g_maps_late_len = 0;

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
// the line of a maps text with the highest end address <= addr (what lay just below an unmapped address), printed to stderr
inline bool say_mapping_below(const char *maps, int len, unsigned long addr) {
  int p = 0, best_p = -1, best_e = -1;
  unsigned long best_hi = 0;
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
    ++q;
    hex(hi);
    if (hi <= addr && hi > best_hi) best_hi = hi, best_p = p, best_e = e;
    p = e + 1;
  }
  if (best_p < 0) return false;
  (void)!write(2, maps + best_p, (size_t)(best_e - best_p));
  say("\n");
  return true;
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
  say("[crash_trace] mapped there at the latest snapshot (GPU initialised): ");
  if (!say_mapping(g_maps_late, g_maps_late_len, pc)) {
    say("nothing; nearest mapping below it then: ");
    if (!say_mapping_below(g_maps_late, g_maps_late_len, pc)) say("none\n");
  }
  static char now[kMapsMax];
  const int now_len = slurp("/proc/self/maps", now, kMapsMax);
  say("[crash_trace] mapped there now: ");
  if (!say_mapping(now, now_len, pc)) {
    say("nothing (code that has been unmapped, or a wild jump); nearest mapping below it now: ");
    if (!say_mapping_below(now, now_len, pc)) say("none\n");
  }
  {
    const greg_t *gr = ((ucontext_t *)uctx)->uc_mcontext.gregs;
    const unsigned long rsp = (unsigned long)gr[REG_RSP], rbp = (unsigned long)gr[REG_RBP];
    say("[crash_trace] rsp ");
    say_hex(rsp);
    say(", rbp ");
    say_hex(rbp);
    // the word at [rsp] is the return address when the fault is the first instruction after a call through a bad pointer;
    // read it through /proc/self/mem so that a bad stack pointer cannot fault the handler
    const int fd = open("/proc/self/mem", O_RDONLY);
    unsigned long ret = 0;
    if (fd >= 0 && pread(fd, &ret, sizeof ret, (off_t)rsp) == (long)sizeof ret) {
      say(", [rsp] ");
      say_hex(ret);
      say(" which lies in: ");
      if (!say_mapping(now, now_len, ret)) say("nothing mapped\n");
    } else say("\n");
    if (fd >= 0) close(fd);
  }
  say("[crash_trace] call stack:\n");
  void *frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  say("[crash_trace] end\n");
  signal(sig, SIG_DFL);
  raise(sig);
}
// take the picture of /proc/self/maps again: after the first HIP call, before main returns
inline void snapshot() { g_maps_late_len = slurp("/proc/self/maps", g_maps_late, kMapsMax); }
// BRK_CENSUS=2: every other thread is also asked (SIGUSR2 sent to that thread) to print its own call stack, module + offset per
// frame -- which library's code a runtime thread is sitting in
inline void stack_on_request(int, siginfo_t *, void *) {
  say("[census]   thread ");
  say_dec((long)syscall(SYS_gettid));
  say(" is in:\n");
  void *frames[24];
  const int n = backtrace(frames, 24);
  backtrace_symbols_fd(frames, n, 2);
}
// every thread of this process: id, name (comm); printed only when BRK_CENSUS is set.  Not for use inside the handler.
inline void census(const char *label) {
  const char *want = getenv("BRK_CENSUS");
  if (!want || !*want || *want == '0') return;
  const bool stacks = *want == '2';
  if (stacks) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = stack_on_request;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGUSR2, &sa, nullptr);
  }
  long others[8];
  int nothers = 0;
  fprintf(stderr, "[census] %s: pid %ld;", label, (long)getpid());
  if (DIR *d = opendir("/proc/self/task")) {
    while (struct dirent *e = readdir(d)) {
      if (e->d_name[0] == '.') continue;
      char path[96], name[64] = "?";
      snprintf(path, sizeof path, "/proc/self/task/%s/comm", e->d_name);
      const int n = slurp(path, name, sizeof name);
      if (n > 0 && name[n - 1] == '\n') name[n - 1] = 0;
      fprintf(stderr, " %s=%s", e->d_name, name);
      if (atol(e->d_name) != (long)getpid() && nothers < 8) others[nothers++] = atol(e->d_name);
    }
    closedir(d);
  }
  fprintf(stderr, "\n");
  if (stacks && nothers < 8) // (not with the OpenMP team alive: hundreds of threads)
    for (int q = 0; q < nothers; ++q) {
      syscall(SYS_tgkill, (long)getpid(), others[q], SIGUSR2);
      usleep(50000);
    }
}
inline void install() {
  void *warm[2];
  (void)backtrace(warm, 2); // loads libgcc's unwinder now, not inside the handler
  g_maps_len = slurp("/proc/self/maps", g_maps, kMapsMax);
  memcpy(g_maps_late, g_maps, (size_t)g_maps_len + 1);
  g_maps_late_len = g_maps_len;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO;
  sigemptyset(&sa.sa_mask);
  const int sigs[3] = {SIGSEGV, SIGABRT, SIGBUS};
  for (int s : sigs) sigaction(s, &sa, nullptr);
}
} // namespace brk_crash
