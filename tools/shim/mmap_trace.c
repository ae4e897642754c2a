// LD_PRELOAD shim: prints a backtrace for every anonymous mmap of 150-260 MB (the size of a hardware compute queue's
// context-save area on MI355X) - who creates a queue in the middle of a run?  tools/stall_probe.py, DESIGN.md section 6.
//   gcc -O1 -g -shared -fPIC tools/shim/mmap_trace.c -o tools/shim/libmmaptrace.so -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

void *mmap(void *addr, size_t len, int prot, int flags, int fd, off_t off) {
  static void *(*real)(void *, size_t, int, int, int, off_t);
  if (!real) real = (void *(*)(void *, size_t, int, int, int, off_t))dlsym(RTLD_NEXT, "mmap");
  void *r = real(addr, len, prot, flags, fd, off);
  if (len > 150000000 && len < 260000000 && (flags & MAP_ANONYMOUS)) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    char buf[128];
    int n = snprintf(buf, sizeof buf, "\n[mmap_trace] %.1f MB at %ld.%03ld s, thread %d\n", len / 1e6, (long)ts.tv_sec, ts.tv_nsec / 1000000, (int)gettid());
    write(2, buf, n);
    void *bt[48];
    int k = backtrace(bt, 48);
    backtrace_symbols_fd(bt, k, 2);
  }
  return r;
}
