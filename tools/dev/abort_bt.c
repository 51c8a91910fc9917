// LD_PRELOAD shim: print a native backtrace when abort() is called (debugging aid only)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
void abort(void) {
  void* bt[64];
  int n = backtrace(bt, 64);
  const char msg[] = "\n==== abort() called; native backtrace ====\n";
  write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(bt, n, 2);
  void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
  real();
  _exit(134);
}
