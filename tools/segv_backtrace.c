/* LD_PRELOAD helper: prints a native backtrace on SIGSEGV/SIGABRT (the GPU boxes have no gdb).
 *   gcc -shared -fPIC -o build/libsegv.so tools/segv_backtrace.c
 *   LD_PRELOAD=build/libsegv.so python -m pytest -p no:faulthandler ...                          */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <ucontext.h>

static void on_fault(int sig, siginfo_t* si, void* ctx)
{
  void* frames[64];
  char msg[96];
  int n = snprintf(msg, sizeof msg, "\n== signal %d at address %p ==\n", sig, si ? si->si_addr : 0);
  (void)!write(2, msg, (size_t)n);
#if defined(__x86_64__)
  {
    ucontext_t* uc = (ucontext_t*)ctx;
    n = snprintf(msg, sizeof msg, "rip=%p rsp=%p\n", (void*)uc->uc_mcontext.gregs[REG_RIP], (void*)uc->uc_mcontext.gregs[REG_RSP]);
    (void)!write(2, msg, (size_t)n);
  }
#endif
  int k = backtrace(frames, 64);
  backtrace_symbols_fd(frames, k, 2);
  FILE* f = fopen("/proc/self/maps", "r");
  if (f) {
    char line[512];
    while (fgets(line, sizeof line, f))
      if (strstr(line, "r-xp") && (strstr(line, "heif") || strstr(line, "amdhip") || strstr(line, "hsa")))
        (void)!write(2, line, strlen(line));
    fclose(f);
  }
  _exit(139);
}

__attribute__((constructor)) static void install(void)
{
  static char altstack[1 << 16];
  stack_t ss;
  ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0;
  sigaltstack(&ss, 0);   /* a stack overflow leaves no room for the handler on the faulting stack */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
