/* CPU-TEST-ONLY: GCC 11's libtsan has no interceptor for pthread_cond_clockwait (what libstdc++'s condition_variable::wait_until(steady_clock) calls);
 * linked into a ThreadSanitizer executable this definition takes precedence over glibc's and goes through the intercepted pthread_cond_timedwait
 * (see tests/emu/tsan_host.cc; tools/emu_tsan_libheif.sh links it into the C host). */
#define _GNU_SOURCE
#include <pthread.h>
#include <time.h>
int pthread_cond_clockwait(pthread_cond_t* c, pthread_mutex_t* m, clockid_t clk, const struct timespec* abstime)
{
  struct timespec now_clk, now_rt, t;
  long long ns;
  clock_gettime(clk, &now_clk);
  clock_gettime(CLOCK_REALTIME, &now_rt);
  ns = ((long long)abstime->tv_sec - now_clk.tv_sec) * 1000000000ll + (abstime->tv_nsec - now_clk.tv_nsec);
  if (ns < 0) ns = 0;
  ns += (long long)now_rt.tv_sec * 1000000000ll + now_rt.tv_nsec;
  t.tv_sec = (time_t)(ns / 1000000000ll); t.tv_nsec = (long)(ns % 1000000000ll);
  return pthread_cond_timedwait(c, m, &t);
}
