// color_emu.cc — CPU-TEST-ONLY: the host-side glue color.hip expects from runtime.hip, so that the colour kernels and their
// C entry points (hipdec_color_*, the very functions of the product) can be compiled for the host against the SIMT emulator
// and checked against the colour oracle (which is pinned to the compiled reference ops) without a GPU.  NOT part of the product.
#include <cstdarg>
#include <cstdio>
#include <string>
#include "hipdec_internal.h"

#ifdef HIPEMU_WHOLE_LIBRARY   // libheifhip_emu.so: runtime.hip itself is compiled in
extern "C" const char* hipdec_last_error(void);
extern "C" const char* emu_color_last_error() { return hipdec_last_error(); }
#else
namespace hipdec {

static thread_local std::string t_err;
int set_error(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_err = buf;
  return code;
}
int ensure_init() { return 0; }
hipStream_t default_stream() { return nullptr; }
uint32_t parse_wave_budget() { return 0; }
hipError_t arena_acquire(void** out, size_t bytes, size_t* capacity) { *capacity = bytes; return hipMalloc(out, bytes); }
void arena_release(void* p, size_t) { (void)hipFree(p); }

}  // namespace hipdec

extern "C" const char* emu_color_last_error() { return hipdec::t_err.c_str(); }
#endif
