// hipdec_internal.h — internal glue shared by the host side of libheifhip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include "heif_hipdec.h"

namespace hipdec {

int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int ensure_init();
hipStream_t default_stream();

#define HIPDEC_CHECK_HIP(expr)                                                                  \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return hipdec::set_error(HIPDEC_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

}  // namespace hipdec
