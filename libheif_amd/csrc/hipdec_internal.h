// hipdec_internal.h — internal glue shared by the host side of libheifhip.so
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <new>
#include <exception>
#include "heif_hipdec.h"

namespace hipdec {

int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int ensure_init();
// The device hipdec_init() selected — or, inside a DeviceScope on this thread, the scope's device: arenas, streams and launches of a
// multi-device grid decode (grid.hip) are created under the scope of the shard's device.
int active_device();
class DeviceScope {
 public:
  explicit DeviceScope(int device);
  ~DeviceScope();
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
 private:
  int prev_;
};
hipStream_t default_stream();
hipStream_t post_stream();      // everything behind the CABAC kernel when stage overlap is on (hipdec_set_stage_overlap)
bool stage_overlap();
int max_devices();              // size of the per-device stream table: device indices beyond it are refused
hipStream_t upload_stream();    // H2D copies of large batches (overlaps the kernels of the batch before)
uint32_t parse_wave_budget();   // CABAC pool waves one batch may launch (wave slots / concurrent batches)

// Process-wide pool of decode arenas.  libheif creates and destroys one plugin decoder per item and per grid tile
// (libheif/codecs/decoder.cc:388-405,549), so arenas are recycled by size instead of hipMalloc / hipFree per image.
hipError_t arena_acquire(void** out, size_t bytes, size_t* capacity);
void arena_release(void* p, size_t capacity);
void arena_pool_clear();
bool arena_oom_take();   // did an arena_acquire of this thread fail since the last call?  (clears the mark: a chain that ran out of device memory is retried shorter)
void set_memory_pressure_handler(void (*fn)());   // called when a device allocation fails even with the arena pool empty, before the last retry
hipError_t pinned_acquire(void** out, size_t bytes, size_t* capacity);   // pinned host staging for large uploads, recycled
void pinned_release(void* p, size_t capacity);
void pinned_pool_clear();
int32_t* status_slot_acquire();                 // pinned 64-byte slot for a batch's status word (NULL: none left)
void status_slot_release(int32_t* p);

// A small pool of HIP streams: concurrent plugin decoder instances (libheif decodes grid tiles on several threads,
// libheif/image-items/grid.cc:436) each run on their own stream so that their kernels overlap on the GPU.
hipStream_t stream_acquire();
hipStream_t stream_acquire_priority();   // for image-level calls beside the decoder's launch sets (released with stream_release)
void stream_release(hipStream_t s);

// the stream follow-up work of a batch goes on (made to wait for the batch's last recorded work when that ran elsewhere: decoder.hip follow_stream)
}  // namespace hipdec
struct hipdec_batch;
namespace hipdec {
hipStream_t batch_follow_stream(hipdec_batch* b, hipStream_t s);

// Colour stage of a whole batch as ONE launch (color.hip): between begin and launch this thread's hipdec_color_* calls record
// their parameter blocks instead of launching; the blocks live in a device array owned by the batch.
struct ColorBatchState {
  void* dev = nullptr;
  size_t dev_bytes = 0;
  std::vector<uint8_t> host;   // the blocks last uploaded
  std::vector<uint8_t> prev;   // the upload before: kept alive while a copy from it may still be pending (pageable source of an async copy)
};
void color_capture_begin();
void color_capture_abort();
int color_capture_launch(ColorBatchState& st, hipStream_t s);
int color_capture_take(ColorBatchState& st, hipStream_t s, const void** dev, int* uniform_variant, int* count);   // upload only
int color_variant_rgb24_u8();
void color_batch_state_free(ColorBatchState& st);

// No C++ exception may cross the C ABI (the caller is libheif, or cgo / JNI / ctypes): every entry point that parses untrusted
// input or allocates runs its body through guarded().
template <class F> int guarded(const char* what, F&& body)
{
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return set_error(HIPDEC_ERR_MEMORY, "%s: out of host memory", what);
  } catch (const std::exception& e) {
    return set_error(HIPDEC_ERR_BITSTREAM, "%s: %s", what, e.what());
  } catch (...) {
    return set_error(HIPDEC_ERR_BITSTREAM, "%s: unknown failure", what);
  }
}

#define HIPDEC_CHECK_HIP(expr)                                                                  \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return hipdec::set_error(HIPDEC_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

}  // namespace hipdec
