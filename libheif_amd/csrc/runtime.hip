// runtime.hip — library life cycle, error reporting and the small device-memory helpers of the
// C ABI (include/heif_hipdec.h).  The product path has NO CPU fallback: without a HIP device every
// compute entry point fails with HIPDEC_ERR_DEVICE.
#include "hipdec_internal.h"
#include <atomic>
#include <algorithm>
#include <unistd.h>
#include <mutex>
#include <utility>
#include <vector>
#include <cstdint>

namespace hipdec {

static thread_local std::string t_last_error = "";
static std::mutex g_init_mutex;
static std::atomic<bool> g_initialised{false};   // read without the mutex by every entry point (ensure_init): release-stored once everything below is set up
static std::atomic<int> g_device{0};          // the device hipdec_init() selected: what every entry point uses ...
static thread_local int t_device_override = -1;   // ... unless a DeviceScope is active on this thread (multi-device grid decode)
constexpr int kMaxDevices = 16;
struct DeviceStreams { hipStream_t stream = nullptr, upload = nullptr, post = nullptr; };
static std::atomic<int> g_stage_overlap{0};
static DeviceStreams g_streams[kMaxDevices];   // created on first use of a device, destroyed by hipdec_shutdown()
static std::mutex g_streams_mu;
static std::atomic<int> g_cu_count{256};      // compute units of the selected device
static std::atomic<int> g_concurrent{1};      // batches the host keeps in flight at a time (hipdec_set_concurrent_batches)
static std::atomic<int> g_reserved_slots{getenv("HIPDEC_RESERVED_WAVE_SLOTS") ? atoi(getenv("HIPDEC_RESERVED_WAVE_SLOTS")) : 0};   // wave slots per SIMD the
                                              // CABAC pools leave free (hipdec_set_reserved_wave_slots)

int set_error(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  t_last_error = buf;
  return code;
}

int active_device() { return t_device_override >= 0 ? t_device_override : g_device.load(std::memory_order_relaxed); }

int ensure_init()
{
  if (g_initialised.load(std::memory_order_acquire)) {
    // every host thread needs the device selected once; hipSetDevice is cheap
    const int dev = active_device();
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) return set_error(HIPDEC_ERR_DEVICE, "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
    return 0;
  }
  return hipdec_init(-1);
}

DeviceScope::DeviceScope(int device) : prev_(t_device_override)
{
  if (device >= 0 && device != active_device()) { t_device_override = device; (void)hipSetDevice(device); }
  else if (device >= 0) t_device_override = device;
}
DeviceScope::~DeviceScope()
{
  const int was = active_device();
  t_device_override = prev_;
  if (active_device() != was) (void)hipSetDevice(active_device());
}

static DeviceStreams& streams_of_active_device()
{
  // (device indices are validated where they enter the library: hipdec_init, hipdec_grid_create — max_devices())
  int dev = active_device();
  if (dev < 0 || dev >= kMaxDevices) dev = 0;
  DeviceStreams& d = g_streams[dev];
  {
    std::lock_guard<std::mutex> lock(g_streams_mu);   // always taken: the creation below must not race a reader of d.stream
    if (!d.stream) {
      (void)hipSetDevice(dev);
      hipStream_t up = nullptr, st = nullptr, po = nullptr;
      (void)hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
      (void)hipStreamCreateWithFlags(&po, hipStreamNonBlocking);
      (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      d.upload = up;
      d.post = po;
      d.stream = st;
    }
  }
  return d;
}
int max_devices() { return kMaxDevices; }
hipStream_t default_stream() { return streams_of_active_device().stream; }
hipStream_t upload_stream() { DeviceStreams& d = streams_of_active_device(); return d.upload ? d.upload : d.stream; }
hipStream_t post_stream() { DeviceStreams& d = streams_of_active_device(); return d.post ? d.post : d.stream; }
bool stage_overlap() { return g_stage_overlap.load(std::memory_order_relaxed) != 0; }

// Waves of the CABAC work pool one batch may launch: the pool only works while ALL its waves are resident (8 per SIMD with
// the kernel's register budget), so concurrent batches have to share the machine's wave slots.
uint32_t parse_wave_budget()
{
  const int c = g_concurrent.load(std::memory_order_relaxed);
  // k_parse_occ8: 64 VGPRs, 8 waves per SIMD.  A host that runs image-level kernels (colour conversion, transformations) beside decoding keeps
  // one of them free: the pool's waves are resident for the whole launch set, and a small kernel queued behind a full chip waits for the
  // set to end (measured, tools/concurrency_probe.py: one 4K colour conversion 354 ms beside an 8-per-SIMD pool, 2 ms beside a 7-per-SIMD one)
  int reserved = g_reserved_slots.load(std::memory_order_relaxed);
  reserved = reserved < 0 ? 0 : (reserved > 4 ? 4 : reserved);
  const uint32_t slots = (uint32_t)g_cu_count.load(std::memory_order_relaxed) * 4u * (uint32_t)(8 - reserved);
  return slots / (uint32_t)(c < 1 ? 1 : c);
}

namespace {
struct ArenaEntry { size_t first; void* second; int device; };   // (capacity, pointer, device that owns the allocation)
struct ArenaPool {
  std::mutex mu;
  std::vector<ArenaEntry> free_list;
  size_t cached_bytes = 0;
};
ArenaPool g_pool;
// Defaults suit the plugin path (one arena per item / grid photo).  A throughput host that streams large batches raises both
// with hipdec_set_arena_cache_bytes(): hipFree() of a batch arena synchronises the device, which would serialise the
// double-buffered "parse + upload batch k+1 while batch k decodes" pipeline.
// (round 3: sized for the 288 GB of an MI355X.  The decoder path builds a batch per launch set — hundreds of stills, several GiB —
//  and with the round-2 defaults (8 GiB parked, arenas above 1 GiB never) every set paid a hipMalloc and a device-synchronising hipFree.)
// (round 4, ADVICE: the caps are derived from the device at hipdec_init — a third / two ninths of its memory, i.e. 96 / 64 GiB on the 288 GB of
//  an MI355X and proportionally less on a smaller or shared device; hipdec_set_arena_cache_bytes() or HIPDEC_ARENA_CACHE_GIB override them.)
std::atomic<size_t> g_max_cached_bytes{size_t(8) << 30};    // keep at most this much parked
std::atomic<size_t> g_max_pooled_arena{size_t(4) << 30};    // bigger arenas are not cached

struct PinnedPool {
  std::mutex mu;
  std::vector<std::pair<size_t, void*>> free_list;
  size_t cached_bytes = 0;
};
std::atomic<size_t> g_max_pinned_cached{size_t(2) << 30};   // pinned buffers parked for reuse (upload staging, plane staging of the decoder path): an eighth
PinnedPool g_pinned;                                         // of the host's RAM, at most 24 GiB (set at hipdec_init)
}  // namespace

// Pinned host staging buffers for the upload region of large batches (so that the H2D copy is asynchronous and runs at PCIe
// speed); recycled, because hipHostMalloc of a GiB costs more than the copy it feeds.
hipError_t pinned_acquire(void** out, size_t bytes, size_t* capacity)
{
  // size classes: 256 KiB steps up to 4 MiB, 1 MiB steps up to 64 MiB, 16 MiB steps above (a launch set of thumbnails must not pin 16 MiB each)
  const size_t kClass = bytes <= (size_t(4) << 20) ? (size_t(256) << 10) : (bytes <= (size_t(64) << 20) ? (size_t(1) << 20) : (size_t(16) << 20));
  bytes = (bytes + kClass - 1) / kClass * kClass;
  {
    std::lock_guard<std::mutex> lock(g_pinned.mu);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < g_pinned.free_list.size(); i++) {
      const size_t cap = g_pinned.free_list[i].first;
      if (cap >= bytes && cap <= 2 * bytes && (best == SIZE_MAX || cap < g_pinned.free_list[best].first)) best = i;
    }
    if (best != SIZE_MAX) {
      *out = g_pinned.free_list[best].second; *capacity = g_pinned.free_list[best].first;
      g_pinned.cached_bytes -= *capacity;
      g_pinned.free_list[best] = g_pinned.free_list.back();
      g_pinned.free_list.pop_back();
      return hipSuccess;
    }
  }
  *capacity = bytes;
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {   // out of pinnable memory: hand the parked buffers back and retry once (as arena_acquire does)
    (void)hipGetLastError();
    pinned_pool_clear();
    e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  }
  return e;
}

void pinned_release(void* p, size_t capacity)
{
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_pinned.mu);
    if (g_pinned.cached_bytes + capacity <= g_max_pinned_cached.load() && g_pinned.free_list.size() < 1024) {
      g_pinned.free_list.emplace_back(capacity, p); g_pinned.cached_bytes += capacity; return;
    }
  }
  (void)hipHostFree(p);
}

// One pinned 64-byte slot per live batch for the status word its last run copies back (decoder.hip): carved out of a slab so that
// creating / retiring a batch costs no hipHostMalloc / hipHostFree (the latter may synchronise the device).
namespace {
struct StatusSlab {
  std::mutex mu;
  uint8_t* base = nullptr;
  std::vector<int32_t*> free_list;
};
StatusSlab g_status;
constexpr size_t kStatusSlots = 4096;
}  // namespace

int32_t* status_slot_acquire()
{
  std::lock_guard<std::mutex> lock(g_status.mu);
  if (!g_status.base) {
    if (hipHostMalloc((void**)&g_status.base, kStatusSlots * 64, hipHostMallocDefault) != hipSuccess) { g_status.base = nullptr; return nullptr; }
    for (size_t i = kStatusSlots; i-- > 0;) g_status.free_list.push_back((int32_t*)(g_status.base + i * 64));
  }
  if (g_status.free_list.empty()) return nullptr;
  int32_t* p = g_status.free_list.back();
  g_status.free_list.pop_back();
  return p;
}

void status_slot_release(int32_t* p)
{
  if (!p) return;
  std::lock_guard<std::mutex> lock(g_status.mu);
  g_status.free_list.push_back(p);
}

void pinned_pool_clear()
{
  std::lock_guard<std::mutex> lock(g_pinned.mu);
  for (auto& e : g_pinned.free_list) (void)hipHostFree(e.second);
  g_pinned.free_list.clear();
  g_pinned.cached_bytes = 0;
}

static std::atomic<void (*)()> g_memory_pressure{nullptr};
static thread_local bool t_arena_oom = false;   // the calling thread's last arena_acquire failed for want of device memory
bool arena_oom_take() { const bool v = t_arena_oom; t_arena_oom = false; return v; }
void set_memory_pressure_handler(void (*fn)()) { g_memory_pressure.store(fn); }

hipError_t arena_acquire(void** out, size_t bytes, size_t* capacity)
{
  // round small arenas up so that items of similar size share a class
  // size classes: 4 MiB steps up to 64 MiB, then a quarter of the largest power of two below the size (at most 25 % slack, a handful of
  // classes per octave), so that launch sets of varying size share parked arenas
  size_t kClass = size_t(4) << 20;
  if (bytes > (size_t(64) << 20)) { size_t p2 = size_t(1) << 26; while ((p2 << 1) <= bytes) p2 <<= 1; kClass = p2 >> 2; }
  if (bytes <= g_max_pooled_arena.load()) bytes = (bytes + kClass - 1) / kClass * kClass;
  {
    std::lock_guard<std::mutex> lock(g_pool.mu);
    size_t best = SIZE_MAX;
    const int dev = active_device();
    for (size_t i = 0; i < g_pool.free_list.size(); i++) {
      const size_t cap = g_pool.free_list[i].first;
      if (g_pool.free_list[i].device != dev) continue;
      if (cap >= bytes && cap <= 2 * bytes && (best == SIZE_MAX || cap < g_pool.free_list[best].first)) best = i;
    }
    if (best != SIZE_MAX) {
      *out = g_pool.free_list[best].second; *capacity = g_pool.free_list[best].first;
      g_pool.cached_bytes -= *capacity;
      g_pool.free_list.erase(g_pool.free_list.begin() + (long)best);
      return hipSuccess;
    }
  }
  const size_t cap = bytes;
  hipError_t e = hipMalloc(out, cap);
  if (e != hipSuccess) {   // out of memory: drop the cache and retry once
    arena_pool_clear();
    e = hipMalloc(out, cap);
  }
  if (e != hipSuccess) {   // still none: whatever only a cache keeps alive goes too (the resident-plane registry of decoder.hip pins whole batch arenas)
    (void)hipGetLastError();
    if (auto cb = g_memory_pressure.load()) { cb(); arena_pool_clear(); e = hipMalloc(out, cap); }
  }
  *capacity = cap;
  if (e != hipSuccess) t_arena_oom = true;
  return e;
}

void arena_release(void* p, size_t capacity)
{
  if (!p) return;
  if (capacity <= g_max_pooled_arena.load()) {
    std::lock_guard<std::mutex> lock(g_pool.mu);
    if (g_pool.cached_bytes + capacity <= g_max_cached_bytes.load()) {
      g_pool.free_list.push_back(ArenaEntry{capacity, p, active_device()});
      g_pool.cached_bytes += capacity;
      return;
    }
  }
  (void)hipFree(p);
}

namespace {
std::mutex g_stream_mu;
std::vector<std::pair<hipStream_t, int>> g_free_streams;   // (stream, device)
}  // namespace

// HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), in order within a queue.  Pipelined look-ahead chains
// (decoder_chains.inc) keep three launch sets of two streams each in flight, and a CABAC launch that shares a hardware queue with the pixel steps of
// the set in front of it waits for them instead of running beside them (measured on MI355X, one 720p IPPP track, pipeline 3: 370 fps with 4
// queues, 559 fps with 16; round 5 saw the same for tracks side by side: profiles/r05_sequence_fps.txt, call 12).  The runtime reads the variable
// when it initialises - at the process's first HIP call, after this library was loaded -, so it is set here unless the host has chosen a value.
__attribute__((constructor)) static void hipdec_more_hardware_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }

hipStream_t stream_acquire()
{
  const int dev = active_device();
  {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    for (size_t i = g_free_streams.size(); i-- > 0;)
      if (g_free_streams[i].second == dev) { hipStream_t s = g_free_streams[i].first; g_free_streams.erase(g_free_streams.begin() + (long)i); return s; }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return default_stream();   // fall back to the shared stream
  return s;
}

// Streams of the image-level calls (colour conversion, transformations): created with the device's highest priority.  HIP multiplexes its
// streams onto a few hardware queues, in order within a queue: a conversion on an ordinary stream shares a queue with some launch set's CABAC
// kernel and waits for it to end (measured through libheif, 256 threads: 218 ms per conversion); priority streams live on queues of their own,
// and the command processor serves them first.
namespace {
std::vector<std::pair<hipStream_t, int>> g_free_prio_streams;
std::vector<hipStream_t> g_all_prio_streams;
}  // namespace

hipStream_t stream_acquire_priority()
{
  const int dev = active_device();
  {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    for (size_t i = g_free_prio_streams.size(); i-- > 0;)
      if (g_free_prio_streams[i].second == dev) { hipStream_t s = g_free_prio_streams[i].first; g_free_prio_streams.erase(g_free_prio_streams.begin() + (long)i); return s; }
  }
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); return stream_acquire(); }
  std::lock_guard<std::mutex> lock(g_stream_mu);
  g_all_prio_streams.push_back(s);
  return s;
}

void stream_release(hipStream_t s)
{
  if (!s) return;
  {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    for (hipStream_t p : g_all_prio_streams)
      if (p == s) {
        if (g_free_prio_streams.size() < 256) { g_free_prio_streams.emplace_back(s, active_device()); return; }
        g_all_prio_streams.erase(std::find(g_all_prio_streams.begin(), g_all_prio_streams.end(), s));
        (void)hipStreamDestroy(s);
        return;
      }
  }
  for (const auto& d : g_streams) if (s == d.stream || s == d.upload || s == d.post) return;
  std::lock_guard<std::mutex> lock(g_stream_mu);
  if (g_free_streams.size() < 64) g_free_streams.emplace_back(s, active_device()); else (void)hipStreamDestroy(s);
}

void arena_pool_clear()
{
  std::lock_guard<std::mutex> lock(g_pool.mu);
  for (auto& e : g_pool.free_list) (void)hipFree(e.second);
  g_pool.free_list.clear();
  g_pool.cached_bytes = 0;
}

}  // namespace hipdec

using namespace hipdec;

extern "C" {

int hipdec_init(int device_index)
{
  std::lock_guard<std::mutex> lock(g_init_mutex);
  if (g_initialised.load(std::memory_order_relaxed) && (device_index < 0 || device_index == g_device.load(std::memory_order_relaxed))) return 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(HIPDEC_ERR_DEVICE, "no HIP device available (%s); libheifhip has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  int dev = device_index < 0 ? 0 : device_index;
  if (dev >= n || dev >= kMaxDevices) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices, at most %d supported)", dev, n, kMaxDevices);
  HIPDEC_CHECK_HIP(hipSetDevice(dev));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) g_cu_count = cus;
    // cache caps from the device and the host
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
      size_t cached = total_b / 3, single = total_b / 9 * 2;
      if (const char* q = getenv("HIPDEC_ARENA_CACHE_GIB")) { cached = (size_t)std::max(0L, atol(q)) << 30; single = cached; }
      g_max_cached_bytes.store(cached);
      g_max_pooled_arena.store(std::max(single, size_t(1) << 30));
    }
    const long pages = sysconf(_SC_PHYS_PAGES), page = sysconf(_SC_PAGESIZE);
    if (pages > 0 && page > 0) g_max_pinned_cached.store(std::min((size_t)pages * (size_t)page / 8, size_t(24) << 30));
  }
  const int prev_dev = g_device.load(std::memory_order_relaxed);
  g_device.store(dev, std::memory_order_relaxed);   // (default_stream() is the stream of the active device)
  if (!default_stream()) { g_device.store(prev_dev, std::memory_order_relaxed); return set_error(HIPDEC_ERR_DEVICE, "could not create a HIP stream on device %d", dev); }
  // published LAST: a thread that finds the flag set in ensure_init() skips this mutex (a plain bool read there was a data race ThreadSanitizer
  // reported when application threads made the library's first calls concurrently - and the flag used to be set before the stream existed)
  g_initialised.store(true, std::memory_order_release);
  return 0;
}

void hipdec_shutdown(void)
{
  std::lock_guard<std::mutex> lock(g_init_mutex);
  if (!g_initialised.load(std::memory_order_relaxed)) return;
  hipdec_forget_resident_planes();
  arena_pool_clear();
  pinned_pool_clear();
  {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    for (auto& st : g_free_streams) (void)hipStreamDestroy(st.first);
    g_free_streams.clear();
    for (auto& st : g_free_prio_streams) (void)hipStreamDestroy(st.first);
    g_free_prio_streams.clear(); g_all_prio_streams.clear();
  }
  {
    std::lock_guard<std::mutex> lock(g_streams_mu);
    for (auto& d : g_streams) {
      if (d.stream) (void)hipStreamDestroy(d.stream);
      if (d.upload) (void)hipStreamDestroy(d.upload);
      if (d.post) (void)hipStreamDestroy(d.post);
      d = DeviceStreams{};
    }
  }
  g_initialised.store(false, std::memory_order_release);
}

int hipdec_set_arena_cache_bytes(size_t bytes)
{
  g_max_cached_bytes.store(bytes);
  g_max_pooled_arena.store(bytes > (size_t(1) << 30) ? bytes : (size_t(1) << 30));
  if (bytes == 0) { arena_pool_clear(); pinned_pool_clear(); }
  return 0;
}

int hipdec_set_stage_overlap(int on)
{
  g_stage_overlap.store(on ? 1 : 0, std::memory_order_relaxed);
  return 0;
}

int hipdec_set_concurrent_batches(int n)
{
  if (n < 1 || n > 64) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "set_concurrent_batches: n must be in [1, 64]");
  g_concurrent.store(n, std::memory_order_relaxed);
  return 0;
}

int hipdec_set_reserved_wave_slots(int per_simd)
{
  if (per_simd < 0 || per_simd > 4) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "set_reserved_wave_slots: 0 .. 4 wave slots per SIMD");
  if (!getenv("HIPDEC_RESERVED_WAVE_SLOTS")) g_reserved_slots.store(per_simd, std::memory_order_relaxed);
  return 0;
}

const char* hipdec_last_error(void) { return t_last_error.c_str(); }
const char* hipdec_version(void) { return "libheifhip 0.1 (gfx950)"; }

int hipdec_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void* hipdec_malloc(size_t bytes)
{
  if (ensure_init()) return nullptr;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) { set_error(HIPDEC_ERR_DEVICE, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
  return p;
}
void hipdec_free(void* dptr) { if (dptr) (void)hipFree(dptr); }
int hipdec_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
  if (int rc = ensure_init()) return rc;
  HIPDEC_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int hipdec_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
  if (int rc = ensure_init()) return rc;
  HIPDEC_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int hipdec_memset(void* dst, int value, size_t bytes)
{
  if (int rc = ensure_init()) return rc;
  HIPDEC_CHECK_HIP(hipMemset(dst, value, bytes));
  return 0;
}
void* hipdec_stream_create(void)
{
  if (ensure_init()) return nullptr;
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) { set_error(HIPDEC_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); return nullptr; }
  return (void*)s;
}
void hipdec_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }
int hipdec_stream_synchronize(void* stream)
{
  if (int rc = ensure_init()) return rc;
  HIPDEC_CHECK_HIP(hipStreamSynchronize(stream ? (hipStream_t)stream : default_stream()));
  return 0;
}

}  // extern "C"
