// CPU-TEST-ONLY stand-in for <hip/hip_runtime.h>: a small SIMT emulator that lets the `-m "not gpu"` suite compile the
// kernel sources of libheif_amd/csrc (*.hip) with g++ and run them on the host, so that their logic is checked against
// the oracle without a GPU.  NOT part of the product: nothing under libheif_amd/ includes this file.
//
// Model: a workgroup runs on one OS thread; each of its threads is a ucontext coroutine.  Threads run until they reach a
// convergence point (wave barrier, __syncthreads, ballot, shuffle, readfirstlane) and are resumed when all live threads of
// the wave / group have arrived, which is exactly the guarantee the kernels rely on.  Workgroups of one launch are handed
// to a pool of OS threads in launch order, so a group only ever waits (progress words in "HBM") for groups that hold an
// earlier ticket and are therefore running or finished — as on the device.  `__shared__` variables are thread_local
// statics of the OS thread that runs the group.
#pragma once
#include <ucontext.h>
#include <sched.h>
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
typedef void* hipStream_t;
typedef void* hipEvent_t;
// The runtime API the HOST side of the library uses (runtime.hip, decoder.hip, color.hip, ...), so that the product's own orchestration - launch
// sets, chains, the coalescers, pools - runs on the CPU as well (tests/emu/libheifhip_emu.so).  Everything is SYNCHRONOUS: a kernel has finished when
// its launch returns, so copies are memcpy, streams and events are tokens, synchronisation is a no-op; "device memory" is host memory.  That keeps
// every ordering the product asks for (it only ever waits for work it enqueued earlier) and hides none of its logic.  HIPEMU_DEVICES: emulated GPUs.
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorPeerAccessAlreadyEnabled = 704 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
namespace hipemu {
inline thread_local int current_device = 0;
inline int device_count() { const char* e = getenv("HIPEMU_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
// HIPEMU_POISON=<byte>: device and pinned allocations come back filled with that byte instead of whatever malloc hands out (mostly fresh zero pages):
// device memory is NOT zeroed by hipMalloc, so a kernel or a read-back that leans on zeros shows up as a mismatch against the oracle
namespace hipemu { inline int poison_byte() { static const int v = getenv("HIPEMU_POISON") ? (int)strtol(getenv("HIPEMU_POISON"), nullptr, 0) & 255 : -1; return v; } }
// HIPEMU_FAIL_ALLOC=<n>: every n-th device / pinned allocation of the process fails with hipErrorOutOfMemory - the error paths behind arena_acquire / pinned_acquire
// (memory-pressure hook, fall-backs, clean-up of half-built launch sets) under the sanitizers (tools/emu_api_fuzz.sh)
namespace hipemu { inline bool fail_this_alloc() { static const long n = getenv("HIPEMU_FAIL_ALLOC") ? atol(getenv("HIPEMU_FAIL_ALLOC")) : 0; static std::atomic<long> k{0}; return n > 0 && (k.fetch_add(1) + 1) % n == 0; } }
static inline hipError_t hipMalloc(void** p, size_t n) { if (hipemu::fail_this_alloc()) { *p = nullptr; return hipErrorOutOfMemory; } *p = malloc(n ? n : 1); if (*p && hipemu::poison_byte() >= 0) memset(*p, hipemu::poison_byte(), n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { if (hipemu::fail_this_alloc()) { *p = nullptr; return hipErrorOutOfMemory; } *p = malloc(n ? n : 1); if (*p && hipemu::poison_byte() >= 0) memset(*p, hipemu::poison_byte(), n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{
  for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
  return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = size_t(48) << 30; *total_b = size_t(64) << 30; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = hipemu::device_count(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= hipemu::device_count()) return hipErrorInvalidValue; hipemu::current_device = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = hipemu::current_device; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(8); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
using std::min;
using std::max;

#if defined(__SANITIZE_THREAD__)
#define HIPEMU_NOSAN __attribute__((no_sanitize_thread))
#else
#define HIPEMU_NOSAN
#endif
namespace hipemu {

constexpr size_t kStack = 256 << 10;

struct Lane {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
};

struct Coll {             // one convergence point of a wave (or of the whole group)
  int arrived = 0;
  uint64_t gen = 0;
  uint64_t ballot_acc = 0, part_acc = 0;
  uint32_t val_acc[64];
  uint64_t ballot_res[2], part_res[2];
  uint32_t val_res[2][64];
  void finalize()
  {
    const int s = (int)(gen & 1);
    ballot_res[s] = ballot_acc; part_res[s] = part_acc;
    memcpy(val_res[s], val_acc, sizeof(val_acc));
    ballot_acc = 0; part_acc = 0; arrived = 0;
    gen++;
  }
};

struct Group {
  dim3 bid, bdim, gdim;
  std::vector<Lane> lanes;
  std::vector<Coll> waves;
  std::vector<int> alive_in_wave;
  Coll bar;
  int alive = 0, cur = 0;
  ucontext_t sched;
  const std::function<void()>* body = nullptr;
};

inline thread_local Group* g = nullptr;

HIPEMU_NOSAN inline void yield_lane() { Group* G = g; swapcontext(&G->lanes[G->cur].ctx, &G->sched); }

// returns the result slot of the convergence point the calling lane took part in
HIPEMU_NOSAN inline int wave_converge(bool pred, uint32_t val)
{
  Group* G = g;
  const int t = G->cur, w = t >> 6, l = t & 63;
  Coll& C = G->waves[w];
  const uint64_t my = C.gen;
  if (pred) C.ballot_acc |= 1ull << l;
  C.part_acc |= 1ull << l;
  C.val_acc[l] = val;
  C.arrived++;
  if (C.arrived == G->alive_in_wave[w]) C.finalize();
  else while (C.gen == my) yield_lane();
  return (int)(my & 1);
}

HIPEMU_NOSAN inline void group_converge()
{
  Group* G = g;
  Coll& C = G->bar;
  const uint64_t my = C.gen;
  C.arrived++;
  if (C.arrived == G->alive) C.finalize();
  else while (C.gen == my) yield_lane();
}

HIPEMU_NOSAN inline void lane_entry()
{
  Group* G = g;
  (*G->body)();
  G = g;
  Lane& L = G->lanes[G->cur];
  L.done = true;
  const int w = G->cur >> 6;
  G->alive--; G->alive_in_wave[w]--;
  // the threads still waiting at a convergence point no longer wait for this one
  Coll& C = G->waves[w];
  if (C.arrived > 0 && C.arrived == G->alive_in_wave[w]) C.finalize();
  if (G->bar.arrived > 0 && G->bar.arrived == G->alive) G->bar.finalize();
  swapcontext(&L.ctx, &G->sched);
}

HIPEMU_NOSAN inline void run_group(Group& G)
{
  const int n = (int)(G.bdim.x * G.bdim.y * G.bdim.z);
  g = &G;
  G.alive = n;
  const int nw = (n + 63) / 64;
  G.waves.assign((size_t)nw, Coll());
  G.alive_in_wave.assign((size_t)nw, 0);
  G.bar = Coll();
  for (int i = 0; i < n; i++) {
    Lane& L = G.lanes[i];
    L.done = false;
    L.tid = dim3((unsigned)i % G.bdim.x, ((unsigned)i / G.bdim.x) % G.bdim.y, (unsigned)i / (G.bdim.x * G.bdim.y));
    G.alive_in_wave[i >> 6]++;
    getcontext(&L.ctx);
    L.ctx.uc_stack.ss_sp = L.stack; L.ctx.uc_stack.ss_size = kStack; L.ctx.uc_link = &G.sched;
    makecontext(&L.ctx, (void (*)())lane_entry, 0);
  }
  // HIPEMU_LANE_ORDER: 0 = lanes are resumed in index order (default), 1 = in reverse order, 2 = in a pseudo-random order that changes with every pass.
  // Between two convergence points the hardware promises no order among the lanes of a group (and none a compiler has to respect even inside a wave
  // without a fence): a kernel whose result depends on the order - a missing barrier between an LDS write and another lane's read - decodes
  // differently under 1 or 2 (tools/emu_random_sweep*.py under HIPEMU_LANE_ORDER).
  static const int lane_order = getenv("HIPEMU_LANE_ORDER") ? atoi(getenv("HIPEMU_LANE_ORDER")) : 0;
  uint32_t lcg = 12345u + G.bid.x * 2654435761u + G.bid.y * 40503u;
  while (G.alive > 0) {
    lcg = lcg * 1664525u + 1013904223u;
    const int start = lane_order == 2 ? (int)((lcg >> 8) % (uint32_t)n) : 0, stride = lane_order == 2 && n > 2 ? ((int)((lcg >> 20) % (uint32_t)(n - 1)) | 1) : 1;
    // (a stride coprime to n visits every lane; fall back to 1 when it is not)
    int st = stride;
    if (lane_order == 2) { int a = n, b = st; while (b) { const int t = a % b; a = b; b = t; } if (a != 1) st = 1; }
    for (int k = 0; k < n; k++) {
      const int i = lane_order == 1 ? n - 1 - k : (lane_order == 2 ? (int)(((long)start + (long)k * st) % n) : k);
      if (G.lanes[i].done) continue;
      G.cur = i;
      swapcontext(&G.sched, &G.lanes[i].ctx);
    }
  }
  g = nullptr;
}

inline int pool_threads()
{
  const char* e = getenv("HIPEMU_THREADS");
  return e ? atoi(e) : 96;
}

template <typename F>
HIPEMU_NOSAN void launch(dim3 grid, dim3 block, F&& fn)
{
  const std::function<void()> body = fn;
  const size_t total = (size_t)grid.x * grid.y * grid.z;
  const int n = (int)(block.x * block.y * block.z);
  std::atomic<size_t> next{0};
  auto worker = [&]() HIPEMU_NOSAN {
    Group G;
    G.gdim = grid; G.bdim = block; G.body = &body;
    G.lanes.resize((size_t)n);
    char* stacks = (char*)mmap(nullptr, kStack * (size_t)n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char*)MAP_FAILED) { fprintf(stderr, "hipemu: mmap failed\n"); abort(); }
    for (int i = 0; i < n; i++) G.lanes[(size_t)i].stack = stacks + kStack * (size_t)i;
    for (;;) {
      const size_t w = next.fetch_add(1);
      if (w >= total) break;
      G.bid = dim3((unsigned)(w % grid.x), (unsigned)((w / grid.x) % grid.y), (unsigned)(w / ((size_t)grid.x * grid.y)));
      run_group(G);
    }
    munmap(stacks, kStack * (size_t)n);
  };
  const int nt = (int)std::min<size_t>(total, (size_t)std::max(1, pool_threads()));
  if (nt <= 1) { worker(); return; }   // (HIPEMU_THREADS=1: the groups run on the launching thread - what the ThreadSanitizer host wants, tools/emu_tsan_host.sh)
  std::vector<std::thread> th;
  for (int i = 0; i < nt; i++) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

}  // namespace hipemu

// ---- language surface used by libheif_amd/csrc/*.hip -----------------------------------------------------------------
#if defined(__SANITIZE_THREAD__)
// ThreadSanitizer builds (tools/emu_tsan_objects.sh, ALL=1): device code runs on ucontext lanes the sanitizer cannot follow - kernels and device
// functions stay uninstrumented, the host functions of the same translation unit (launchers, planner, capture state) are watched
#define __global__ __attribute__((no_sanitize_thread))
#define __device__ __attribute__((no_sanitize_thread))
#else
#define __global__
#define __device__
#endif
#define __host__
#define __forceinline__ inline
#define __constant__ static const
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define threadIdx (hipemu::g->lanes[hipemu::g->cur].tid)
#define blockIdx (hipemu::g->bid)
#define blockDim (hipemu::g->bdim)
#define gridDim (hipemu::g->gdim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  do { (void)(stream); hipemu::launch((grid), (block), [&]() HIPEMU_NOSAN { kernel(__VA_ARGS__); }); } while (0)

#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_ACQ_REL)
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_sleep(n) do { hipemu::yield_lane(); sched_yield(); } while (0)
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::wave_converge(false, 0))
#define __syncthreads() hipemu::group_converge()

template <typename T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline int atomicCAS(int* p, int cmp, int v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE); return cmp; }
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE); return cmp; }

HIPEMU_NOSAN static inline uint64_t __ballot(int pred)
{
  const int s = hipemu::wave_converge(pred != 0, 0);
  return hipemu::g->waves[hipemu::g->cur >> 6].ballot_res[s];
}
HIPEMU_NOSAN static inline int __shfl_xor(int v, int mask)
{
  const int s = hipemu::wave_converge(false, (uint32_t)v);
  return (int)hipemu::g->waves[hipemu::g->cur >> 6].val_res[s][((hipemu::g->cur & 63) ^ mask) & 63];
}
HIPEMU_NOSAN static inline int __shfl(int v, int src)
{
  const int s = hipemu::wave_converge(false, (uint32_t)v);
  return (int)hipemu::g->waves[hipemu::g->cur >> 6].val_res[s][src & 63];
}
HIPEMU_NOSAN static inline int hipemu_readfirstlane(int v)
{
  const int s = hipemu::wave_converge(false, (uint32_t)v);
  const hipemu::Coll& C = hipemu::g->waves[hipemu::g->cur >> 6];
  return (int)C.val_res[s][__builtin_ctzll(C.part_res[s])];
}
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
static inline int __mul24(int a, int b) { return a * b; }
static inline unsigned __umul24(unsigned a, unsigned b) { return a * b; }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
