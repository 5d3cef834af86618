// decoder.hip — host orchestration of the decode pipeline and the hipdec_decoder_* / hipdec_batch_*
// entry points of the C ABI (include/heif_hipdec.h).
//
// hipdec_decoder mirrors the life cycle libheif drives through heif_decoder_plugin
// (libheif/codecs/decoder.cc:355-563; reference implementation libheif/plugins/decoder_libde265.cc);
// hipdec_batch is the device-side form of libheif's per-tile fan-out
// (libheif/image-items/grid.cc:405-453): all items of a grid / batch are parsed on the host, uploaded
// with ONE copy and decoded by ONE set of kernel launches, every CABAC substream and every CTB row
// of every item being an independent unit of GPU work.
#include "hipdec_internal.h"
#include "hevc_headers.h"
#include "kernels.h"
#include "batch_layout.h"
#include "color_device.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <unordered_map>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>


using namespace hipdec;

constexpr int kEv = 8;   // events per timing slot: start, parse, residual, recon, deblock, sao | colour begin, colour end

// resident-RGB policy: conversions that could have been served from a launch set's RGB earn credit, RGB that nobody fetched spends it
namespace { std::atomic<int> g_rgb_credit{0}; }
static void rgb_note_unused(int n) { if (n > 0) { int c = g_rgb_credit.load(); while (c > 0 && !g_rgb_credit.compare_exchange_weak(c, std::max(0, c - n))) {} } }
static void rgb_note_wanted() { int c = g_rgb_credit.load(); while (c < 4096 && !g_rgb_credit.compare_exchange_weak(c, std::min(4096, c + 64))) {} }
static bool rgb_wanted() { static const bool off = getenv("HIPDEC_NO_RESIDENT_RGB") != nullptr; return !off && g_rgb_credit.load(std::memory_order_relaxed) > 0; }

struct hipdec_batch : BatchLayout {
  int device = 0;               // the device the arena lives on: every entry point that touches the batch runs under its scope
  uint8_t* arena = nullptr;
  size_t arena_capacity = 0;
  void* staging = nullptr;      // pinned upload staging (large batches), returned to its pool once the upload has completed
  size_t staging_capacity = 0;
  void* upload_dev = nullptr;   // a batch that takes over a busy arena: its upload region lands here first (beside the predecessor's kernels), build_batch
  size_t upload_dev_capacity = 0;
  hipEvent_t uploaded = nullptr;   // recorded on the upload stream behind the H2D copy; launch streams wait on it
  hipEvent_t done = nullptr;       // recorded behind the last piece of work enqueued for this batch (decode, colour stage, packs)
  bool done_recorded = false;
  int32_t* host_status = nullptr;  // pinned: the device status word of the last run, copied behind its kernels (the arena may already
                                   // belong to the next batch when hipdec_batch_status() looks)
  std::vector<hipEvent_t> ev;   // kEv events per timing slot; run k records into slot k % slots
  std::vector<hipEvent_t> chain_events;   // chain batches: one per motion step (the motion stream's progress, awaited by the pixel steps)
  std::vector<uint8_t> colour_timed;   // per slot: the colour stage of that run was recorded
  uint64_t runs = 0;
  hipStream_t last_stream = nullptr;
  bool ran = false;
  bool retired = false;         // its arena went to another batch (hipdec_batch_create_recycling): only status / timing / free remain
  uint32_t wave_share = 1;      // the CABAC work pool of this batch takes 1 / wave_share of the wave budget (launch sets of the decoder path overlap in pairs)
  ColorBatchState color;        // parameter blocks of hipdec_batch_to_rgb_all
  // decoder path (plugin): the output planes of every item staged in pinned host memory by ONE set of asynchronous copies behind the
  // kernels, so that N decoder instances sharing the batch do not queue N x 3 pageable device-to-host copies (stage_planes_to_host)
  struct HostItem { void* p = nullptr; size_t off[3] = {0, 0, 0}; const uint8_t* rgb = nullptr; };   // p points into one of host_chunks; rgb: tight RGB24 rows (resident RGB)
  std::vector<HostItem> host_items;
  // pinned chunks of ONE size (64 MiB; a larger item gets a chunk of its own), items sub-allocated at 256-B alignment: a launch set of 256
  // thumbnails pins one chunk instead of 256 x 16 MiB, and sets of any size recycle the same chunks from the pool
  std::vector<std::pair<void*, size_t>> host_chunks;
  // Resident RGB (drop-in through a libheif with the integration ops): when the host has been asking for interleaved RGB24 of
  // the images it decodes, the launch set runs the SAO kernel with the fused RGB24 emission (k_sao_rgb) into rgb_dev and stages the rows to pinned
  // host memory beside the planes: the colour conversion of libheif's pipeline then is a host copy instead of a kernel queued behind CABAC pools
  void* rgb_dev = nullptr;
  size_t rgb_capacity = 0;
  std::vector<size_t> rgb_off, rgb_stride;
  std::vector<std::pair<void*, size_t>> rgb_chunks;   // pinned
  std::atomic<int> rgb_consumed{0};
  bool fused_rgb = false;                 // the last hipdec_batch_run_rgb took the fused form (k_sao_rgb)
  // Pipelined chains (decoder_chains.inc): a chain launch set that is enqueued while earlier chains of the same tracks are still running.  Its CABAC
  // and residual launches need nothing of them and start at once; its motion derivation waits for their motion fields, its pixel steps for their
  // pictures.  The set keeps its streams until somebody has looked at its status (a stream handed back early would carry the NEXT set's CABAC
  // launch behind this set's pixel steps), and that look happens once, by whichever decoder instance needs a picture of the set first.
  std::vector<std::shared_ptr<hipdec_batch>> after;
  bool hold_streams = false;
  std::vector<hipStream_t> held_streams;
  bool motion_recorded = false;           // chain_events[0] stands behind the motion fields of ALL pictures
  std::mutex fin_mu;
  bool finished = false;
  int fin_rc = 0;
  std::string fin_err;
  // Waits for everything enqueued for THIS batch — not for the stream, which may already carry the next batch.
  hipError_t wait() const
  {
    hipError_t e = hipSuccess;
    if (uploaded) e = hipEventSynchronize(uploaded);
    if (e == hipSuccess && done_recorded) e = hipEventSynchronize(done);
    return e;
  }
  void mark_done(hipStream_t s)
  {
    if (done && hipEventRecord(done, s) == hipSuccess) done_recorded = true;
  }
  void release_staging()
  {
    if (!staging) return;
    if (uploaded) (void)hipEventSynchronize(uploaded);
    pinned_release(staging, staging_capacity);
    staging = nullptr;
    if (upload_dev) { arena_release(upload_dev, upload_dev_capacity); upload_dev = nullptr; }
  }
  ~hipdec_batch()
  {
    DeviceScope scope(device);
    if (arena || staging) (void)wait();   // nothing of this batch may still be running when the arena is recycled
    release_staging();
    for (hipStream_t hs : held_streams) stream_release(hs);
    for (auto& c : host_chunks) pinned_release(c.first, c.second);
    for (auto& c : rgb_chunks) pinned_release(c.first, c.second);
    if (rgb_dev) { arena_release(rgb_dev, rgb_capacity); rgb_note_unused((int)rgb_off.size() - rgb_consumed.load()); }
    color_batch_state_free(color);
    if (arena) arena_release(arena, arena_capacity);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : chain_events) if (e) (void)hipEventDestroy(e);
    if (uploaded) (void)hipEventDestroy(uploaded);
    if (done) (void)hipEventDestroy(done);
    status_slot_release(host_status);
  }
};

namespace {

// The host side of "compressed bytes in host memory -> decode" (SURVEY.md §8d): header parsing (worker threads for large
// batches), staging and the upload.  Large upload regions go through pinned staging and ONE asynchronous copy on the
// library's upload stream, so that hipdec_batch_create() of batch k+1 overlaps the kernels of batch k; small ones (a still, the
// tiles of a grid photo) are copied synchronously from pageable memory, which is quicker than pinning.
// chains: the n items are consecutive samples of sequence tracks (BatchLayout::chain; items [first[t], first[t] + count[t]) belong to track t whose
// state is *seqs[t]): the batch may come out EMPTY (every sample was a RASL picture 8.3.3 drops) - then nothing is allocated
struct ChainPlan { int n_tracks; const int* first; const int* count; const SeqContext* const* seqs; int* bad_track; };
int build_batch(hipdec_batch& b, int n, const void* const* data, const size_t* sizes, uint64_t max_pixels, hipdec_batch* recycle = nullptr,
                const SeqContext* const* seqs = nullptr, const ChainPlan* chains = nullptr)
{
  std::string err;
  b.device = active_device();
  int rc = chains ? layout_batch_plan_chains(b, chains->n_tracks, chains->first, chains->count, data, sizes, max_pixels, err, chains->seqs, chains->bad_track)
                  : layout_batch_plan(b, n, data, sizes, max_pixels, err, seqs);
  if (rc != HIPDEC_OK) return set_error(rc, "%s", err.c_str());
  if (b.pics.empty()) return 0;
  HIPDEC_CHECK_HIP(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
  b.host_status = status_slot_acquire();
  if (!b.host_status) return set_error(HIPDEC_ERR_MEMORY, "batch_create: more than 4096 live batches (no pinned status slot left)");
  *b.host_status = 0;
  // a retired batch of the same shape hands its arena over (hipdec_batch_create_recycling): the upload is ordered behind
  // everything that batch still has in flight, so one arena serves a stream of batches
  hipEvent_t after = nullptr;
  if (recycle && recycle->arena && recycle->arena_capacity >= b.arena_size && recycle->device == b.device) {
    b.arena = recycle->arena; b.arena_capacity = recycle->arena_capacity;
    recycle->arena = nullptr; recycle->arena_capacity = 0;
    if (recycle->done_recorded) after = recycle->done;
    recycle->retired = true;
  }
  // if anything below fails, the arena goes back to the pool with this batch: not before the retired batch's kernels have left it
  struct TakeoverGuard {
    hipEvent_t after; bool ok;
    ~TakeoverGuard() { if (!ok && after) (void)hipEventSynchronize(after); }
  } takeover{after, false};
  static const bool sync_upload = getenv("HIPDEC_SYNC_UPLOAD") != nullptr;   // profiling knob: one stream, no cross-stream event waits
  // (a chain that is enqueued beside chains in flight never takes the synchronous copy: hipMemcpy from pageable memory waits for the device - measured,
  //  80 ms per chain of 16 720p pictures, i.e. for everything in flight)
  if ((b.upload_size > (size_t(4) << 20) || b.hold_streams) && !sync_upload) {
    HIPDEC_CHECK_HIP(pinned_acquire(&b.staging, b.upload_size, &b.staging_capacity));
    if (!b.arena) HIPDEC_CHECK_HIP(arena_acquire((void**)&b.arena, b.arena_size, &b.arena_capacity));   // (first: a chain's reference tables hold addresses inside it)
    layout_batch_fill(b, data, sizes, (uint8_t*)b.staging, (uint64_t)(uintptr_t)b.arena);
    HIPDEC_CHECK_HIP(hipEventCreateWithFlags(&b.uploaded, hipEventDisableTiming));
    hipStream_t us = upload_stream();
    // An arena taken over from a batch whose kernels are still running cannot receive the upload yet: the bytes cross PCIe NOW into a bounce
    // buffer in HBM (beside those kernels) and move into the arena with a device copy once the predecessor has finished - 2 GB per 2048 4K
    // stills are ~ 40 ms of link time against ~ 1 ms of HBM copy between two batches.  No memory for the bounce buffer: the upload waits.
    static const bool no_bounce = getenv("HIPDEC_NO_UPLOAD_BOUNCE") != nullptr;   // A/B knob
    if (after && !no_bounce && arena_acquire(&b.upload_dev, b.upload_size, &b.upload_dev_capacity) != hipSuccess) { b.upload_dev = nullptr; (void)hipGetLastError(); }
    if (after && b.upload_dev) {
      HIPDEC_CHECK_HIP(hipMemcpyAsync(b.upload_dev, b.staging, b.upload_size, hipMemcpyHostToDevice, us));
      HIPDEC_CHECK_HIP(hipStreamWaitEvent(us, after, 0));
      HIPDEC_CHECK_HIP(hipMemcpyAsync(b.arena, b.upload_dev, b.upload_size, hipMemcpyDeviceToDevice, us));
    } else {
      if (after) HIPDEC_CHECK_HIP(hipStreamWaitEvent(us, after, 0));
      HIPDEC_CHECK_HIP(hipMemcpyAsync(b.arena, b.staging, b.upload_size, hipMemcpyHostToDevice, us));
    }
    HIPDEC_CHECK_HIP(hipEventRecord(b.uploaded, us));
  } else {
    std::vector<uint8_t> host(b.upload_size);
    if (!b.arena) HIPDEC_CHECK_HIP(arena_acquire((void**)&b.arena, b.arena_size, &b.arena_capacity));
    layout_batch_fill(b, data, sizes, host.data(), (uint64_t)(uintptr_t)b.arena);
    if (after) HIPDEC_CHECK_HIP(hipEventSynchronize(after));
    HIPDEC_CHECK_HIP(hipMemcpy(b.arena, host.data(), b.upload_size, hipMemcpyHostToDevice));
  }
  b.ev.assign(kEv, nullptr);
  b.colour_timed.assign(1, 0);
  for (auto& e : b.ev) HIPDEC_CHECK_HIP(hipEventCreate(&e));
  takeover.ok = true;
  return 0;
}

// The stream follow-up work of a batch (colour stage, packs, pastes) goes on: the caller's, made to wait for the batch's last recorded
// work when that was enqueued elsewhere — with hipdec_set_stage_overlap(1) the pixel stages of a run leave the caller's stream for the
// post stream, so a copy queued on the caller's stream right behind hipdec_batch_run() would otherwise read unfinished planes
//  — or the batch's own last stream.
hipStream_t follow_stream(hipdec_batch* b, void* stream)
{
  if (!stream) return b->last_stream ? b->last_stream : default_stream();
  hipStream_t s = (hipStream_t)stream;
  if (s != b->last_stream && b->done_recorded) (void)hipStreamWaitEvent(s, b->done, 0);
  return s;
}

}  // namespace
namespace hipdec { hipStream_t batch_follow_stream(hipdec_batch* b, hipStream_t s) { return follow_stream(b, (void*)s); } }
namespace {

int launch_all(hipdec_batch& b, hipStream_t s, const void* fused_rgb_params = nullptr)
{
  const int n = (int)b.params.size();
  ParseArgs pa{};
  pa.pics = (const PicParams*)(b.arena + b.off_pics); pa.subs = (const Substream*)(b.arena + b.off_subs);
  pa.waves = (const ParseWave*)(b.arena + b.off_waves); pa.num_waves = b.num_waves; pa.arena = b.arena;
  pa.progress = (uint32_t*)(b.arena + b.off_progress); pa.ctx_store = b.arena + b.off_ctx;
  pa.ticket = (uint32_t*)(b.arena + b.off_ticket); pa.status = (int32_t*)(b.arena + b.off_status);
  if (b.pool) {
    // pool size: every pool wave must be resident, so the batches in flight share the wave slots (runtime.hip); measured on
    // MI355X, 1024 4K stills: 7168 waves for one batch, 2 x 3584 for two overlapping ones (2 x 4096 oversubscribes and
    // loses 20 %)
    uint32_t waves = getenv("HIPDEC_POOL_WAVES") ? (uint32_t)atoi(getenv("HIPDEC_POOL_WAVES")) : parse_wave_budget() / (b.wave_share ? b.wave_share : 1u);
    waves = waves > b.num_subs ? b.num_subs : waves;
    pa.num_waves = waves < 1 ? 1 : waves;
  }
  // a row hands its wave back after every CTB: the ready queue then advances all rows breadth-first and rows rarely run into
  // the row above (measured: parse 698 -> 591 ms against "run until blocked")
  pa.yield_ctbs = getenv("HIPDEC_POOL_YIELD") ? (uint32_t)atoi(getenv("HIPDEC_POOL_YIELD")) : 1;
  pa.wake_hyst = getenv("HIPDEC_POOL_HYST") ? (uint32_t)atoi(getenv("HIPDEC_POOL_HYST")) : 0u;
  pa.general_chroma = 0;
  for (const PicParams& P : b.params) if (P.chroma_format_idc >= 2) pa.general_chroma = 1;
  pa.inter = b.any_inter ? 1 : 0;
  if (pa.inter && pa.general_chroma) return set_error(HIPDEC_ERR_UNSUPPORTED, "a batch that mixes P pictures with 4:2:2 / 4:4:4 pictures");
  pa.pool = b.pool; pa.queue_cap = b.queue_cap; pa.num_subs = b.num_subs;
  pa.waitneed = (uint32_t*)(b.arena + b.off_waitneed); pa.resume_k = (uint32_t*)(b.arena + b.off_resume_k);
  pa.queue = (uint32_t*)(b.arena + b.off_queue); pa.qctl = (uint32_t*)(b.arena + b.off_qctl); pa.saved = (uint32_t*)(b.arena + b.off_saved);
  ReconArgs ra{(const PicParams*)(b.arena + b.off_pics), (const ReconWave*)(b.arena + b.off_rwaves), b.num_rwaves, b.arena,
               (uint32_t*)(b.arena + b.off_row_progress), (uint32_t*)(b.arena + b.off_ticket) + 1, (int32_t*)(b.arena + b.off_status)};
  FilterArgs fa{(const PicParams*)(b.arena + b.off_pics), b.arena, (const int32_t*)(b.arena + b.off_status)};
  const bool dbg = getenv("HIPDEC_DEBUG_SYNC") != nullptr;  // isolate a faulting kernel
  auto step = [&](const char* what) -> int {
    if (!dbg) return 0;
    fprintf(stderr, "[hipdec] %s ...\n", what); fflush(stderr);
    hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, "[hipdec] %s: %s\n", what, hipGetErrorString(e)); fflush(stderr);
    return e == hipSuccess ? 0 : set_error(HIPDEC_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
  };
  const size_t slot = b.runs % (b.ev.size() / kEv);
  hipEvent_t* ev = b.ev.data() + kEv * slot;
  b.colour_timed[slot] = 0;
  b.runs++;
  if (b.uploaded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(s, b.uploaded, 0));
  // Two-stage pipeline across batches (hipdec_set_stage_overlap): the CABAC kernel runs on the caller's stream, everything behind it on the
  // device's post stream.  With two batches alternating, batch k+1's parse — scalar / vector issue bound, its dependency tail leaves the
  // chip half empty — overlaps batch k's reconstruction, filters and colour stage (VALU + HBM).  A batch's own previous run must have
  // left the arena before the control words are zeroed again.
  const bool split = stage_overlap();
  hipStream_t ps = split ? post_stream() : s;
  if (split && b.done_recorded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(s, b.done, 0));
  HIPDEC_CHECK_HIP(hipMemsetAsync(b.arena + b.off_ctrl, 0, b.ctrl_size, s));
  HIPDEC_CHECK_HIP(hipEventRecord(ev[0], s));
  if (int rc = step("memset")) return rc;
#ifdef HIPDEC_POOL_TRACE   // measurement build: per pool wave {start, end of its last row, last look at the queue, time waiting for work, rows run, time in rows}
  {                        // in 100 MHz ticks; the previous run's table is appended to $HIPDEC_POOL_TRACE (binary, 32 x uint64 per wave - [8 + k]: time waiting for work in the k-th 42 ms of its life -, 8192 waves per run)
    static unsigned long long* trace = nullptr;
    const size_t bytes = 8192 * 32 * sizeof(unsigned long long);
    if (!trace) { HIPDEC_CHECK_HIP(hipMalloc((void**)&trace, bytes)); HIPDEC_CHECK_HIP(hipMemset(trace, 0, bytes)); }
    else if (const char* path = getenv("HIPDEC_POOL_TRACE")) {
      HIPDEC_CHECK_HIP(hipDeviceSynchronize());
      std::vector<unsigned long long> h(8192 * 32);
      HIPDEC_CHECK_HIP(hipMemcpy(h.data(), trace, bytes, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "ab")) { fwrite(h.data(), 1, bytes, f); fclose(f); }
      HIPDEC_CHECK_HIP(hipMemset(trace, 0, bytes));
    }
    pa.trace = trace;
  }
#endif
  launch_parse(pa, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[1], s));
  if (int rc = step("parse")) return rc;
  if (split) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, ev[1], 0));
  static const bool parse_only = getenv("HIPDEC_DEBUG_PARSE_ONLY") != nullptr;   // tuning knob: isolate the CABAC kernel
  if (!parse_only) launch_residual(fa, n, b.max_ctbs, pa.general_chroma != 0, ps);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[2], ps));
  if (int rc = step("residual")) return rc;
  if (!parse_only && b.chain) {
    // A chain (consecutive samples of one track, parsed and inverse-transformed together above).  Motion derivation needs the parser's output and
    // the collocated picture's motion field only, so the motion steps run on a stream of their own behind the parser, beside the pixel steps of
    // earlier pictures; a pixel step (prediction, reconstruction, filters of pictures that do not predict from each other) waits for the motion
    // fields of its pictures, and stream order makes every reference picture - earlier items of this batch among them - complete.
    static const bool motion_by_steps = getenv("HIPDEC_CHAIN_MOTION_STEPS") != nullptr;   // A/B knob: one k_motion launch per motion step (the form before the
                                                                                          // kernel waited for its collocated picture's rows itself)
    if (b.any_inter && !motion_by_steps) {
      // the motion fields of ALL pictures with one launch on a stream of its own (kernels.h: launch_chain_motion_all), the pixel steps behind it
      hipStream_t ms = stream_acquire();
      if (b.chain_events.empty()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { stream_release(ms); return set_error(HIPDEC_ERR_DEVICE, "chain: no event"); }
        b.chain_events.push_back(e);
      }
      HIPDEC_CHECK_HIP(hipStreamWaitEvent(ms, ev[1], 0));
      for (const auto& pred : b.after) {   // collocated pictures of chains still in flight: their motion fields
        if (pred->motion_recorded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ms, pred->chain_events[0], 0));
        else if (pred->done_recorded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ms, pred->done, 0));
      }
      launch_chain_motion_all(b, b.arena, ms);
      HIPDEC_CHECK_HIP(hipEventRecord(b.chain_events[0], ms));
      b.motion_recorded = true;
      if (b.hold_streams) b.held_streams.push_back(ms); else stream_release(ms);
      for (const auto& pred : b.after)     // reference pictures of chains still in flight
        if (pred->done_recorded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, pred->done, 0));
      bool waited = false;
      for (size_t k = 0; k < b.pixel_steps.size(); k++) {
        if (b.pixel_steps[k].any_inter && !waited) { HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, b.chain_events[0], 0)); waited = true; }
        launch_chain_pixels(b, b.arena, (int)k, ps);
      }
      if (!waited) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, b.chain_events[0], 0));   // nothing of this batch outlives its `done` event
      HIPDEC_CHECK_HIP(hipEventRecord(ev[3], ps));
      HIPDEC_CHECK_HIP(hipEventRecord(ev[4], ps));
      HIPDEC_CHECK_HIP(hipEventRecord(ev[5], ps));
      if (int rc = step("chain pixel stages")) return rc;
      HIPDEC_CHECK_HIP(hipGetLastError());
      // (a set left in flight queues NO copy behind its kernels: a DMA copy that waits for the pixel steps holds its copy engine's queue, and the upload
      //  of the next set - whose CABAC launch is to run beside those pixel steps - sits behind it; batch_finish copies once the kernels are done)
      if (!b.hold_streams) HIPDEC_CHECK_HIP(hipMemcpyAsync(b.host_status, b.arena + b.off_status, sizeof(int32_t), hipMemcpyDeviceToHost, ps));
      b.last_stream = ps;
      b.mark_done(ps);
      return 0;
    }
    for (const auto& pred : b.after)       // (the forms below keep the simple order: everything of this chain's later stages behind the chains in flight)
      if (pred->done_recorded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, pred->done, 0));
    hipStream_t ms = nullptr;
    if (b.any_inter && b.motion_steps.size() > 1) {
      ms = stream_acquire();
      while (b.chain_events.size() < b.motion_steps.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { stream_release(ms); return set_error(HIPDEC_ERR_DEVICE, "chain: no event"); }
        b.chain_events.push_back(e);
      }
      HIPDEC_CHECK_HIP(hipStreamWaitEvent(ms, ev[1], 0));
      for (const auto& pred : b.after) if (pred->done_recorded) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ms, pred->done, 0));
      for (size_t k = 0; k < b.motion_steps.size(); k++) {
        launch_chain_motion(b, b.arena, (int)k, ms);
        HIPDEC_CHECK_HIP(hipEventRecord(b.chain_events[k], ms));
      }
      stream_release(ms);   // (everything it carries is ordered into `ps` through the events below)
    }
    int motion_done = 0;
    for (size_t k = 0; k < b.pixel_steps.size(); k++) {
      const BatchLayout::ChainStep& st = b.pixel_steps[k];
      const int need = st.motion_need;
      if (ms) { if (st.any_inter) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, b.chain_events[(size_t)need - 1], 0)); }
      else for (; motion_done < need; motion_done++) launch_chain_motion(b, b.arena, motion_done, ps);
      launch_chain_pixels(b, b.arena, (int)k, ps);
    }
    if (ms) HIPDEC_CHECK_HIP(hipStreamWaitEvent(ps, b.chain_events[b.motion_steps.size() - 1], 0));   // nothing of this batch outlives its `done` event
    HIPDEC_CHECK_HIP(hipEventRecord(ev[3], ps));
    HIPDEC_CHECK_HIP(hipEventRecord(ev[4], ps));
    HIPDEC_CHECK_HIP(hipEventRecord(ev[5], ps));
    if (int rc = step("chain pixel stages")) return rc;
    HIPDEC_CHECK_HIP(hipGetLastError());
    if (!b.hold_streams) HIPDEC_CHECK_HIP(hipMemcpyAsync(b.host_status, b.arena + b.off_status, sizeof(int32_t), hipMemcpyDeviceToHost, ps));
    b.last_stream = ps;
    b.mark_done(ps);
    return 0;
  }
  if (!parse_only && b.any_inter) {
    // P pictures: motion vectors (merge / AMVP candidates: a 2-CTB wavefront over CTB rows), then the motion-compensated prediction of every
    // inter coded sample into the reconstruction planes; k_recon adds the residuals and predicts the intra blocks around them
    MotionArgs ma{(const PicParams*)(b.arena + b.off_pics), (const RowDesc*)(b.arena + b.off_rows), b.num_rows, b.arena,
                  (uint32_t*)(b.arena + b.off_row_progress), (uint32_t*)(b.arena + b.off_ticket) + 2, (int32_t*)(b.arena + b.off_status)};
    launch_motion(ma, ps);
    launch_mc(fa, n, b.max_w, b.max_h, b.wide, ps, inter_residual_in_mc());
    ra.inter_from_plane = inter_residual_in_mc() ? 1u : 0u;
    if (int rc = step("motion + mc")) return rc;
  }
  if (!parse_only) launch_recon(ra, b.wide, ps, b.any_inter);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[3], ps));
  if (int rc = step("recon")) return rc;
  if (!parse_only) launch_deblock(fa, n, b.max_w, b.max_h, b.wide, ps, !b.any_inter && !pa.general_chroma);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[4], ps));
  if (int rc = step("deblock")) return rc;
  if (!parse_only) {
    bool may_keep = false, restricted = false;
    for (const PicParams& P : b.params) {
      if (P.transquant_bypass_enabled || (P.pcm_enabled && P.pcm_loop_filter_disabled)) may_keep = true;
      if (!P.sao_free_neighbours) restricted = true;
    }
    if (fused_rgb_params)   // SAO + crop + RGB24 in one pass
      launch_sao_rgb(fa, fused_rgb_params, n, b.max_ow, b.max_oh, ps, may_keep, restricted, b.params.data(),
                     b.color.host.size() == (size_t)n * sizeof(colordev::ColorParams) ? b.color.host.data() : nullptr);
    else launch_sao(fa, n, b.max_ow, b.max_oh, b.wide, ps, may_keep, restricted);
  }
  HIPDEC_CHECK_HIP(hipEventRecord(ev[5], ps));
  if (int rc = step("sao")) return rc;
  HIPDEC_CHECK_HIP(hipGetLastError());
  HIPDEC_CHECK_HIP(hipMemcpyAsync(b.host_status, b.arena + b.off_status, sizeof(int32_t), hipMemcpyDeviceToHost, ps));
  b.last_stream = ps;     // the colour stage, packs and plane reads of this run follow its last kernel
  b.mark_done(ps);
  return 0;
}

const char* dev_err_name(int code)
{
  switch (code & 0xff) {
    case DEV_ERR_TERMINATE: return "CABAC substream did not terminate where the slice header says (desynchronised bitstream)";
    case DEV_ERR_BITSTREAM_END: return "CABAC read past the end of a substream";
    case DEV_ERR_SYNTAX: return "syntax element out of range";
    case DEV_ERR_TIMEOUT: return "dependency wait timed out / aborted";
    default: return "unknown device error";
  }
}

// The decoder path's hand-over: every item's cropped planes, tight rows, into a pinned buffer of its own (pooled), queued on the
// launch set's stream behind its kernels.  hipdec_decoder_read_plane() then is a host memcpy into libheif's plane — in parallel on the
// application's threads — instead of one synchronous pageable device-to-host copy per plane and instance.
int stage_planes_to_host(hipdec_batch& b, hipStream_t s)
{
  static const bool off = getenv("HIPDEC_NO_HOST_STAGING") != nullptr;
  if (off) return 0;
  const size_t es = b.wide ? 2 : 1;
  constexpr size_t kChunk = size_t(64) << 20;
  std::vector<hipdec_batch::HostItem> items(b.params.size());
  std::vector<size_t> need;                       // bytes used of chunk k
  std::vector<std::pair<int, size_t>> place(b.params.size(), {-1, 0});   // item -> (chunk, offset)
  for (size_t i = 0; i < b.params.size(); i++) {
    const PicParams& P = b.params[i];
    size_t total = 0;
    for (int c = 0; c < (P.chroma_format_idc ? 3 : 1); c++) {
      items[i].off[c] = total;
      total += (size_t)(c ? P.out_cwidth : P.out_width) * es * (size_t)(c ? P.out_cheight : P.out_height);
      total = (total + 255) & ~size_t(255);
    }
    if (!total) continue;
    if (need.empty() || need.back() + total > std::max(kChunk, need.back() ? kChunk : total)) need.push_back(0);
    place[i] = {(int)need.size() - 1, need.back()};
    need.back() += total;
  }
  if (need.empty()) return 0;
  // chunks this batch already holds are reused (a second run over the same arena); missing ones come from the pool
  bool ok = true;
  for (size_t k = 0; k < need.size() && ok; k++) {
    const size_t want = std::max(kChunk, need[k]);
    if (k < b.host_chunks.size() && b.host_chunks[k].second >= want) continue;
    if (k < b.host_chunks.size()) { pinned_release(b.host_chunks[k].first, b.host_chunks[k].second); b.host_chunks[k] = {nullptr, 0}; }
    else b.host_chunks.emplace_back(nullptr, 0);
    if (pinned_acquire(&b.host_chunks[k].first, want, &b.host_chunks[k].second) != hipSuccess) { (void)hipGetLastError(); b.host_chunks[k] = {nullptr, 0}; ok = false; }
  }
  if (!ok) {
    // no pinned memory even after the pool was emptied (pinned_acquire retries once): the instances read their planes straight from the
    // device instead (hipdec_batch_read_plane's unstaged path) - slower, but not an error
    for (auto& c : b.host_chunks) pinned_release(c.first, c.second);
    b.host_chunks.clear();
    b.host_items.clear();
    return 0;
  }
  for (size_t i = 0; i < b.params.size(); i++) {
    if (place[i].first < 0) continue;
    const PicParams& P = b.params[i];
    hipdec_batch::HostItem& h = items[i];
    h.p = (uint8_t*)b.host_chunks[(size_t)place[i].first].first + place[i].second;
    for (int c = 0; c < (P.chroma_format_idc ? 3 : 1); c++) {
      const size_t w = (size_t)(c ? P.out_cwidth : P.out_width) * es, hh = (size_t)(c ? P.out_cheight : P.out_height);
      if (!w || !hh) continue;
      if (P.out_stride[c] == w) HIPDEC_CHECK_HIP(hipMemcpyAsync((uint8_t*)h.p + h.off[c], b.arena + P.off_out[c], w * hh, hipMemcpyDeviceToHost, s));   // (the usual case: one DMA)
      else HIPDEC_CHECK_HIP(hipMemcpy2DAsync((uint8_t*)h.p + h.off[c], w, b.arena + P.off_out[c], P.out_stride[c], w, hh, hipMemcpyDeviceToHost, s));
    }
  }
  // resident RGB: the fused colour stage's rows, tight, one pinned chunk set of their own
  if (b.rgb_dev && b.rgb_off.size() == b.params.size()) {
    std::vector<size_t> rneed;
    std::vector<std::pair<int, size_t>> rplace(b.params.size(), {-1, 0});
    for (size_t i = 0; i < b.params.size(); i++) {
      const size_t total = ((size_t)b.params[i].out_width * 3 * (size_t)b.params[i].out_height + 255) & ~size_t(255);
      if (!total) continue;
      if (rneed.empty() || rneed.back() + total > std::max(kChunk, rneed.back() ? kChunk : total)) rneed.push_back(0);
      rplace[i] = {(int)rneed.size() - 1, rneed.back()};
      rneed.back() += total;
    }
    bool rok = true;
    for (size_t k = 0; k < rneed.size() && rok; k++) {
      const size_t want = std::max(kChunk, rneed[k]);
      if (k < b.rgb_chunks.size() && b.rgb_chunks[k].second >= want) continue;
      if (k < b.rgb_chunks.size()) { pinned_release(b.rgb_chunks[k].first, b.rgb_chunks[k].second); b.rgb_chunks[k] = {nullptr, 0}; }
      else b.rgb_chunks.emplace_back(nullptr, 0);
      if (pinned_acquire(&b.rgb_chunks[k].first, want, &b.rgb_chunks[k].second) != hipSuccess) { (void)hipGetLastError(); b.rgb_chunks[k] = {nullptr, 0}; rok = false; }
    }
    if (rok)
      for (size_t i = 0; i < b.params.size(); i++) {
        if (rplace[i].first < 0) continue;
        const PicParams& P = b.params[i];
        uint8_t* dst = (uint8_t*)b.rgb_chunks[(size_t)rplace[i].first].first + rplace[i].second;
        const size_t row = (size_t)P.out_width * 3;
        if (b.rgb_stride[i] == row) HIPDEC_CHECK_HIP(hipMemcpyAsync(dst, (uint8_t*)b.rgb_dev + b.rgb_off[i], row * (size_t)P.out_height, hipMemcpyDeviceToHost, s));
        else HIPDEC_CHECK_HIP(hipMemcpy2DAsync(dst, row, (uint8_t*)b.rgb_dev + b.rgb_off[i], b.rgb_stride[i], row, (size_t)P.out_height, hipMemcpyDeviceToHost, s));
        items[i].rgb = dst;
      }
    else { for (auto& c : b.rgb_chunks) pinned_release(c.first, c.second); b.rgb_chunks.clear(); }
  }
  b.host_items.swap(items);
  b.mark_done(s);
  return 0;
}

int copy_plane_d2h(const hipdec_batch& b, size_t off, uint32_t stride, int w, int h, void* dst, size_t dst_stride)
{
  const size_t es = b.wide ? 2 : 1;
  if (w <= 0 || h <= 0) return 0;
  // (a stride below the row length would put the last row past the end of a buffer of h * stride bytes)
  if (dst_stride < (size_t)w * es) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: dst_stride %zu is smaller than a row of %zu bytes", dst_stride, (size_t)w * es);
  HIPDEC_CHECK_HIP(hipMemcpy2D(dst, dst_stride, b.arena + off, stride, (size_t)w * es, (size_t)h, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace

extern "C" {

int hipdec_batch_create(hipdec_batch** out, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels)
{
  if (!out || n <= 0 || !data || !sizes) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_create: bad arguments");
  *out = nullptr;
  if (int rc = ensure_init()) return rc;
  return guarded("batch_create", [&]() -> int {
    std::unique_ptr<hipdec_batch> b(new hipdec_batch());
    if (int rc = build_batch(*b, n, data, sizes, max_image_size_pixels)) return rc;
    *out = b.release();
    return 0;
  });
}

int hipdec_batch_create_recycling(hipdec_batch** out, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels,
                                  hipdec_batch* recycle)
{
  if (!out || n <= 0 || !data || !sizes) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_create: bad arguments");
  *out = nullptr;
  if (int rc = ensure_init()) return rc;
  return guarded("batch_create", [&]() -> int {
    std::unique_ptr<hipdec_batch> b(new hipdec_batch());
    if (int rc = build_batch(*b, n, data, sizes, max_image_size_pixels, recycle)) return rc;
    *out = b.release();
    return 0;
  });
}

void hipdec_batch_free(hipdec_batch* b) { delete b; }
int hipdec_batch_count(const hipdec_batch* b) { return b ? (int)b->pics.size() : 0; }

int hipdec_batch_info(const hipdec_batch* b, int i, hipdec_image_info* info)
{
  if (!b || !info || i < 0 || i >= (int)b->pics.size()) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_info: bad arguments");
  *info = b->pics[i].info;
  return 0;
}

int hipdec_batch_run(hipdec_batch* b, void* stream)
{
  if (!b) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_run: NULL batch");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_run: the batch's arena was handed to another batch");
  if (int rc = ensure_init()) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  b->last_stream = s;
  b->ran = true;
  return launch_all(*b, s);   // (may move last_stream to the post stream)
}

int hipdec_batch_status(hipdec_batch* b)
{
  if (!b || !b->ran) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_status: batch has not been run");
  DeviceScope scope(b->device);
  if (int rc = ensure_init()) return rc;
  hipError_t e = b->wait();
  if (e != hipSuccess) return set_error(HIPDEC_ERR_DEVICE, "decode kernels failed: %s", hipGetErrorString(e));
  b->release_staging();
  const int32_t st = *b->host_status;
  if (st != 0) return set_error(HIPDEC_ERR_DECODE, "device decode error 0x%x: %s", st, dev_err_name(st));
  return 0;
}

int hipdec_batch_read_plane(hipdec_batch* b, int i, int c, void* dst, size_t dst_stride)
{
  if (!b || i < 0 || i >= (int)b->pics.size() || c < 0 || c > 2 || !dst) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: bad arguments");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: the batch's arena was handed to another batch");
  const PicParams& P = b->params[i];
  if (c > 0 && !P.chroma_format_idc) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: monochrome image has no chroma planes");
  return copy_plane_d2h(*b, P.off_out[c], P.out_stride[c], c ? P.out_cwidth : P.out_width, c ? P.out_cheight : P.out_height, dst, dst_stride);
}

int hipdec_batch_device_plane(hipdec_batch* b, int i, int c, const void** dptr, size_t* stride)
{
  if (!b || i < 0 || i >= (int)b->pics.size() || c < 0 || c > 2 || !dptr || !stride) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "device_plane: bad arguments");
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "device_plane: the batch's arena was handed to another batch");
  const PicParams& P = b->params[i];
  *dptr = b->arena + P.off_out[c];
  *stride = P.out_stride[c];
  return 0;
}

size_t hipdec_batch_item_packed_bytes(const hipdec_batch* b, int i)
{
  if (!b || i < 0 || i >= (int)b->pics.size()) return 0;
  const PicParams& P = b->params[i];
  const size_t es = b->wide ? 2 : 1;
  return ((size_t)P.out_width * P.out_height + 2 * (size_t)P.out_cwidth * P.out_cheight) * es;
}

int hipdec_batch_pack_item(hipdec_batch* b, int i, void* dst_dev, size_t dst_bytes, void* stream)
{
  if (!b || i < 0 || i >= (int)b->pics.size() || !dst_dev) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "pack_item: bad arguments");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "pack_item: the batch's arena was handed to another batch");
  if (dst_bytes < hipdec_batch_item_packed_bytes(b, i)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "pack_item: destination too small");
  const PicParams& P = b->params[i];
  const size_t es = b->wide ? 2 : 1;
  hipStream_t s = follow_stream(b, stream);
  uint8_t* dst = (uint8_t*)dst_dev;
  for (int c = 0; c < (P.chroma_format_idc ? 3 : 1); c++) {
    const size_t w = c ? P.out_cwidth : P.out_width, h = c ? P.out_cheight : P.out_height;
    if (w && h) HIPDEC_CHECK_HIP(hipMemcpy2DAsync(dst, w * es, b->arena + P.off_out[c], P.out_stride[c], w * es, h, hipMemcpyDeviceToDevice, s));
    dst += w * h * es;
  }
  b->mark_done(s);
  return 0;
}

int hipdec_copy2d_d2d(void* dst_dev, size_t dst_stride, const void* src_dev, size_t src_stride, size_t width_bytes, size_t height, void* stream)
{
  if (!dst_dev || !src_dev) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "copy2d: NULL pointer");
  if (int rc = ensure_init()) return rc;
  if (!width_bytes || !height) return 0;
  HIPDEC_CHECK_HIP(hipMemcpy2DAsync(dst_dev, dst_stride, src_dev, src_stride, width_bytes, height, hipMemcpyDeviceToDevice,
                                    stream ? (hipStream_t)stream : default_stream()));
  return 0;
}

int hipdec_batch_to_rgb(hipdec_batch* b, int i, int out_chroma, void* out_dev, size_t out_stride, void* stream)
{
  if (!b || i < 0 || i >= (int)b->pics.size() || !out_dev) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_rgb: bad arguments");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_rgb: the batch's arena was handed to another batch");
  const PicParams& P = b->params[i];
  const hipdec_image_info& I = b->pics[i].info;
  {
    const size_t bpp = out_chroma == 10 ? 3 : (out_chroma == 11 ? 4 : (out_chroma == 12 || out_chroma == 14 ? 6 : 0));
    if (bpp && out_stride < (size_t)P.out_width * bpp)
      return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_rgb: out_stride %zu is smaller than a row of %zu bytes", out_stride, (size_t)P.out_width * bpp);
  }
  if (!P.chroma_format_idc) {   // Op_mono_to_RGB24_32: 8-bit only, as in the reference
    if (b->wide || (out_chroma != 10 && out_chroma != 11)) return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: monochrome input goes to 8-bit RGB / RGBA only");
    void* ms = stream ? (void*)follow_stream(b, stream) : (void*)b->last_stream;
    const int rc = hipdec_color_mono_to_rgb24(b->arena + P.off_out[0], P.out_stride[0], nullptr, 0, P.out_width, P.out_height, out_dev, out_stride, out_chroma == 11, ms);
    b->mark_done(ms ? (hipStream_t)ms : default_stream());
    return rc;
  }
  // the decoder reports the VUI colour description exactly as the libde265 plugin would attach it
  hipdec_nclx nclx{1, I.colour_primaries, I.transfer_characteristics, I.matrix_coeffs, I.full_range_flag};
  const uint8_t* y = b->arena + P.off_out[0]; const uint8_t* cb = b->arena + P.off_out[1]; const uint8_t* cr = b->arena + P.off_out[2];
  void* s = stream ? (void*)follow_stream(b, stream) : (void*)b->last_stream;
  struct MarkDone {   // whatever is enqueued below belongs to this batch (hipdec_batch_status / free wait for it)
    hipdec_batch* b; hipStream_t s;
    ~MarkDone() { b->mark_done(s ? s : default_stream()); }
  } mark{b, (hipStream_t)s};
  if ((out_chroma == 10 || out_chroma == 11) && b->wide) {
    // > 8-bit planes to 8-bit RGB(A): one of the two chains of the planner (nearest-neighbour upsampling is this entry point's), fused into one pass
    const int m = I.matrix_coeffs == 2 ? 6 : I.matrix_coeffs;
    const int sdr_first = P.chroma_format_idc == 1 && I.full_range_flag && m != 0 && m != 8;
    return hipdec_color_hdr_to_rgb24(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, I.bit_depth_luma,
                                     P.chroma_format_idc, &nclx, out_dev, out_stride, out_chroma == 11, sdr_first, s);
  }
  if (out_chroma == 10 || out_chroma == 11) {
    // planner rule: integer op only for full range and a matrix it accepts
    const int m = I.matrix_coeffs == 2 ? 6 : I.matrix_coeffs;
    if (P.chroma_format_idc == 1 && I.full_range_flag && m != 0 && m != 8)
      return hipdec_color_420_to_rgb24(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, &nclx, out_dev,
                                       out_stride, out_chroma == 11, s);
    // 4:4:4 planes take Op_YCbCr_to_RGB<uint8_t> + Op_RGB_to_RGB24_32 whatever the range (the only chain the planner has for them)
    return hipdec_color_ycbcr_to_rgb24_float(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height,
                                             P.chroma_format_idc, &nclx, out_dev, out_stride, out_chroma == 11, s);
  }
  if (out_chroma == 12 || out_chroma == 14) {
    if (!b->wide) return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: RRGGBB output needs >8-bit planes");
    const int m = I.matrix_coeffs == 2 ? 6 : I.matrix_coeffs;
    if (P.chroma_format_idc != 1 || m == 0 || m == 8)     // planner rule: the 4:2:0 op does not take these; Op_YCbCr_to_RGB<uint16_t> + the interleave does
      return hipdec_color_ycbcr_to_rrggbb_float(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, I.bit_depth_luma,
                                                P.chroma_format_idc, &nclx, out_dev, out_stride, out_chroma == 14, s);
    return hipdec_color_420_to_rrggbb(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, I.bit_depth_luma,
                                      &nclx, out_dev, out_stride, out_chroma == 14, s);
  }
  return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: unsupported output chroma %d", out_chroma);
}

int hipdec_batch_to_rgb_all(hipdec_batch* b, int out_chroma, void* const* outs_dev, const size_t* out_strides, void* stream)
{
  if (!b || !outs_dev || !out_strides) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_rgb_all: bad arguments");
  DeviceScope scope(b->device);
  hipStream_t s = follow_stream(b, stream);
  // every item goes through the per-item entry point (argument checks, planner rule, coefficients) in capture mode
  color_capture_begin();
  for (int i = 0; i < (int)b->pics.size(); i++)
    if (int rc = hipdec_batch_to_rgb(b, i, out_chroma, outs_dev[i], out_strides[i], (void*)s)) { color_capture_abort(); return rc; }
  // device time of the colour stage goes into the timing slot of the decode run it follows
  const size_t slot = b->runs ? (b->runs - 1) % (b->ev.size() / kEv) : 0;
  hipEvent_t* ev = b->ev.data() + kEv * slot;
  HIPDEC_CHECK_HIP(hipEventRecord(ev[6], s));
  int rc = color_capture_launch(b->color, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[7], s));
  if (!rc && b->runs) b->colour_timed[slot] = 1;
  b->mark_done(s);
  return rc;
}

// decode + colour stage of every item as ONE call: for 8-bit 4:2:0 batches going to interleaved RGB24 the colour conversion is fused into
// the SAO kernel's store path (planes are still written: the ABI hands them out); every other case runs the decode kernels followed by the
// batched colour kernel, exactly like hipdec_batch_run + hipdec_batch_to_rgb_all.
int hipdec_batch_run_rgb(hipdec_batch* b, int out_chroma, void* const* outs_dev, const size_t* out_strides, void* stream)
{
  if (!b || !outs_dev || !out_strides) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "run_rgb: bad arguments");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "run_rgb: the batch's arena was handed to another batch");
  if (int rc = ensure_init()) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  static const bool no_fusion = getenv("HIPDEC_NO_SAO_RGB_FUSION") != nullptr;
  bool fuse = out_chroma == 10 && !b->wide && !no_fusion;
  for (const auto& P : b->params) fuse = fuse && P.chroma_format_idc == 1;
  if (!fuse) {
    if (int rc = hipdec_batch_run(b, stream)) return rc;
    return hipdec_batch_to_rgb_all(b, out_chroma, outs_dev, out_strides, nullptr);
  }
  // the per-item entry points check the arguments, apply the planner rule and fill the coefficient blocks (capture mode: nothing is launched)
  hipStream_t ps = stage_overlap() ? post_stream() : s;
  color_capture_begin();
  for (int i = 0; i < (int)b->pics.size(); i++)
    if (int rc = hipdec_batch_to_rgb(b, i, out_chroma, outs_dev[i], out_strides[i], (void*)ps)) { color_capture_abort(); return rc; }
  const void* dev = nullptr;
  int variant = -1, count = 0;
  if (int rc = color_capture_take(b->color, ps, &dev, &variant, &count)) return rc;
  if (variant != color_variant_rgb24_u8() || count != (int)b->pics.size()) return set_error(HIPDEC_ERR_UNSUPPORTED, "run_rgb: unexpected colour variant");
  b->last_stream = s;
  b->ran = true;
  b->fused_rgb = true;
  return launch_all(*b, s, dev);
}

int hipdec_batch_timing_slots(hipdec_batch* b, int slots)
{
  if (!b || slots < 1 || slots > 4096) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "timing_slots: bad arguments");
  DeviceScope scope(b->device);
  if (b->ran) HIPDEC_CHECK_HIP(b->wait());
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  b->ev.assign((size_t)slots * kEv, nullptr);
  b->colour_timed.assign((size_t)slots, 0);
  for (auto& e : b->ev) HIPDEC_CHECK_HIP(hipEventCreate(&e));
  b->runs = 0; b->ran = false;
  return 0;
}

int hipdec_batch_slot_kernel_timing_us(hipdec_batch* b, int slot, float out[8])
{
  if (!b || !out || slot < 0 || (size_t)slot >= b->ev.size() / kEv || (uint64_t)slot >= b->runs)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "slot_timing: no run recorded in slot %d", slot);
  DeviceScope scope(b->device);
  hipEvent_t* ev = b->ev.data() + kEv * (size_t)slot;
  HIPDEC_CHECK_HIP(hipEventSynchronize(ev[5]));
  for (int k = 0; k < 5; k++) {
    float ms = 0;
    HIPDEC_CHECK_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
    out[k] = ms * 1000.0f;
  }
  out[5] = 0.0f;
  if (b->colour_timed[(size_t)slot]) {
    float ms = 0;
    HIPDEC_CHECK_HIP(hipEventSynchronize(ev[7]));
    HIPDEC_CHECK_HIP(hipEventElapsedTime(&ms, ev[6], ev[7]));
    out[5] = ms * 1000.0f;
  }
  float ms = 0;
  HIPDEC_CHECK_HIP(hipEventElapsedTime(&ms, ev[0], ev[5]));
  out[6] = ms * 1000.0f;
  out[7] = 0.0f;
  return 0;
}

int hipdec_batch_slot_timing_us(hipdec_batch* b, int slot, float out[5])
{
  float k[8];
  if (int rc = hipdec_batch_slot_kernel_timing_us(b, slot, k)) return rc;
  out[0] = k[0]; out[1] = k[1] + k[2]; out[2] = k[3]; out[3] = k[4]; out[4] = k[6];
  return 0;
}

int hipdec_batch_last_timing_us(hipdec_batch* b, float out[5])
{
  if (!b || !b->ran || !out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "last_timing: batch has not been run");
  return hipdec_batch_slot_timing_us(b, (int)((b->runs - 1) % (b->ev.size() / kEv)), out);
}

int hipdec_batch_read_tap(hipdec_batch* b, int i, int which, int c, void* dst, size_t dst_stride)
{
  if (!b || i < 0 || i >= (int)b->pics.size() || c < 0 || c > 2 || !dst) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_tap: bad arguments");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_tap: the batch's arena was handed to another batch");
  (void)which;  // the reconstruction buffer holds the deblocked picture after a full run
  const PicParams& P = b->params[i];
  return copy_plane_d2h(*b, P.off_rec[c], P.rec_stride[c], c ? P.cwidth : P.width, c ? P.cheight : P.height, dst, dst_stride);
}

int hipdec_batch_read_maps(hipdec_batch* b, int i, uint8_t* log2_tb, uint8_t* log2_cb, uint8_t* intra_luma, uint8_t* intra_chroma, int8_t* qp_y,
                           uint8_t* flags, size_t map_elems)
{
  if (!b || i < 0 || i >= (int)b->pics.size()) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_maps: bad arguments");
  DeviceScope scope(b->device);
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_maps: the batch's arena was handed to another batch");
  const PicParams& P = b->params[i];
  const int uw = (P.width + 3) / 4, uh = (P.height + 3) / 4;
  if (map_elems < (size_t)uw * uh) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_maps: buffers too small");
  const size_t nunits = ((size_t)P.ctb_w * P.ctb_h) << P.units_per_ctb_log2;
  std::vector<uint8_t> sz(nunits), fl(nunits), ipm(nunits), ipmc(nunits), qp(nunits);
  HIPDEC_CHECK_HIP(hipMemcpy(sz.data(), b->arena + P.off_u_size, nunits, hipMemcpyDeviceToHost));
  HIPDEC_CHECK_HIP(hipMemcpy(fl.data(), b->arena + P.off_u_flags, nunits, hipMemcpyDeviceToHost));
  HIPDEC_CHECK_HIP(hipMemcpy(ipm.data(), b->arena + P.off_u_ipm, nunits, hipMemcpyDeviceToHost));
  HIPDEC_CHECK_HIP(hipMemcpy(ipmc.data(), b->arena + P.off_u_ipmc, nunits, hipMemcpyDeviceToHost));
  HIPDEC_CHECK_HIP(hipMemcpy(qp.data(), b->arena + P.off_u_qp, nunits, hipMemcpyDeviceToHost));
  const int l = P.log2_ctb - 2, mask = (1 << l) - 1;
  auto il = [](uint32_t x, uint32_t y) {
    x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55; y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55; return x | (y << 1);
  };
  for (int uy = 0; uy < uh; uy++)
    for (int ux = 0; ux < uw; ux++) {
      const size_t idx = (((size_t)(uy >> l) * P.ctb_w + (ux >> l)) << P.units_per_ctb_log2) + il(ux & mask, uy & mask);
      const size_t o = (size_t)uy * uw + ux;
      if (log2_tb) log2_tb[o] = sz[idx] & 15;
      if (log2_cb) log2_cb[o] = sz[idx] >> 4;
      if (intra_luma) intra_luma[o] = ipm[idx] & 63;
      if (intra_chroma) intra_chroma[o] = ipmc[idx];
      if (qp_y) qp_y[o] = (int8_t)qp[idx];
      if (flags) flags[o] = fl[idx];
    }
  return 0;
}

int hipdec_probe(const void* data, size_t size, uint64_t max_image_size_pixels, hipdec_image_info* info)
{
  if (!data || !info) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "probe: bad arguments");
  return guarded("probe", [&]() -> int {
    ParsedPicture pp;
    std::string err;
    int rc = parse_picture((const uint8_t*)data, size, max_image_size_pixels, pp, err);
    if (rc != HIPDEC_OK) return set_error(rc, "%s", err.c_str());
    *info = pp.info;
    return 0;
  });
}


}  // extern "C"

// The rest of this translation unit, by subject (one 2600-line file held five subsystems):
#include "decoder_object.inc"          // the decoder object of the plugin life cycle + the still-image coalescer
#include "decoder_chains.inc"          // look-ahead chains of sequence tracks, DPB / output order
#include "decoder_color_boundary.inc"  // resident planes, colour planner, hipdec_color_convert, image transforms
#include "decoder_grid.inc"            // grid photos over the node's GPUs
