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
  }
  ~hipdec_batch()
  {
    DeviceScope scope(device);
    if (arena || staging) (void)wait();   // nothing of this batch may still be running when the arena is recycled
    release_staging();
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
  if (b.upload_size > (size_t(4) << 20) && !sync_upload) {
    HIPDEC_CHECK_HIP(pinned_acquire(&b.staging, b.upload_size, &b.staging_capacity));
    if (!b.arena) HIPDEC_CHECK_HIP(arena_acquire((void**)&b.arena, b.arena_size, &b.arena_capacity));   // (first: a chain's reference tables hold addresses inside it)
    layout_batch_fill(b, data, sizes, (uint8_t*)b.staging, (uint64_t)(uintptr_t)b.arena);
    HIPDEC_CHECK_HIP(hipEventCreateWithFlags(&b.uploaded, hipEventDisableTiming));
    hipStream_t us = upload_stream();
    if (after) HIPDEC_CHECK_HIP(hipStreamWaitEvent(us, after, 0));
    HIPDEC_CHECK_HIP(hipMemcpyAsync(b.arena, b.staging, b.upload_size, hipMemcpyHostToDevice, us));
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
      launch_chain_motion_all(b, b.arena, ms);
      HIPDEC_CHECK_HIP(hipEventRecord(b.chain_events[0], ms));
      stream_release(ms);
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
      HIPDEC_CHECK_HIP(hipMemcpyAsync(b.host_status, b.arena + b.off_status, sizeof(int32_t), hipMemcpyDeviceToHost, ps));
      b.last_stream = ps;
      b.mark_done(ps);
      return 0;
    }
    hipStream_t ms = nullptr;
    if (b.any_inter && b.motion_steps.size() > 1) {
      ms = stream_acquire();
      while (b.chain_events.size() < b.motion_steps.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { stream_release(ms); return set_error(HIPDEC_ERR_DEVICE, "chain: no event"); }
        b.chain_events.push_back(e);
      }
      HIPDEC_CHECK_HIP(hipStreamWaitEvent(ms, ev[1], 0));
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
    HIPDEC_CHECK_HIP(hipMemcpyAsync(b.host_status, b.arena + b.off_status, sizeof(int32_t), hipMemcpyDeviceToHost, ps));
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
    if (fused_rgb_params) launch_sao_rgb(fa, fused_rgb_params, n, b.max_ow, b.max_oh, ps, may_keep, restricted);   // SAO + crop + RGB24 in one pass
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

// ---- single-image decoder: the plugin life cycle ------------------------------------------------
//
// libheif drives one decoder instance per coded item and, for 'grid' images, one worker thread per tile
// (libheif/image-items/grid.cc:405-453).  A CABAC substream is sequential, so a lone tile occupies a handful of
// waves for its whole latency; the GPU only pays off when the tiles of a photo are decoded TOGETHER.  The plugin
// boundary has no batch call, so concurrent hipdec_decoder_decode() calls are coalesced here: the first caller
// becomes the leader, gathers the requests of the other threads for a short window and runs ONE batch (one upload,
// one set of launches) for all of them; every instance then reads its own planes out of the shared batch.
// A host that decodes serially never waits: the window only opens when other instances exist that have not decoded
// yet, or when overlapping requests were seen a moment ago.
struct hipdec_decoder {
  int strict = 0;
  uint64_t max_pixels = 0;
  std::shared_ptr<hipdec_batch> batch;   // shared with the other instances decoded in the same launch
  int item = 0;                          // this instance's picture inside `batch`
  bool decoded = false;
  bool counted = false;                  // included in Coalescer::armed
  // ---- sequence tracks: a push after a decode continues a sequence; the pictures decoded so far that later P pictures may
  //      reference stay alive here (their batches' arenas, or an uncropped copy when the conformance window cuts samples off), and the POC
  //      state / reference picture list of 8.3.1 / 8.3.2 lives in `seq`
  struct DpbHold { int poc = 0; std::shared_ptr<hipdec_batch> keep; void* full = nullptr; size_t full_capacity = 0; int device = 0; };
  bool seq_active = false;
  SeqContext seq;
  std::vector<DpbHold> dpb;
  // ---- output order (C.5.2.2 "bumping"): with B pictures the coding order is not the output order.  Decoded pictures wait here until more
  //      than sps_max_num_reorder_pics of their coded video sequence are waiting (or the host flushes); hipdec_decoder_next_picture hands them
  //      out by increasing POC.  `out` is the picture the plane readers currently serve (empty: the picture decoded last, the still-image use)
  struct Output { std::shared_ptr<hipdec_batch> batch; int item = 0; int poc = 0; uint64_t cvs = 0; uintptr_t user_data = 0; bool pic_output = true; };
  std::vector<Output> waiting;
  Output out;
  uint64_t cvs = 0;                      // coded video sequence counter (a new one starts at every IDR / first IRAP)
  // ---- look-ahead: the samples behind the first picture wait here (one access unit each, the parameter sets known at its push in front) until
  //      HIPDEC_SEQ_LOOKAHEAD of them are there or the host flushes; they are then decoded as ONE chain (batch_layout.h): one CABAC launch over all of
  //      them - parsing needs nothing of another picture - and the pixel stages picture by picture.  libheif's track loop pushes the next sample
  //      whenever decode_next_image2 returns no image (sequences/track_visual.cc:200-260), so holding samples back costs it nothing.
  SampleQueue sq;                        // (hevc_headers.h: pure host logic, tested and fuzzed on the CPU)
  std::chrono::steady_clock::time_point chain_active{};   // when the instance last asked for / got a chain (ChainCoalescer: whom a leader waits for)
  bool chain_in_flight = false;                           // its chain is inside a running launch set
  hipdec_batch* plane_batch() const { return out.batch ? out.batch.get() : batch.get(); }
  int plane_item() const { return out.batch ? out.item : item; }
  ~hipdec_decoder()
  {
    for (auto& h : dpb) if (h.full) { DeviceScope scope(h.device); arena_release(h.full, h.full_capacity); }
  }
};

namespace {

using Clock = std::chrono::steady_clock;

struct DecodeRequest {
  hipdec_decoder* d = nullptr;
  int rc = 0;
  std::string err;
  bool taken = false, done = false;
};

struct Coalescer {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<DecodeRequest*> pending;
  bool collecting = false;               // a leader is gathering `pending`
  int armed = 0;                         // live instances that have not been decoded (potential joiners)
  int in_flight = 0;                     // requests inside a running batch
  Clock::time_point last_arrival{}, last_overlap{};
  long window_us = -1, quiet_us = 300;
  long busy_requests = 16;               // a leader keeps gathering while max_sets launch sets with more requests than this are running ...
  long hold_us = 1000000;                // ... for at most this long (HIPDEC_COALESCE_BUSY / HIPDEC_COALESCE_HOLD_US).  The hold only happens while max_sets big
                                         // sets occupy the GPU, i.e. when a new set could not start earlier anyway; 300 ms (tried in round 4 for ADVICE round 3's
                                         // "a lone still behind a holding leader waits that long") launches small extra sets instead and costs throughput:
                                         // 2.56 / 4.68 against 2.88 / 5.66 Gpixel/s through libheif with 256 / 1024 threads (profiles/r04_dropin_throughput.txt)
  int max_sets = 3;                      // launch sets in flight before a leader holds (HIPDEC_COALESCE_SETS): they overlap, each with a third of the pool
                                         // waves (measured, direct C ABI, 256 / 1024 threads: 3.8 / 7.4 Gpixel/s with 2, 4.2 / 7.9 with 3, 3.5 / 7.7 with 4)
  long max_set = 256;                    // requests per launch set at most (HIPDEC_COALESCE_MAX_SET): keeps the sets' arenas and staging buffers in a
                                         // few size classes the pools can serve (a 683-still set spent 1.3 s in hipMalloc / hipHostMalloc), and a batch
                                         // of 256 4K stills already parses within ~25 % of the asymptotic rate
  int sets_in_flight = 0;
  uint64_t n_requests = 0, n_launch_sets = 0, n_shared = 0;   // statistics (hipdec_decoder_coalesce_stats)
} g_co;

long coalesce_window_us()
{
  // the knobs are read ONCE, by whichever thread asks first (a function-local static: initialised under the language's own lock - application threads
  // reach this concurrently, and an unguarded "if (window_us < 0)" was a data race ThreadSanitizer pointed at on the host build of the library)
  static const bool once = [] {
    const char* e = std::getenv("HIPDEC_COALESCE_WINDOW_US");   // 0 disables coalescing
    if (const char* q = std::getenv("HIPDEC_COALESCE_QUIET_US")) g_co.quiet_us = std::max(1L, std::atol(q));
    if (const char* q = std::getenv("HIPDEC_COALESCE_BUSY")) g_co.busy_requests = std::max(0L, std::atol(q));
    if (const char* q = std::getenv("HIPDEC_COALESCE_HOLD_US")) g_co.hold_us = std::max(0L, std::atol(q));
    if (const char* q = std::getenv("HIPDEC_COALESCE_SETS")) g_co.max_sets = (int)std::max(1L, std::atol(q));
    if (const char* q = std::getenv("HIPDEC_COALESCE_MAX_SET")) g_co.max_set = std::max(1L, std::atol(q));
    g_co.window_us = e ? std::max(0L, std::atol(e)) : 2000;
    return true;
  }();
  (void)once;
  return g_co.window_us;
}

// hipdec_batch_create with the sequence contexts of the decoder instances (P pictures name their reference pictures through them)
int create_batch_seq(hipdec_batch** out, int n, const void* const* data, const size_t* sizes, uint64_t max_pixels, const SeqContext* const* seqs)
{
  *out = nullptr;
  if (int rc = ensure_init()) return rc;
  return guarded("batch_create", [&]() -> int {
    std::unique_ptr<hipdec_batch> b(new hipdec_batch());
    if (int rc = build_batch(*b, n, data, sizes, max_pixels, nullptr, seqs)) return rc;
    *out = b.release();
    return 0;
  });
}

// The picture the instance decoded last becomes a reference picture of the sequence (called when the next sample is pushed): POC state and DPB
// follow 8.3.1 / 8.3.2 (seq_commit), its planes are the batch's output planes when they ARE the coded picture - else one more SAO pass
// without the conformance window writes an uncropped copy (references are addressed in coded coordinates, the rows a window cuts off included).
int commit_reference(hipdec_decoder* d)
{
  hipdec_batch* b = d->batch.get();
  if (!b || d->item < 0 || d->item >= (int)b->pics.size()) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "sequence: no decoded picture to keep");
  DeviceScope scope(b->device);
  const ParsedPicture& pp = b->pics[(size_t)d->item];
  const PicParams& P = b->params[(size_t)d->item];
  seq_commit(d->seq, pp);
  // drop what the RPS no longer names
  for (size_t i = 0; i < d->dpb.size();) {
    bool keep = false;
    for (const RefPicture& rp : d->seq.dpb) if (rp.poc == d->dpb[i].poc) keep = true;
    if (keep) { i++; continue; }
    if (d->dpb[i].full) arena_release(d->dpb[i].full, d->dpb[i].full_capacity);
    d->dpb.erase(d->dpb.begin() + (long)i);
  }
  hipdec_decoder::DpbHold h;
  h.poc = pp.poc; h.keep = d->batch; h.device = b->device;
  RefPicture rp;
  rp.poc = pp.poc;
  rp.width = P.width; rp.height = P.height; rp.chroma_format_idc = P.chroma_format_idc; rp.bit_depth_luma = P.bit_depth_luma; rp.bit_depth_chroma = P.bit_depth_chroma;
  rp.log2_ctb = P.log2_ctb;
  const bool cropped = P.out_width != P.width || P.out_height != P.height || P.crop_x || P.crop_y;
  // A first picture that was decoded in a SHARED launch set (the coalescer put up to 256 instances' pictures into one arena) gets a copy of its own as
  // well: holding the shared arena for as long as the track references the picture would pin every other instance's memory with it
  const bool shared_set = b->pics.size() > 1;
  if (!cropped && !shared_set) {
    for (int c = 0; c < 3; c++) { rp.plane[c] = (uint64_t)(uintptr_t)(b->arena + P.off_out[c]); rp.stride[c] = P.out_stride[c]; }
  } else {
    const size_t es = b->wide ? 2 : 1;
    size_t off[3], total = 256;   // [0, 256): the PicParams copy the SAO launch reads
    uint32_t stride[3];
    for (int c = 0; c < 3; c++) {
      const size_t w = c ? (size_t)P.cwidth : (size_t)P.width, hh = c ? (size_t)P.cheight : (size_t)P.height;
      stride[c] = (uint32_t)(((w ? w : 1) * es + 63) / 64 * 64);
      off[c] = total; total += (size_t)stride[c] * (hh ? hh : 1) + 256;
    }
    static_assert(sizeof(PicParams) <= 512, "PicParams copy");
    total += 512;
    HIPDEC_CHECK_HIP(arena_acquire(&h.full, total, &h.full_capacity));
    uint8_t* base = (uint8_t*)h.full + 512;
    PicParams F = P;
    F.crop_x = F.crop_y = 0; F.out_width = P.width; F.out_height = P.height; F.out_cwidth = P.cwidth; F.out_cheight = P.cheight;
    for (int c = 0; c < 3; c++) { F.off_out[c] = (uint64_t)(uintptr_t)(base + off[c]) - (uint64_t)(uintptr_t)b->arena; F.out_stride[c] = stride[c]; }   // (offsets are added to the arena base)
    hipStream_t s = default_stream();
    hipError_t e = hipMemcpyAsync(h.full, &F, sizeof(F), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
      FilterArgs fa{(const PicParams*)h.full, b->arena, (const int32_t*)(b->arena + b->off_status)};
      const bool may_keep = P.transquant_bypass_enabled || (P.pcm_enabled && P.pcm_loop_filter_disabled), restricted = !P.sao_free_neighbours;
      launch_sao(fa, 1, P.width, P.height, b->wide, s, may_keep, restricted);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) { arena_release(h.full, h.full_capacity); return set_error(HIPDEC_ERR_DEVICE, "sequence: reference picture copy: %s", hipGetErrorString(e)); }
    for (int c = 0; c < 3; c++) { rp.plane[c] = (uint64_t)(uintptr_t)(base + off[c]); rp.stride[c] = stride[c]; }
    if (!P.is_inter) h.keep.reset();   // the copy is all later pictures need of an intra picture (an inter picture's motion field lives in the arena)
  }
  rp.mf = P.is_inter ? (uint64_t)(uintptr_t)(b->arena + P.off_mf) : 0;   // the collocated picture of later temporal candidates (the batch stays alive with it)
  d->seq.dpb.push_back(rp);
  d->dpb.push_back(std::move(h));
  return 0;
}

// hipdec_batch_run for a launch set of the decoder path - or, when the host has been converting the images it decodes to interleaved RGB24, the same
// launch set with the colour conversion fused into the SAO kernel's store path (hipdec_batch_run_rgb) into a device buffer the batch owns: resident RGB
std::atomic<uint64_t> g_rgb_produced{0}, g_rgb_served{0};
int run_decoder_batch(hipdec_batch* b, hipStream_t s)
{
  bool rgb = rgb_wanted() && !b->wide && !b->chain && !b->any_inter && !b->rgb_dev;
  for (const PicParams& P : b->params) rgb = rgb && P.chroma_format_idc == 1 && P.out_width > 0 && P.out_height > 0;
  if (!rgb) return hipdec_batch_run(b, (void*)s);
  size_t total = 0;
  b->rgb_off.clear(); b->rgb_stride.clear();
  for (const PicParams& P : b->params) {
    const size_t stride = ((size_t)P.out_width * 3 + 255) & ~size_t(255);
    b->rgb_off.push_back(total); b->rgb_stride.push_back(stride);
    total += stride * (size_t)P.out_height;
  }
  {
    DeviceScope scope(b->device);
    if (arena_acquire(&b->rgb_dev, total, &b->rgb_capacity) != hipSuccess) { (void)hipGetLastError(); b->rgb_dev = nullptr; b->rgb_off.clear(); b->rgb_stride.clear(); return hipdec_batch_run(b, (void*)s); }
  }
  std::vector<void*> outs;
  for (size_t i = 0; i < b->params.size(); i++) outs.push_back((uint8_t*)b->rgb_dev + b->rgb_off[i]);
  const int rc = hipdec_batch_run_rgb(b, 10, outs.data(), b->rgb_stride.data(), (void*)s);
  if (rc == 0 && b->fused_rgb) { g_rgb_produced += b->params.size(); return 0; }
  // (items whose colour descriptions ask for different kernels, or a run that did not take the fused form: the RGB is dropped and the planes-only
  //  launch set follows.  A failed run_rgb may have queued kernels that write into rgb_dev - the unfused form converts item by item behind a
  //  complete decode -, so the stream is drained on EVERY path before the buffer goes back to the pool)
  { DeviceScope scope(b->device); if (rc == 0) (void)b->wait(); else (void)hipStreamSynchronize(s); arena_release(b->rgb_dev, b->rgb_capacity); }
  b->rgb_dev = nullptr; b->rgb_off.clear(); b->rgb_stride.clear();
  return rc == 0 ? 0 : hipdec_batch_run(b, (void*)s);
}

// one decoder in a batch of its own: the reference behaviour, and the fallback that gives every request its own
// error when a shared batch could not be built or failed on the device
void run_single(DecodeRequest& r, hipStream_t s)
{
  hipdec_decoder* d = r.d;
  const void* ptrs[1] = {d->sq.first.data()};
  const size_t sizes[1] = {d->sq.first.size()};
  hipdec_batch* b = nullptr;
  const SeqContext* seqs[1] = {d->seq_active ? &d->seq : nullptr};
  r.rc = create_batch_seq(&b, 1, ptrs, sizes, d->max_pixels, seqs);
  if (!r.rc) {
    r.rc = run_decoder_batch(b, s);
    if (!r.rc) r.rc = stage_planes_to_host(*b, follow_stream(b, (void*)s));
    if (!r.rc) r.rc = hipdec_batch_status(b);   // synchronises s
    else (void)hipStreamSynchronize(s);
    b->last_stream = nullptr;                   // the stream goes back to the pool: nothing of this batch is in flight
  }
  {
    std::lock_guard<std::mutex> lock(g_co.mu);
    g_co.n_launch_sets++;
  }
  if (r.rc) { r.err = hipdec_last_error(); delete b; return; }
  d->batch.reset(b);
  d->item = 0;
}

void run_group(std::vector<DecodeRequest*>& group, hipStream_t s, uint32_t wave_share)
{
  if (group.size() == 1) { run_single(*group[0], s); return; }
  std::vector<const void*> ptrs;
  std::vector<size_t> sizes;
  std::vector<const SeqContext*> seqs;
  for (auto* r : group) { ptrs.push_back(r->d->sq.first.data()); sizes.push_back(r->d->sq.first.size()); seqs.push_back(r->d->seq_active ? &r->d->seq : nullptr); }
  hipdec_batch* b = nullptr;
  static const bool trace = getenv("HIPDEC_COALESCE_TRACE") != nullptr;   // dev knob: where a launch set's wall time goes
  const auto t0 = Clock::now();
  int rc = create_batch_seq(&b, (int)group.size(), ptrs.data(), sizes.data(), group[0]->d->max_pixels, seqs.data());
  const auto t1 = Clock::now();
  auto t2 = t1, t3 = t1;
  if (!rc) {
    b->wave_share = wave_share;   // the launch sets in flight when this one started share the CABAC pool's wave slots (a lone burst gets them all)
    rc = run_decoder_batch(b, s);
    t2 = Clock::now();
    if (!rc) rc = stage_planes_to_host(*b, follow_stream(b, (void*)s));
    t3 = Clock::now();
    if (!rc) rc = hipdec_batch_status(b);
    else (void)hipStreamSynchronize(s);
    b->last_stream = nullptr;
  }
  if (trace) {
    auto ms = [](Clock::time_point a, Clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    fprintf(stderr, "[hipdec] launch set of %zu: create %.1f ms, launch %.1f ms, stage-enqueue %.1f ms, wait %.1f ms\n", group.size(), ms(t0, t1), ms(t1, t2),
            ms(t2, t3), ms(t3, Clock::now()));
  }
  if (!rc) {
    std::shared_ptr<hipdec_batch> sp(b);
    for (size_t i = 0; i < group.size(); i++) { group[i]->d->batch = sp; group[i]->d->item = (int)i; group[i]->rc = 0; }
    std::lock_guard<std::mutex> lock(g_co.mu);
    g_co.n_launch_sets++; g_co.n_shared += group.size();
    return;
  }
  delete b;
  // a bad item, or a mix the batch layout refuses (8-bit with 10-bit items): halve the group until the culprit is alone,
  // so that it alone gets the error and the others still share launch sets
  std::vector<DecodeRequest*> lo(group.begin(), group.begin() + (long)(group.size() / 2)), hi(group.begin() + (long)(group.size() / 2), group.end());
  run_group(lo, s, wave_share);
  run_group(hi, s, wave_share);
}

void run_requests(std::vector<DecodeRequest*>& take, uint32_t wave_share)
{
  hipStream_t s = stream_acquire();   // own stream per launch set: batches of different leaders overlap on the GPU
  std::vector<bool> used(take.size(), false);
  for (size_t i = 0; i < take.size(); i++) {
    if (used[i]) continue;
    std::vector<DecodeRequest*> group;   // security limits are per instance: only equal limits share a batch
    for (size_t j = i; j < take.size(); j++)
      if (!used[j] && take[j]->d->max_pixels == take[i]->d->max_pixels) { used[j] = true; group.push_back(take[j]); }
    run_group(group, s, wave_share);
  }
  stream_release(s);
}

void uncount(hipdec_decoder* d)   // g_co.mu held
{
  if (d->counted) { d->counted = false; g_co.armed--; }
}

}  // namespace

namespace { void chain_member_add(hipdec_decoder* d); void chain_member_remove(hipdec_decoder* d); void g_chains_notify(); }
// the first picture was decoded and more data arrives: the instance becomes a sequence decoder; that picture may be referenced by the samples that follow
static int seq_activate(hipdec_decoder* d)
{
  if (int rc = commit_reference(d)) return rc;
  d->seq_active = true;
  chain_member_add(d);
  return 0;
}

int hipdec_decoder_new(hipdec_decoder** out, int strict_decoding, uint64_t max_image_size_pixels)
{
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "decoder_new: out is NULL");
  *out = nullptr;
  if (int rc = ensure_init()) return rc;  // fail loudly when there is no GPU: there is no CPU fallback
  hipdec_decoder* d = new (std::nothrow) hipdec_decoder();
  if (!d) return set_error(HIPDEC_ERR_MEMORY, "decoder_new: out of host memory");
  d->strict = strict_decoding; d->max_pixels = max_image_size_pixels;
  {
    std::lock_guard<std::mutex> lock(g_co.mu);
    d->counted = true;
    g_co.armed++;
  }
  *out = d;
  return 0;
}

void hipdec_decoder_free(hipdec_decoder* d)
{
  if (!d) return;
  {
    std::lock_guard<std::mutex> lock(g_co.mu);
    uncount(d);
    g_co.cv.notify_all();   // a leader may be waiting for this instance to join
  }
  if (d->seq_active) {
    chain_member_remove(d);
    g_chains_notify();      // (the same for the leader of a chain launch set)
  }
  delete d;
}

void hipdec_decoder_set_strict(hipdec_decoder* d, int strict) { if (d) d->strict = strict; }

int hipdec_decoder_push_data(hipdec_decoder* d, const void* data, size_t size)
{
  if (!d || (!data && size)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "push_data: bad arguments");
  // validate the framing now, as decoder_libde265.cc:322-368 does
  const uint8_t* p = (const uint8_t*)data;
  size_t ptr = 0;
  while (ptr < size) {
    if (size - ptr < 4) return set_error(HIPDEC_ERR_END_OF_DATA, "truncated NAL length field");
    uint32_t n = ((uint32_t)p[ptr] << 24) | ((uint32_t)p[ptr + 1] << 16) | ((uint32_t)p[ptr + 2] << 8) | p[ptr + 3];
    ptr += 4;
    if (n > size - ptr) return set_error(HIPDEC_ERR_END_OF_DATA, "NAL size exceeds the pushed data");
    ptr += n;
  }
  return guarded("push_data", [&]() -> int {
    // Access units (7.4.2.4.4): a coded slice segment with first_slice_segment_in_pic_flag, or a parameter set / AUD / prefix SEI behind the last
    // slice of a picture, starts the next one.  The first access unit is the still-image case (everything pushed before the decode, as
    // decoder_libde265.cc:322-368 takes it); every later one is a sample of a sequence track (libheif/sequences/track_visual.cc:200-280 pushes them
    // one by one; only a chunk's first sample carries the parameter sets, codecs/decoder.cc:422) and waits in the look-ahead queue.
    if (d->decoded && !d->seq_active) {
      // the first picture was decoded and more data arrives: the instance becomes a sequence decoder; that picture may be referenced by the samples that follow
      if (int rc = seq_activate(d)) return rc;
      d->sq.first_closed = true;
    }
    d->sq.push(p, size);
    return 0;
  });
}

static int decoder_decode_impl(hipdec_decoder* d, hipdec_image_info* info);
static int decode_chain(hipdec_decoder* d, size_t n, std::vector<hipdec_decoder::Output>* outputs);
static std::atomic<long> g_seq_lookahead{-1};
static long seq_lookahead()
{
  long k = g_seq_lookahead.load(std::memory_order_relaxed);
  if (k < 0) {
    const char* e = std::getenv("HIPDEC_SEQ_LOOKAHEAD");
    k = e ? std::atol(e) : 32;
    k = k < 0 ? 0 : (k > 64 ? 64 : k);
    g_seq_lookahead.store(k, std::memory_order_relaxed);
  }
  return k;
}
void hipdec_set_sequence_lookahead(int samples) { g_seq_lookahead.store(samples < 0 ? 0 : (samples > 64 ? 64 : samples), std::memory_order_relaxed); }
// decode_next_image in DECODING order (the still-image call, and sequence hosts that want every sample's picture at once): the first picture, or the
// oldest queued sample on its own (a chain of one); the plane readers then serve that picture.  Output order: hipdec_decoder_next_picture.
int hipdec_decoder_decode(hipdec_decoder* d, hipdec_image_info* info)
{
  return guarded("decode", [&]() -> int {
    if (!d) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "decode: NULL decoder");
    if (!d->decoded) return decoder_decode_impl(d, info);
    for (;;) {
      while (!d->sq.queue.empty() && !d->sq.queue.front().has_vcl && d->sq.queue.size() > 1) d->sq.queue.pop_front();   // (parameter sets / SEI only: nothing to decode)
      if (d->sq.queue.empty() || !d->sq.queue.front().has_vcl) return set_error(HIPDEC_ERR_NO_IMAGE, "no further image");
      if (!d->seq_active) { if (int rc = seq_activate(d)) return rc; }
      std::vector<hipdec_decoder::Output> outs;
      if (int rc = decode_chain(d, 1, &outs)) return rc;
      if (outs.empty()) continue;   // the sample was a RASL picture 8.3.3 drops: the next one
      d->out = hipdec_decoder::Output{};
      d->batch = outs[0].batch; d->item = outs[0].item;
      if (info) *info = d->batch->pics[(size_t)d->item].info;
      return 0;
    }
  });
}
static int decoder_decode_impl(hipdec_decoder* d, hipdec_image_info* info)
{
  if (!d) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "decode: NULL decoder");
  if (d->decoded) return set_error(HIPDEC_ERR_NO_IMAGE, "no further image");
  if (d->sq.first.empty() || !d->sq.first_has_vcl) return set_error(HIPDEC_ERR_NO_IMAGE, "no data was pushed");
  DecodeRequest req;
  req.d = d;
  const long window = coalesce_window_us();
  {
    std::lock_guard<std::mutex> lock(g_co.mu);
    g_co.n_requests++;
  }
  if (window == 0) {
    hipStream_t s = stream_acquire();
    run_single(req, s);
    stream_release(s);
    std::lock_guard<std::mutex> lock(g_co.mu);
    uncount(d);
  } else {
    std::unique_lock<std::mutex> lk(g_co.mu);
    const auto now = Clock::now();
    if (!g_co.pending.empty() || g_co.in_flight > 0) g_co.last_overlap = now;
    g_co.last_arrival = now;
    g_co.pending.push_back(&req);
    g_co.cv.notify_all();
    while (!req.done) {
      if (req.taken || g_co.collecting) { g_co.cv.wait(lk); continue; }
      // leader: gather the requests of the other threads, then run them as one batch
      g_co.collecting = true;
      std::vector<DecodeRequest*> take;
      bool counted_in_flight = false;
      try {
        const auto t0 = Clock::now();
        const auto deadline = t0 + std::chrono::microseconds(window);
        const bool overlapping = g_co.last_overlap.time_since_epoch().count() != 0 &&
                                 t0 - g_co.last_overlap < std::chrono::milliseconds(250);
        // Batching while busy: while max_sets launch sets with many requests are running, another small set beside them buys nothing
        // (a lone still is one CABAC critical path, ~250 ms, whatever else runs) and gathering lets the NEXT set be large enough for the
        // work pool; so the leader keeps collecting until one of them is done.  A few sets overlap (each with its share of the pool's
        // waves): the application threads rotate through them and one set's dependency tail is covered by the others' bulk.
        // A host with a few threads never holds; one with hundreds gets sets of a hundred instead of sets of three (measured through the
        // real libheif, 256 threads x 4K stills: 0.36 -> 2.4 Gpixel/s with one set at a time, profiles/r03_dropin_*.txt).
        const auto hold_until = t0 + std::chrono::microseconds(g_co.hold_us);
        while (g_co.sets_in_flight >= g_co.max_sets && g_co.in_flight > g_co.busy_requests && Clock::now() < hold_until) g_co.cv.wait_until(lk, hold_until);
        for (;;) {
          const auto t = Clock::now();
          if (t >= deadline) break;
          int pending_counted = 0;
          for (auto* r : g_co.pending) pending_counted += r->d->counted ? 1 : 0;
          const bool joiners = g_co.armed > pending_counted;                                  // instances that exist and have not asked yet
          const auto quiet_at = g_co.last_arrival + std::chrono::microseconds(g_co.quiet_us);
          const bool quiet = t >= quiet_at;
          if (!joiners && (!overlapping || quiet)) break;
          g_co.cv.wait_until(lk, joiners ? deadline : std::min(deadline, quiet_at));
        }
        if ((long)g_co.pending.size() <= g_co.max_set) take.swap(g_co.pending);
        else {   // the leader's own request first, then the oldest ones; the rest elect the next leader
          take.push_back(&req);
          for (auto* r : g_co.pending) if (r != &req && (long)take.size() < g_co.max_set) take.push_back(r);
          g_co.pending.erase(std::remove_if(g_co.pending.begin(), g_co.pending.end(), [&](DecodeRequest* r) { return std::find(take.begin(), take.end(), r) != take.end(); }),
                             g_co.pending.end());
        }
        for (auto* r : take) { r->taken = true; uncount(r->d); }
        g_co.in_flight += (int)take.size();
        g_co.sets_in_flight++;
        // a lone burst (nothing else in flight, no other instance waiting to decode) takes the whole CABAC pool; while other
        // sets run or are about to, every set takes 1 / max_sets - a set that grabbed everything beside later ones oversubscribes the wave
        // slots (measured: 2 x 4096 waves lose 20 % against 2 x 3584)
        const uint32_t wave_share = (g_co.sets_in_flight == 1 && g_co.armed == 0 && g_co.pending.empty()) ? 1u : (uint32_t)g_co.max_sets;
        counted_in_flight = true;
        g_co.collecting = false;
        g_co.cv.notify_all();
        lk.unlock();
        run_requests(take, wave_share);
        lk.lock();
      } catch (...) {
        // (bad_alloc in the group vectors / batch construction): no follower may be left waiting on a request this leader took, and the
        // pointer to this frame's request must not stay queued
        if (!lk.owns_lock()) lk.lock();
        g_co.collecting = false;
        g_co.pending.erase(std::remove(g_co.pending.begin(), g_co.pending.end(), &req), g_co.pending.end());
        if (counted_in_flight) { g_co.in_flight -= (int)take.size(); g_co.sets_in_flight--; }
        for (auto* r : take)
          if (!r->done) {
            if (!r->rc && !r->d->batch) { r->rc = HIPDEC_ERR_MEMORY; r->err = "decode: out of memory while building a shared launch set"; }
            r->done = true;
          }
        uncount(d);
        g_co.cv.notify_all();
        throw;
      }
      g_co.in_flight -= (int)take.size();
      g_co.sets_in_flight--;
      for (auto* r : take) r->done = true;
      g_co.cv.notify_all();
    }
  }
  if (req.rc) return set_error(req.rc, "%s", req.err.c_str());
  d->decoded = true;
  d->sq.first.clear(); d->sq.first.shrink_to_fit();
  if (info) *info = d->batch->pics[d->item].info;
  return 0;
}

void hipdec_decoder_coalesce_stats(uint64_t* requests, uint64_t* launch_sets, uint64_t* shared_requests)
{
  std::lock_guard<std::mutex> lock(g_co.mu);
  if (requests) *requests = g_co.n_requests;
  if (launch_sets) *launch_sets = g_co.n_launch_sets;
  if (shared_requests) *shared_requests = g_co.n_shared;
}

int hipdec_decoder_read_plane(hipdec_decoder* d, int c, void* dst, size_t dst_stride)
{
  if (!d || (!d->decoded && !d->out.batch)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: nothing decoded");
  hipdec_batch* b = d->plane_batch();
  const int item = d->plane_item();
  if (b && c >= 0 && c <= 2 && dst && item < (int)b->host_items.size() && b->host_items[(size_t)item].p) {   // staged by the launch set
    const PicParams& P = b->params[(size_t)item];
    if (c > 0 && !P.chroma_format_idc) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: monochrome image has no chroma planes");
    const size_t w = (size_t)(c ? P.out_cwidth : P.out_width) * (b->wide ? 2 : 1), h = (size_t)(c ? P.out_cheight : P.out_height);
    const uint8_t* src = (const uint8_t*)b->host_items[(size_t)item].p + b->host_items[(size_t)item].off[c];
    if (dst_stride < w) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: dst_stride %zu is smaller than a row of %zu bytes", dst_stride, w);
    if (dst_stride == w) memcpy(dst, src, w * h);
    else for (size_t y = 0; y < h; y++) memcpy((uint8_t*)dst + y * dst_stride, src + y * w, w);
    return 0;
  }
  return hipdec_batch_read_plane(b, item, c, dst, dst_stride);
}

int hipdec_decoder_device_plane(hipdec_decoder* d, int c, const void** dptr, size_t* stride)
{
  if (!d || (!d->decoded && !d->out.batch)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "device_plane: nothing decoded");
  return hipdec_batch_device_plane(d->plane_batch(), d->plane_item(), c, dptr, stride);
}

void hipdec_decoder_set_user_data(hipdec_decoder* d, uintptr_t user_data)
{
  if (!d) return;
  d->sq.set_user_data(user_data);   // of the sample(s) the last push brought (push_data2's argument)
}

// ---- chains ------------------------------------------------------------------------------------------------------------------------------
// The queued samples [0, n) of a track as ONE chain (batch_layout.h): parsed against the track's sequence state one after the other on the host, one
// CABAC launch and one residual launch over all of them, the pixel stages step by step.  Tracks that are decoded side by side (one decoder
// instance and one host thread each, as libheif's Track_Visual objects are) ask within a few milliseconds of each other - they were served by the
// same launch set a moment ago - so, like the still-image requests above, their chains are gathered by the first one that asks and run as ONE
// launch set: step k holds the k-th step of every track, and the small wavefront kernels (k_motion: one wave per CTB row) of 16 tracks become
// one launch of 16 x as many waves instead of 16 launches that mostly run one after the other (profiles/r05_sequence_fps.txt).
namespace {

struct ChainRequest {
  hipdec_decoder* d = nullptr;
  size_t n = 0;
  int device = 0;
  int rc = 0;
  std::string err;
  bool taken = false, done = false;
  bool oom = false;                      // rc is a failed device allocation: the same samples may fit as a shorter chain
  std::shared_ptr<hipdec_batch> batch;   // the launch set that decoded the samples ...
  int track = 0;                         // ... and which of its tracks they are
};

struct ChainCoalescer {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<ChainRequest*> pending;
  bool collecting = false;
  std::vector<hipdec_decoder*> members;   // live sequence decoders
  int in_flight = 0;                      // chains inside running launch sets
  Clock::time_point last_overlap{};       // when a request last arrived while another one was waiting or running: the host decodes tracks side by side
  long window_us = -1;                    // how long a leader waits for tracks that were active a moment ago and are idle now (HIPDEC_CHAIN_WINDOW_US; 0: every
                                          // chain on its own); for tracks whose chain is inside a running launch set it waits up to flight_us: a track that fell
                                          // out of step would otherwise run beside the others' set for good (a chain alone takes as long as 16 together)
  long flight_us = 250000;
  long max_pictures = 1024;               // samples per launch set at most (HIPDEC_CHAIN_MAX_PICTURES): bounds the arena (~7 MB per 720p picture)
  uint64_t n_sets = 0, n_shared_sets = 0, n_chains = 0;
} g_chains;

long chain_window_us()
{
  static const bool once = [] {      // (read once, thread-safely: see coalesce_window_us)
    const char* e = std::getenv("HIPDEC_CHAIN_WINDOW_US");
    if (const char* q = std::getenv("HIPDEC_CHAIN_FLIGHT_US")) g_chains.flight_us = std::max(0L, std::atol(q));
    if (const char* q = std::getenv("HIPDEC_CHAIN_MAX_PICTURES")) g_chains.max_pictures = std::max(1L, std::atol(q));
    g_chains.window_us = e ? std::max(0L, std::atol(e)) : 20000;
    return true;
  }();
  (void)once;
  return g_chains.window_us;
}

// one launch set for the chains of `group` (all on one device, with one security limit); false: it could not be built or failed on the device
bool run_chain_set(std::vector<ChainRequest*>& group)
{
  std::vector<const void*> ptrs;
  std::vector<size_t> sizes;
  std::vector<int> first, count;
  std::vector<const SeqContext*> seqs;
  for (ChainRequest* r : group) {
    first.push_back((int)ptrs.size()); count.push_back((int)r->n); seqs.push_back(&r->d->seq);
    for (size_t i = 0; i < r->n; i++) { ptrs.push_back(r->d->sq.queue[i].blob.data()); sizes.push_back(r->d->sq.queue[i].blob.size()); }
  }
  DeviceScope scope(group[0]->device);
  std::shared_ptr<hipdec_batch> sp(new hipdec_batch());
  hipdec_batch& b = *sp;
  int bad = -1;
  const ChainPlan plan{(int)group.size(), first.data(), count.data(), seqs.data(), &bad};
  static const bool trace = getenv("HIPDEC_CHAIN_TRACE") != nullptr;   // dev knob: where a chain launch set's wall time goes
  (void)arena_oom_take();
  const auto t0 = Clock::now();
  int rc = build_batch(b, (int)ptrs.size(), ptrs.data(), sizes.data(), group[0]->d->max_pixels, nullptr, nullptr, &plan);
  const auto t1 = Clock::now();
  auto t2 = t1;
  if (!rc && !b.pics.empty()) {
    hipStream_t s = stream_acquire();
    rc = hipdec_batch_run(&b, (void*)s);
    if (!rc) rc = stage_planes_to_host(b, follow_stream(&b, (void*)s));
    t2 = Clock::now();
    if (!rc) rc = hipdec_batch_status(&b);   // synchronises
    else (void)hipStreamSynchronize(s);
    b.last_stream = nullptr;
    stream_release(s);
  }
  if (trace) {
    auto ms = [](Clock::time_point a, Clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    fprintf(stderr, "[hipdec] chain set: %zu tracks, %zu pictures, %zu pixel / %zu motion steps: build %.1f ms, enqueue %.1f ms, wait %.1f ms, rc %d\n", group.size(),
            b.pics.size(), b.pixel_steps.size(), b.motion_steps.size(), ms(t0, t1), ms(t1, t2), ms(t2, Clock::now()), rc);
  }
  if (rc) {
    const bool oom = arena_oom_take() || rc == HIPDEC_ERR_MEMORY;
    if (group.size() > 1) return false;   // every track on its own then: the one with the bad sample alone gets the error
    group[0]->rc = rc; group[0]->err = hipdec_last_error(); group[0]->oom = oom;
    return true;
  }
  for (size_t t = 0; t < group.size(); t++) { group[t]->rc = 0; group[t]->batch = sp; group[t]->track = (int)t; }
  return true;
}

void run_chain_requests(std::vector<ChainRequest*>& take)
{
  std::vector<bool> used(take.size(), false);
  for (size_t i = 0; i < take.size(); i++) {
    if (used[i]) continue;
    std::vector<ChainRequest*> group;   // security limits are per instance, arenas per device: only equal ones share a launch set
    for (size_t j = i; j < take.size(); j++)
      if (!used[j] && take[j]->d->max_pixels == take[i]->d->max_pixels && take[j]->device == take[i]->device) { used[j] = true; group.push_back(take[j]); }
    bool ok = false;
    try { ok = run_chain_set(group); } catch (...) { ok = false; }
    if (!ok)
      for (ChainRequest* r : group) {
        std::vector<ChainRequest*> one{r};
        try { (void)run_chain_set(one); } catch (const std::exception& e) { r->rc = HIPDEC_ERR_MEMORY; r->err = std::string("decode: ") + e.what(); r->oom = true; }
      }
    std::lock_guard<std::mutex> lock(g_chains.mu);
    g_chains.n_sets += ok ? 1 : group.size();
    if (ok && group.size() > 1) g_chains.n_shared_sets++;
  }
}

void chain_member_add(hipdec_decoder* d)
{
  std::lock_guard<std::mutex> lock(g_chains.mu);
  if (std::find(g_chains.members.begin(), g_chains.members.end(), d) == g_chains.members.end()) g_chains.members.push_back(d);
  d->chain_active = Clock::now();
}
void chain_member_remove(hipdec_decoder* d)
{
  std::lock_guard<std::mutex> lock(g_chains.mu);
  g_chains.members.erase(std::remove(g_chains.members.begin(), g_chains.members.end(), d), g_chains.members.end());
}
void g_chains_notify() { g_chains.cv.notify_all(); }

}  // namespace

extern "C" void hipdec_decoder_chain_stats(uint64_t* chains, uint64_t* launch_sets, uint64_t* shared_launch_sets)
{
  std::lock_guard<std::mutex> lock(g_chains.mu);
  if (chains) *chains = g_chains.n_chains;
  if (launch_sets) *launch_sets = g_chains.n_sets;
  if (shared_launch_sets) *shared_launch_sets = g_chains.n_shared_sets;
}

// Afterwards the track's state sits behind the last of the samples, the DPB holds name the launch set for its pictures, and `outputs` lists the
// decoded pictures in decoding order.
static int decode_chain_once(hipdec_decoder* d, size_t n, std::vector<hipdec_decoder::Output>* outputs, bool* retry_shorter);
// The chain is bounded in BYTES as well as in pictures: a launch set holds every picture of the chain with its coefficients, unit
// maps, motion field and an uncropped copy - about 10 bytes per luma pixel -, so 32 pictures of an 8K track are > 10 GB.  n is cut to what a quarter
// of the free device memory holds (at least one picture), and a chain whose arena still cannot be allocated is retried at half the length down to
// one picture instead of dropping its samples: a large track decodes as it did in the one-sample-per-poll form, only slower.
static int decode_chain(hipdec_decoder* d, size_t n, std::vector<hipdec_decoder::Output>* outputs)
{
  if (n > d->sq.queue.size()) n = d->sq.queue.size();
  if (!n) return 0;
  if (n > 1 && d->batch && !d->batch->pics.empty()) {
    const hipdec_image_info& ii = d->batch->pics[(size_t)d->item < d->batch->pics.size() ? (size_t)d->item : 0].info;
    const double per_picture = 10.0 * (double)ii.width * (double)ii.height + double(1 << 20);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = size_t(8) << 30; }
    const size_t fit = (size_t)std::max(1.0, (double)free_b / 4.0 / per_picture);
    if (n > fit) n = fit;
  }
  for (;;) {
    bool retry = false;
    const int rc = decode_chain_once(d, n, outputs, &retry);
    if (!retry) return rc;
    n = (n + 1) / 2;
  }
}
static int decode_chain_once(hipdec_decoder* d, size_t n, std::vector<hipdec_decoder::Output>* outputs, bool* retry_shorter)
{
  if (int rc = ensure_init()) return rc;
  ChainRequest req;
  req.d = d; req.n = n; req.device = active_device();
  const long window = chain_window_us();
  {
    std::unique_lock<std::mutex> lk(g_chains.mu);
    g_chains.n_chains++;
    d->chain_active = Clock::now();
    if (!g_chains.pending.empty() || g_chains.in_flight > 0) g_chains.last_overlap = d->chain_active;
    g_chains.pending.push_back(&req);
    g_chains.cv.notify_all();
    while (!req.done) {
      if (req.taken || g_chains.collecting) { g_chains.cv.wait(lk); continue; }
      // leader: wait for the tracks that were served a moment ago (they are reading their pictures out and pushing the next samples), then run
      // everything that asked as one launch set.  A host that decodes its tracks one after the other (requests never overlap) never waits.
      g_chains.collecting = true;
      std::vector<ChainRequest*> take;
      try {
        const auto t0 = Clock::now();
        const bool side_by_side = window > 0 && g_chains.last_overlap.time_since_epoch().count() != 0 && t0 - g_chains.last_overlap < std::chrono::seconds(1);
        const auto idle_deadline = t0 + std::chrono::microseconds(window), flight_deadline = t0 + std::chrono::microseconds(std::max(window, g_chains.flight_us));
        while (side_by_side) {
          const auto t = Clock::now();
          bool idle_joiners = false, flying_joiners = false;
          for (hipdec_decoder* m : g_chains.members) {
            if (t - m->chain_active > std::chrono::seconds(1)) continue;
            bool asked = false;
            for (ChainRequest* r : g_chains.pending) if (r->d == m) asked = true;
            if (asked) continue;
            if (m->chain_in_flight) flying_joiners = true; else idle_joiners = true;
          }
          const bool wait_idle = idle_joiners && t < idle_deadline, wait_flying = flying_joiners && t < flight_deadline;
          if (!wait_idle && !wait_flying) break;
          g_chains.cv.wait_until(lk, wait_flying ? flight_deadline : idle_deadline);
        }
        long pictures = 0;
        take.push_back(&req); pictures += (long)req.n;
        for (ChainRequest* r : g_chains.pending)
          if (r != &req && pictures + (long)r->n <= g_chains.max_pictures) { take.push_back(r); pictures += (long)r->n; }
        g_chains.pending.erase(std::remove_if(g_chains.pending.begin(), g_chains.pending.end(),
                                              [&](ChainRequest* r) { return std::find(take.begin(), take.end(), r) != take.end(); }),
                               g_chains.pending.end());
        for (ChainRequest* r : take) { r->taken = true; r->d->chain_in_flight = true; }
        g_chains.in_flight += (int)take.size();
        g_chains.collecting = false;
        g_chains.cv.notify_all();
        lk.unlock();
        run_chain_requests(take);
        lk.lock();
      } catch (...) {
        if (!lk.owns_lock()) lk.lock();
        g_chains.collecting = false;
        g_chains.pending.erase(std::remove(g_chains.pending.begin(), g_chains.pending.end(), &req), g_chains.pending.end());
        for (ChainRequest* r : take)
          if (!r->done) {
            if (!r->rc && !r->batch) { r->rc = HIPDEC_ERR_MEMORY; r->err = "decode: out of memory while building a shared launch set"; }
            if (r->taken) { r->d->chain_in_flight = false; g_chains.in_flight--; }
            r->done = true;
          }
        g_chains.cv.notify_all();
        throw;
      }
      const auto now = Clock::now();
      g_chains.in_flight -= (int)take.size();
      for (ChainRequest* r : take) { r->done = true; r->d->chain_in_flight = false; r->d->chain_active = now; }
      g_chains.cv.notify_all();
    }
  }
  // (a chain that fails - a sample the front end refuses, a corrupt one - is dropped as a whole: the host gets the error once, not at every poll)
  auto drop = [&]() { d->sq.drop_front(n); };
  {
    std::lock_guard<std::mutex> lock(g_co.mu);
    g_co.n_requests += n;
    g_co.n_launch_sets++;
  }
  if (req.rc && req.oom && n > 1) { *retry_shorter = true; return req.rc; }   // out of device memory: the samples stay queued, the caller halves the chain
  if (req.rc) { drop(); return set_error(req.rc, "%s", req.err.c_str()); }
  std::shared_ptr<hipdec_batch> sp = req.batch;
  hipdec_batch& b = *sp;
  BatchLayout::ChainTrack& tr = b.tracks[(size_t)req.track];
  // the sequence state moves behind the chain; its pictures' memory is the launch set's arena
  std::vector<int> own;
  if (!tr.items.empty()) chain_resolve(b, (uint64_t)(uintptr_t)b.arena, own, req.track);
  std::vector<hipdec_decoder::DpbHold> holds;
  for (const RefPicture& rp : tr.seq_after.dpb) {
    if (std::find(own.begin(), own.end(), rp.poc) != own.end()) {
      hipdec_decoder::DpbHold h; h.poc = rp.poc; h.keep = sp; h.device = b.device;
      holds.push_back(std::move(h));
      continue;
    }
    for (auto& h : d->dpb) if (h.poc == rp.poc && (h.keep || h.full)) { holds.push_back(std::move(h)); h = hipdec_decoder::DpbHold{}; break; }
  }
  for (auto& h : d->dpb) if (h.full) { DeviceScope scope(h.device); arena_release(h.full, h.full_capacity); }
  d->dpb.swap(holds);
  d->seq = tr.seq_after;
  for (size_t k = 0; k < tr.items.size(); k++) {
    const int i = tr.items[k];
    const ParsedPicture& pp = b.pics[(size_t)i];
    if (pp.is_idr) d->cvs++;   // POCs start over: everything still waiting precedes this picture in output order
    if (!outputs) continue;
    hipdec_decoder::Output o;
    o.batch = sp; o.item = i; o.poc = pp.poc; o.cvs = d->cvs; o.user_data = d->sq.queue[(size_t)tr.samples[k]].user_data; o.pic_output = pp.pic_output;
    outputs->push_back(std::move(o));
  }
  drop();
  return 0;
}

// decode_next_image2 with output order (heif_plugin.h:164; decoder_libde265.cc:386-457 around de265_get_next_picture): decodes what is pending - the
// first picture at once, later samples once HIPDEC_SEQ_LOOKAHEAD of them wait or the host flushed - then hands out the next picture in OUTPUT order
// if the bumping process (C.5.2.2) releases one.  *have = 0: push the next sample.
int hipdec_decoder_next_picture(hipdec_decoder* d, int flush, hipdec_image_info* info, int* have, uintptr_t* user_data)
{
  if (!d || !have) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "next_picture: bad arguments");
  *have = 0;
  return guarded("next_picture", [&]() -> int {
    if (!d->decoded && !d->sq.first.empty() && d->sq.first_has_vcl) {
      hipdec_image_info ii;
      if (int rc = decoder_decode_impl(d, &ii)) return rc;
      const ParsedPicture& pp = d->batch->pics[(size_t)d->item];
      d->cvs++;
      hipdec_decoder::Output o;
      o.batch = d->batch; o.item = d->item; o.poc = pp.poc; o.cvs = d->cvs; o.user_data = d->sq.first_user_data; o.pic_output = pp.pic_output;
      if (o.pic_output) d->waiting.push_back(std::move(o));
    }
    auto release = [&](bool force) -> bool {   // the bumping process: the waiting picture that is first in output order, if it may go
      if (d->waiting.empty()) return false;
      size_t first = 0, in_cvs = 0;
      for (size_t i = 0; i < d->waiting.size(); i++) {
        const auto& a = d->waiting[i]; const auto& f = d->waiting[first];
        if (a.cvs < f.cvs || (a.cvs == f.cvs && a.poc < f.poc)) first = i;
        if (a.cvs == d->cvs) in_cvs++;
      }
      const auto& f = d->waiting[first];
      const ParsedPicture& fp = f.batch->pics[(size_t)f.item];
      // DPB fullness (C.5.2.2): pictures waiting for output plus the reference pictures that are not among them
      size_t fullness = in_cvs;
      for (const RefPicture& rp : d->seq.dpb) {
        bool waits = false;
        for (const auto& w : d->waiting) if (w.cvs == d->cvs && w.poc == rp.poc) waits = true;
        if (!waits) fullness++;
      }
      if (!(force || f.cvs < d->cvs || (int)in_cvs > fp.max_num_reorder || (int)fullness > fp.max_dec_pic_buffering)) return false;
      d->out = d->waiting[first];
      d->waiting.erase(d->waiting.begin() + (long)first);
      return true;
    };
    // pictures already decoded go out first; only when none may go are the queued samples decoded - the whole look-ahead window as one chain
    bool got = release(false);
    while (!got) {
      while (!d->sq.queue.empty() && !d->sq.queue.front().has_vcl && (d->sq.queue.size() > 1 || flush)) d->sq.queue.pop_front();
      size_t ready = 0;
      for (const auto& sm : d->sq.queue) if (sm.has_vcl) ready++;
      const size_t k = (size_t)std::max(1L, seq_lookahead());
      if (!d->decoded || !ready || !(flush || ready >= k)) break;
      if (!d->seq_active) { if (int rc = seq_activate(d)) return rc; }
      std::vector<hipdec_decoder::Output> outs;
      // only the leading run of samples that hold a coded picture: a sample of parameter sets / AUD / SEI only (say, pushed behind the last picture
      // before a flush) has no picture to parse and would fail the whole chain; it is skipped at the front of the next round
      size_t lead = 0;
      while (lead < d->sq.queue.size() && lead < k && d->sq.queue[lead].has_vcl) lead++;
      if (int rc = decode_chain(d, lead, &outs)) return rc;
      for (auto& o : outs) if (o.pic_output) d->waiting.push_back(std::move(o));
      got = release(false);
    }
    if (!got && flush && d->sq.queue.empty()) got = release(true);
    if (!got) return 0;
    if (info) { if (int rc = hipdec_batch_info(d->out.batch.get(), d->out.item, info)) return rc; }
    if (user_data) *user_data = d->out.user_data;
    *have = 1;
    return 0;
  });
}

}  // extern "C"

// ---- colour boundary: device-resident planes, the planner (a8) and the image-level conversion ---------------------------
//
// libheif converts a decoded image through its own ColorConversionPipeline (libheif/color-conversion/colorconversion.cc:
// 279-623): a Dijkstra search over the registered ColorConversionOperations, then the chosen chain on host planes.  The
// integration op (libheif_amd/integration/colorconversion_hip.cc, registered in init_ops() with SpeedCosts_Hardware) hands
// the whole conversion to hipdec_color_convert(): the planner below restates the decisions of that search for the in-scope
// states, the fused HIP kernels of color.hip execute the chain, and when the input planes are the ones a
// hipdec decoder has just copied into libheif's image, the kernels read the DEVICE copy instead of uploading them again.
namespace {

struct ResidentPlane {
  const void* host = nullptr;      // where hipdec_decoder_read_plane_tracked() copied the plane
  size_t host_stride = 0;
  int w = 0, h = 0, bits = 0;
  uint64_t sample = 0;             // hash of the host copy at hand-over time: EVERY byte (plane_hash), or - mode 2 - every 16th row
  int mode = 1;                    // the tracking mode the hash was taken in
  std::shared_ptr<hipdec_batch> batch;   // a decoder's output plane: (batch, item, comp) ...
  int item = 0, comp = 0;
  std::shared_ptr<void> buffer;           // ... or a plane of a device buffer this entry keeps alive (transform result, grid canvas)
  const uint8_t* dev = nullptr;
  size_t dev_stride = 0;
  int device = 0;
  uint64_t tick = 0;
  std::chrono::steady_clock::time_point born{};
};
std::mutex g_res_mu;
// (heap-allocated and never destroyed: its entries own batches, and destroying those from a static destructor at process exit would
//  run after the HIP runtime and this library's pools are gone; hipdec_shutdown() / the plugin's deinit empty it in good time)
std::unordered_map<const void*, ResidentPlane>& g_resident = *new std::unordered_map<const void*, ResidentPlane>();   // by host plane address
uint64_t g_res_tick = 0;
std::atomic<uint64_t> g_cb_conversions{0}, g_cb_resident{0}, g_cb_launches{0}, g_xf_transforms{0}, g_grid_canvases{0};

// Content identity of a host plane: EVERY byte of every row (64-bit lanes, four independent multiply-rotate chains, ~10 GB/s on one
// host core).  libheif edits decoded planes in place between the plugin's hand-over and the colour conversion (mirror_inplace,
// image_item.cc:969: same pointer, stride and size), so a sampled hash is not an identity — round 2 sampled 16 rows x 96 bytes and could
// pair a mirrored luma plane with the un-mirrored device chroma when the samples happened to be symmetric.
uint64_t plane_hash(const uint8_t* p, size_t stride, int w_bytes, int h)
{
  const uint64_t K1 = 0x9E3779B185EBCA87ull, K2 = 0xC2B2AE3D27D4EB4Full;
  uint64_t a = K1, b = K2, c = K1 ^ K2, d = ~K1;
  auto rot = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
  for (int y = 0; y < h; y++) {
    const uint8_t* q = p + (size_t)y * stride;
    int i = 0;
    for (; i + 32 <= w_bytes; i += 32) {
      uint64_t w[4];
      memcpy(w, q + i, 32);
      a = rot(a ^ (w[0] * K2), 31) * K1; b = rot(b ^ (w[1] * K2), 29) * K1; c = rot(c ^ (w[2] * K2), 27) * K1; d = rot(d ^ (w[3] * K2), 33) * K1;
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    if (i < w_bytes) {
      memcpy(tail, q + i, (size_t)(w_bytes - i));
      a = rot(a ^ (tail[0] * K2), 31) * K1; b = rot(b ^ (tail[1] * K2), 29) * K1; c = rot(c ^ (tail[2] * K2), 27) * K1; d = rot(d ^ (tail[3] * K2), 33) * K1;
    }
    a ^= (uint64_t)y * K2;   // the row index: swapped rows hash differently
  }
  uint64_t x = a ^ rot(b, 17) ^ rot(c, 33) ^ rot(d, 49);
  x ^= x >> 29; x *= K1; x ^= x >> 32;
  return x;
}

// The identity of tracking mode 2 (the host announces in-place edits through hipdec_forget_plane - the patched libheif of libheif_amd/integration
// does, its only in-place edit between hand-over and conversion being the mirror fall-back): the first and the last row and every 16th one.  Round 5
// measured the full hash at 38 ms per heif_decode_image() call under 256 threads (it reads 12 MB per 4K image twice, hand-over and conversion).
uint64_t plane_hash_mode(const uint8_t* p, size_t stride, int w_bytes, int h, int mode)
{
  if (mode != 2 || h <= 32) return plane_hash(p, stride, w_bytes, h);
  uint64_t x = plane_hash(p, stride * 16, w_bytes, (h + 15) / 16);
  return x ^ (plane_hash(p + (size_t)(h - 1) * stride, stride, w_bytes, 1) * 0x9E3779B185EBCA87ull);
}

// Tracking costs a pass over every decoded plane, so it only runs once a colour conversion has actually arrived at this library (the
// stock libheif never calls hipdec_color_convert: no hashing there); an entry serves ONE conversion and is dropped.  What the registry
// pins is bounded in TIME: libheif converts a decoded image within milliseconds of receiving its planes (same thread, same call), so an entry
// older than kResidentTtlMs (3 s) is stale - the host kept the planes without converting them - and goes at the next insert.  (Rounds 2 - 3 bounded it
// to 6 entries: with hundreds of application threads between read_plane and conversion the entries evicted each other and every conversion
// uploaded its planes again from pageable memory - 0.4 - 1.0 Gpixel/s of RGB through libheif where the planes alone ran at 3.)
// 0 off; 1 on, identity = a hash over every byte; 2 on for a host that ANNOUNCES its in-place edits (hipdec_forget_plane): identity = a hash over
// every 16th row - the safety net behind the announcements, not the identity itself
std::atomic<int> g_track_planes{getenv("HIPDEC_TRACK_PLANES") ? atoi(getenv("HIPDEC_TRACK_PLANES")) : 0};
const long kResidentTtlMs = getenv("HIPDEC_RESIDENT_TTL_MS") ? atol(getenv("HIPDEC_RESIDENT_TTL_MS")) : 3000;
constexpr size_t kMaxResident = 16384;

void resident_insert(struct ResidentPlane&& r);
}  // namespace
extern "C" void hipdec_forget_resident_planes(void);
namespace {

// Device rows -> a host buffer the caller owns (libheif's image planes: pageable memory).  A direct hipMemcpy2DAsync to pageable memory is
// staged by the runtime behind a process-wide lock at a few GB/s - with hundreds of application threads converting at once that was the whole
// drop-in RGB throughput (0.4 Gpixel/s at 1024 threads) - so the copy goes to pinned staging at link speed and the calling thread moves the rows
// itself (the threads do that in parallel).  Synchronises the stream.
// ... and it waits for its stream with a blocking event instead of hipStreamSynchronize's spin: the kernels of an image-level call queue up
// behind the resident CABAC pools of the decoder's launch sets (hundreds of ms under load), and a thousand application threads spinning
// through that starve the threads that feed the decoder (measured through libheif, 1024 threads x 4K stills to RGB: 0.36 Gpixel/s with
// launch sets of 10 stills).  (Admitting only a few callers at a time was tried and is worse - 0.17: each of them still waits for a
// pool to drain, so the waits have to overlap.)
hipError_t wait_stream_blocking(hipStream_t s)
{
  struct Ev { hipEvent_t e = nullptr; int device = -1; ~Ev() { if (e) (void)hipEventDestroy(e); } };
  static thread_local Ev ev;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (ev.e && ev.device != dev) { (void)hipEventDestroy(ev.e); ev.e = nullptr; }
  if (!ev.e) {
    if (hipEventCreateWithFlags(&ev.e, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev.e = nullptr; return hipStreamSynchronize(s); }
    ev.device = dev;
  }
  hipError_t e = hipEventRecord(ev.e, s);
  return e == hipSuccess ? hipEventSynchronize(ev.e) : e;
}

// HIPDEC_IMAGE_OPS_TIMING=1: where the host time of hipdec_color_convert goes, summed over all calls and printed at exit (development aid)
struct OpsTiming {
  std::atomic<uint64_t> calls{0}, find_us{0}, gpu_us{0}, copy_us{0}, hits{0}, misses{0};
  bool on = getenv("HIPDEC_IMAGE_OPS_TIMING") != nullptr;
  ~OpsTiming()
  {
    if (on && calls.load())
      fprintf(stderr, "[hipdec] color_convert: %llu calls; per call: locate planes %.2f ms (resident %llu, uploaded %llu), device work until synced %.2f ms, rows to the caller %.2f ms\n",
              (unsigned long long)calls.load(), find_us.load() / 1e3 / calls.load(), (unsigned long long)hits.load(), (unsigned long long)misses.load(),
              gpu_us.load() / 1e3 / calls.load(), copy_us.load() / 1e3 / calls.load());
  }
} g_ops_timing;
inline uint64_t us_since(Clock::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count(); }

int copy_rows_to_host(void* dst, size_t dst_stride, const void* dsrc, size_t src_stride, size_t row_bytes, int rows, hipStream_t s)
{
  void* pin = nullptr; size_t cap = 0;
  const size_t bytes = row_bytes * (size_t)rows;
  if (bytes >= (64u << 10) && pinned_acquire(&pin, bytes, &cap) == hipSuccess) {
    hipError_t e = hipMemcpy2DAsync(pin, row_bytes, dsrc, src_stride, row_bytes, (size_t)rows, hipMemcpyDeviceToHost, s);
    const auto t_wait = Clock::now();
    if (e == hipSuccess) e = wait_stream_blocking(s);
    if (g_ops_timing.on) g_ops_timing.gpu_us += us_since(t_wait);
    if (e != hipSuccess) { pinned_release(pin, cap); return set_error(HIPDEC_ERR_DEVICE, "copy to host: %s", hipGetErrorString(e)); }
    const auto t_copy = Clock::now();
    if (dst_stride == row_bytes) memcpy(dst, pin, bytes);
    else for (int y = 0; y < rows; y++) memcpy((uint8_t*)dst + (size_t)y * dst_stride, (const uint8_t*)pin + (size_t)y * row_bytes, row_bytes);
    pinned_release(pin, cap);
    if (g_ops_timing.on) g_ops_timing.copy_us += us_since(t_copy);
    return 0;
  }
  (void)hipGetLastError();
  HIPDEC_CHECK_HIP(hipMemcpy2DAsync(dst, dst_stride, dsrc, src_stride, row_bytes, (size_t)rows, hipMemcpyDeviceToHost, s));
  HIPDEC_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

void resident_note(hipdec_decoder* d, int c, const void* host, size_t stride)
{
  if (!g_track_planes.load(std::memory_order_relaxed)) return;
  hipdec_batch* pb = d->plane_batch();
  const int pitem = d->plane_item();
  const PicParams& P = pb->params[pitem];
  ResidentPlane r;
  r.host = host; r.host_stride = stride;
  r.w = c ? P.out_cwidth : P.out_width; r.h = c ? P.out_cheight : P.out_height;
  r.bits = c ? P.bit_depth_chroma : P.bit_depth_luma;
  r.mode = g_track_planes.load(std::memory_order_relaxed);
  r.sample = plane_hash_mode((const uint8_t*)host, stride, r.w * (pb->wide ? 2 : 1), r.h, r.mode);
  r.batch = d->out.batch ? d->out.batch : d->batch; r.item = pitem; r.comp = c;
  resident_insert(std::move(r));
}

// What an entry pins in HBM: a decoder's plane keeps the whole arena of its launch set alive, a buffer entry its transform result / grid canvas.
// Entries of one holder are counted once (the three planes of an image; up to 256 images of one launch set).
const void* resident_holder(const ResidentPlane& r) { return r.batch ? (const void*)r.batch.get() : (const void*)r.buffer.get(); }
size_t resident_holder_bytes(const ResidentPlane& r)
{
  if (r.batch) return r.batch->arena_capacity;
  return (size_t)r.dev_stride * (size_t)(r.h > 0 ? r.h : 1) * 2;   // (a canvas / transform result: its planes, roughly)
}
std::unordered_map<const void*, std::pair<int, size_t>>& g_res_holders = *new std::unordered_map<const void*, std::pair<int, size_t>>();
size_t g_res_bytes = 0;
uint64_t g_res_ops = 0;
// the registry is bounded by what it PINS, not only by age - a host on the patched libheif that never converts colour never consumes
// entries.  Default: an eighth of the device's memory (HIPDEC_RESIDENT_MAX_BYTES overrides; 0 = no tracking at all).
size_t resident_byte_cap()
{
  static const size_t cap = [] {
    if (const char* e = getenv("HIPDEC_RESIDENT_MAX_BYTES")) return (size_t)strtoull(e, nullptr, 10);
    size_t free_b = 0, total = 0;
    if (hipMemGetInfo(&free_b, &total) != hipSuccess) { (void)hipGetLastError(); total = size_t(64) << 30; }
    return total / 8;
  }();
  return cap;
}
void resident_account(const ResidentPlane& r, int sign)   // g_res_mu held
{
  const void* h = resident_holder(r);
  if (!h) return;
  if (sign > 0) {
    auto& e = g_res_holders[h];
    if (e.first++ == 0) { e.second = resident_holder_bytes(r); g_res_bytes += e.second; }
  } else {
    auto it = g_res_holders.find(h);
    if (it != g_res_holders.end() && --it->second.first <= 0) { g_res_bytes -= std::min(g_res_bytes, it->second.second); g_res_holders.erase(it); }
  }
}
// g_res_mu held: entries older than the time to live, and - while the registry pins more than its byte cap or holds too many entries - the oldest ones
void resident_sweep(std::vector<ResidentPlane>& dropped, bool all_expired_only)
{
  const auto now = Clock::now();
  const auto limit = now - std::chrono::milliseconds(kResidentTtlMs);
  for (auto e = g_resident.begin(); e != g_resident.end();)
    if (e->second.born <= limit) { resident_account(e->second, -1); dropped.push_back(std::move(e->second)); e = g_resident.erase(e); } else ++e;
  if (all_expired_only) return;
  const size_t cap = resident_byte_cap();
  if (g_res_bytes <= cap && g_resident.size() < kMaxResident) return;
  std::vector<std::pair<uint64_t, const void*>> by_age;
  for (auto& e : g_resident) by_age.emplace_back(e.second.tick, e.first);
  std::sort(by_age.begin(), by_age.end());
  for (size_t i = 0; i < by_age.size() && (g_res_bytes > cap / 2 || g_resident.size() >= kMaxResident / 2); i++) {   // (down to half: not again at the next insert)
    auto it = g_resident.find(by_age[i].second);
    resident_account(it->second, -1);
    dropped.push_back(std::move(it->second));
    g_resident.erase(it);
  }
}

void resident_insert(ResidentPlane&& r)
{
  if (resident_byte_cap() == 0) return;
  static const bool hooked = [] { set_memory_pressure_handler(hipdec_forget_resident_planes); return true; }();   // a failing device allocation empties the registry
  (void)hooked;
  std::vector<ResidentPlane> dropped;   // what they keep alive dies outside the lock
  std::lock_guard<std::mutex> lock(g_res_mu);
  r.tick = ++g_res_tick;
  r.born = Clock::now();
  auto it = g_resident.find(r.host);
  if (it != g_resident.end()) { resident_account(it->second, -1); dropped.push_back(std::move(it->second)); g_resident.erase(it); }
  resident_account(r, +1);
  const void* key = r.host;
  g_resident.emplace(key, std::move(r));
  if ((++g_res_ops & 15u) == 0 || g_res_bytes > resident_byte_cap() || g_resident.size() >= kMaxResident) resident_sweep(dropped, false);
}

// a host plane that was just filled from a plane of `buffer` (w x h samples of `bits`, on the current device): a transform's result or the grid canvas
void resident_note_buffer(const void* host, size_t stride, int w, int h, int bits, const uint8_t* dev, size_t dev_stride, std::shared_ptr<void> buffer)
{
  if (!g_track_planes.load(std::memory_order_relaxed)) return;
  ResidentPlane r;
  r.host = host; r.host_stride = stride; r.w = w; r.h = h; r.bits = bits;
  r.mode = g_track_planes.load(std::memory_order_relaxed);
  r.sample = plane_hash_mode((const uint8_t*)host, stride, w * (bits > 8 ? 2 : 1), h, r.mode);
  r.buffer = std::move(buffer); r.dev = dev; r.dev_stride = dev_stride;
  (void)hipGetDevice(&r.device);
  resident_insert(std::move(r));
}

// device copy of a host plane handed over by a decoder of this library, if the host plane still holds exactly those bytes; the entry
// is consumed either way
bool resident_find(const void* host, size_t stride, int w, int h, int bits, const uint8_t** dev, size_t* dev_stride, std::shared_ptr<void>& keep,
                   hipdec_batch** from_batch = nullptr, int* from_item = nullptr)
{
  if (from_batch) *from_batch = nullptr;
  if (!g_track_planes.load(std::memory_order_relaxed)) g_track_planes.store(1, std::memory_order_relaxed);
  ResidentPlane r;
  {
    std::vector<ResidentPlane> dropped;
    std::unique_lock<std::mutex> lock(g_res_mu);
    if ((++g_res_ops & 15u) == 0) resident_sweep(dropped, true);   // (expired entries go from here too: a host that only converts still ages the registry)
    auto it = g_resident.find(host);
    const bool found = it != g_resident.end();
    if (found) { resident_account(it->second, -1); r = std::move(it->second); g_resident.erase(it); }
    lock.unlock();
    dropped.clear();
    if (!found) return false;
  }
  if (r.host_stride != stride || r.w != w || r.h != h || r.bits != bits) return false;
  if (r.batch && (r.batch->retired || !r.batch->arena)) return false;
  if (plane_hash_mode((const uint8_t*)host, stride, w * (bits > 8 ? 2 : 1), h, r.mode) != r.sample) return false;
  if (!r.batch) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != r.device) return false;
    *dev = r.dev; *dev_stride = r.dev_stride; keep = r.buffer;
    return true;
  }
  const PicParams& P = r.batch->params[r.item];
  *dev = r.batch->arena + P.off_out[r.comp];
  *dev_stride = P.out_stride[r.comp];
  keep = r.batch;
  if (from_batch) { *from_batch = r.batch.get(); if (from_item) *from_item = r.item; }
  return true;
}

}  // namespace

extern "C" {

void hipdec_set_plane_tracking(int on) { g_track_planes.store(on < 0 ? 0 : (on > 2 ? 2 : on), std::memory_order_relaxed); }

// the host is about to edit (or has freed) the plane at `host_plane`: its device copy must not serve a later conversion
void hipdec_forget_plane(const void* host_plane)
{
  ResidentPlane r;   // (dies outside the lock: an entry may own a batch)
  std::lock_guard<std::mutex> lock(g_res_mu);
  auto it = g_resident.find(host_plane);
  if (it == g_resident.end()) return;
  resident_account(it->second, -1);
  r = std::move(it->second);
  g_resident.erase(it);
}

void hipdec_forget_resident_planes(void)
{
  std::unordered_map<const void*, ResidentPlane> drop;
  {
    std::lock_guard<std::mutex> lock(g_res_mu);
    drop.swap(g_resident);
    g_res_holders.clear();
    g_res_bytes = 0;
  }
}   // the batches die here, outside the lock

void hipdec_resident_rgb_stats(uint64_t* images_produced, uint64_t* conversions_served)
{
  if (images_produced) *images_produced = g_rgb_produced.load();
  if (conversions_served) *conversions_served = g_rgb_served.load();
}

void hipdec_resident_plane_stats(uint64_t* entries, uint64_t* pinned_bytes)
{
  std::lock_guard<std::mutex> lock(g_res_mu);
  if (entries) *entries = g_resident.size();
  if (pinned_bytes) *pinned_bytes = g_res_bytes;
}

int hipdec_decoder_read_plane_tracked(hipdec_decoder* d, int c, void* dst, size_t dst_stride)
{
  int rc = hipdec_decoder_read_plane(d, c, dst, dst_stride);
  if (rc) return rc;
  return guarded("read_plane", [&]() -> int { resident_note(d, c, dst, dst_stride); return 0; });
}

void hipdec_color_boundary_stats(uint64_t* conversions, uint64_t* resident_planes, uint64_t* kernel_launches)
{
  if (conversions) *conversions = g_cb_conversions.load();
  if (resident_planes) *resident_planes = g_cb_resident.load();
  if (kernel_launches) *kernel_launches = g_cb_launches.load();
}

// a8 for the in-scope states: which chain ColorConversionPipeline::construct_pipeline (colorconversion.cc:279-435) ends up with.
// Every stock op costs SpeedCosts_Unoptimized, so the search returns the chain with the fewest steps; the rules below are the
// state_after_conversion() conditions of those ops (yuv2rgb.cc:35-92, :298-341, :430-478, :566-620; chroma_sampling.cc:501-560;
// hdr_sdr.cc:146-190; rgb2rgb.cc:30-70).  nclx "unspecified" (2) counts as the sRGB defaults for planning (nclx.cc:360-373).
int hipdec_color_plan(int bit_depth, int chroma, int has_alpha, const hipdec_nclx* nclx, int out_chroma, int upsampling, int only_preferred,
                      int ops[8], int* n_ops)
{
  if (!ops || !n_ops) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "color_plan: bad arguments");
  *n_ops = 0;
  int matrix = 6, full = 1;
  if (nclx && nclx->has_nclx) { matrix = nclx->matrix_coefficients == 2 ? 6 : nclx->matrix_coefficients; full = nclx->full_range_flag; }
  if (matrix == 11 || matrix == 14) return set_error(HIPDEC_ERR_UNSUPPORTED, "Unsupported color conversion (matrix_coefficients %d), as in the reference", matrix);
  if (chroma == 0) {   // heif_chroma_monochrome: Op_mono_to_RGB24_32 (8-bit; RGB24 only without an alpha plane)
    if (bit_depth != 8 || !(out_chroma == 11 || (out_chroma == 10 && !has_alpha)))
      return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: this monochrome conversion is left to the stock ops");
    ops[(*n_ops)++] = HIPDEC_OP_MONO_TO_RGB24_32;
    return 0;
  }
  if (chroma < 1 || chroma > 3) return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: input chroma %d is outside the HEIC hot path", chroma);
  const bool nn_allowed = !(only_preferred && upsampling != 1);
  auto push = [&](int op) { ops[(*n_ops)++] = op; };
  if (out_chroma == 10 || out_chroma == 11) {          // interleaved RGB / RGBA, 8 bit
    if (out_chroma == 10 && has_alpha) return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: dropping an alpha plane is left to the stock ops");
    if (bit_depth > 8) {
      // > 8-bit planes to 8-bit interleaved RGB: the search of the reference ends on one of two chains (checked state by state against the compiled
      // pipeline, tests/test_color_emu.py): Op_to_sdr_planes FIRST when the 8-bit chain behind it is shorter (the 4:2:0 integer op) or when the
      // preferred upsampling has to run anyway; otherwise the generic op at the input depth, THEN Op_to_sdr_planes on R, G, B
      if (has_alpha) return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: > 8-bit planes with alpha to 8-bit RGB are left to the stock ops");
      const bool int_op = chroma == 1 && nn_allowed && full && matrix != 0 && matrix != 8;
      if (!int_op && !(chroma != 3 && !nn_allowed)) { push(HIPDEC_OP_YCBCR_TO_RGB); push(HIPDEC_OP_TO_SDR); push(HIPDEC_OP_RGB_TO_RGB24_32); return 0; }
      push(HIPDEC_OP_TO_SDR); bit_depth = 8;
    }
    if (chroma == 1 && nn_allowed && full && matrix != 0 && matrix != 8) { push(out_chroma == 10 ? HIPDEC_OP_420_TO_RGB24 : HIPDEC_OP_420_TO_RGB32); return 0; }
    if (chroma != 3 && !nn_allowed) {
      push(chroma == 1 ? HIPDEC_OP_BILINEAR_420_TO_444 : HIPDEC_OP_BILINEAR_422_TO_444);
    }
    push(HIPDEC_OP_YCBCR_TO_RGB); push(HIPDEC_OP_RGB_TO_RGB24_32);
    return 0;
  }
  if (out_chroma == 12 || out_chroma == 14) {          // RRGGBB big / little endian
    if (has_alpha) return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: RRGGBB from an image with alpha is left to the stock ops");
    if (bit_depth <= 8) return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: 8-bit to RRGGBB needs Op_to_hdr_planes, outside the hot path");
    if (chroma == 1 && nn_allowed && matrix != 0 && matrix != 8) { push(HIPDEC_OP_420_TO_RRGGBB); return 0; }
    // every other state: the generic float op on the 16-bit planes (after the preferred upsampling when nearest neighbour is ruled out), then the
    // interleave, then the swap for little endian
    if (chroma != 3 && !nn_allowed) push(chroma == 1 ? HIPDEC_OP_BILINEAR_420_TO_444 : HIPDEC_OP_BILINEAR_422_TO_444);
    push(HIPDEC_OP_YCBCR_TO_RGB); push(HIPDEC_OP_RGB_HDR_TO_RRGGBB_BE);
    if (out_chroma == 14) push(HIPDEC_OP_SWAP_ENDIANNESS);
    return 0;
  }
  return set_error(HIPDEC_ERR_UNSUPPORTED, "color_plan: output chroma %d is outside the HEIC hot path", out_chroma);
}

int hipdec_color_convert(const hipdec_color_image* in, const hipdec_nclx* nclx, int out_chroma, int upsampling, int only_preferred,
                         void* out, size_t out_stride, int out_on_device)
{
  if (!in || !out || in->width <= 0 || in->height <= 0 || !in->plane[0] || (in->chroma != 0 && (!in->plane[1] || !in->plane[2])))
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "color_convert: bad arguments");
  if (int rc = ensure_init()) return rc;
  return guarded("color_convert", [&]() -> int {
    int ops[8], n_ops = 0;
    const bool has_alpha = in->plane[3] != nullptr;
    if (int rc = hipdec_color_plan(in->bit_depth, in->chroma, has_alpha, nclx, out_chroma, upsampling, only_preferred, ops, &n_ops)) return rc;
    const int w = in->width, h = in->height;
    const int cw = in->chroma == 3 ? w : (w + 1) / 2, ch = in->chroma == 1 ? (h + 1) / 2 : h;
    int bits = in->bit_depth;
    size_t es = bits > 8 ? 2 : 1;
    const size_t out_bpp = out_chroma == 10 ? 3 : (out_chroma == 11 ? 4 : 6);
    if ((out_chroma == 10 || out_chroma == 11 || out_chroma == 12 || out_chroma == 14) && out_stride < (size_t)w * out_bpp)
      return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "color_convert: out_stride %zu is smaller than a row of %zu bytes", out_stride, (size_t)w * out_bpp);
    for (int c = 0; c < 4; c++)
      if (in->plane[c] && in->stride[c] < (size_t)((c == 0 || c == 3) ? w : cw) * es)
        return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "color_convert: stride %zu of plane %d is smaller than its rows", in->stride[c], c);
    hipStream_t s = stream_acquire_priority();
    struct Release { hipStream_t s; std::vector<std::pair<void*, size_t>> bufs;
                     ~Release() { (void)hipStreamSynchronize(s); for (auto& b : bufs) arena_release(b.first, b.second); stream_release(s); } } rel{s, {}};
    auto scratch = [&](size_t bytes, uint8_t** p) -> int {
      void* d = nullptr; size_t cap = 0;
      HIPDEC_CHECK_HIP(arena_acquire(&d, bytes ? bytes : 256, &cap));
      rel.bufs.emplace_back(d, cap); *p = (uint8_t*)d;
      return 0;
    };
    // ---- the input planes on the device: the decoder's own copy when the host planes are still the ones it handed over
    const uint8_t* dp[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t ds[4] = {0, 0, 0, 0};
    std::shared_ptr<void> keep[4];
    const auto t_find = Clock::now();
    if (g_ops_timing.on) g_ops_timing.calls++;
    // 8-bit 4:2:0 planes to interleaved RGB24 in ONE op: what a launch set of the decoder path can produce beside the planes (resident RGB).  Asking
    // for it earns the credit that makes the next launch sets carry it.
    // (the planner lists Op_YCbCr_to_RGB + Op_RGB_to_RGB24_32 as two ops; they run as one pass here and in the fused kernel)
    const bool rgb24_single = out_chroma == 10 && !has_alpha && in->chroma == 1 && in->bit_depth == 8 && !out_on_device && !in->on_device &&
                              ((n_ops == 1 && ops[0] == HIPDEC_OP_420_TO_RGB24) ||
                               (n_ops == 2 && ops[0] == HIPDEC_OP_YCBCR_TO_RGB && ops[1] == HIPDEC_OP_RGB_TO_RGB24_32));
    if (rgb24_single) rgb_note_wanted();
    hipdec_batch* from_b[4] = {nullptr, nullptr, nullptr, nullptr};
    int from_i[4] = {-1, -1, -1, -1};
    for (int c = 0; c < 4; c++) {
      if (!in->plane[c]) continue;
      const int pw = (c == 0 || c == 3) ? w : cw, ph = (c == 0 || c == 3) ? h : ch;
      if (in->on_device) { dp[c] = (const uint8_t*)in->plane[c]; ds[c] = in->stride[c]; continue; }
      if (resident_find(in->plane[c], in->stride[c], pw, ph, bits, &dp[c], &ds[c], keep[c], &from_b[c], &from_i[c])) {
        g_cb_resident++; if (g_ops_timing.on) g_ops_timing.hits++;
        if (c == 2 && rgb24_single && from_b[0] && from_b[0] == from_b[1] && from_b[1] == from_b[2] && from_i[0] == from_i[1] && from_i[1] == from_i[2]) {
          // all three planes are the untouched output of one decoded picture (resident_find compared every byte's hash): if its launch set emitted
          // RGB24 with exactly this colour description and the op the planner chose, the rows are already in pinned host memory
          hipdec_batch* rb = from_b[0];
          const size_t item = (size_t)from_i[0];
          if (item < rb->host_items.size() && rb->host_items[item].rgb && item < rb->pics.size()) {
            const hipdec_image_info& I = rb->pics[item].info;
            const int m = I.matrix_coeffs == 2 ? 6 : I.matrix_coeffs;
            const bool fused_int = I.full_range_flag && m != 0 && m != 8;
            if (nclx && nclx->has_nclx && nclx->colour_primaries == I.colour_primaries && nclx->transfer_characteristics == I.transfer_characteristics &&
                nclx->matrix_coefficients == I.matrix_coeffs && nclx->full_range_flag == I.full_range_flag && fused_int == (ops[0] == HIPDEC_OP_420_TO_RGB24) &&
                rb->params[item].out_width == w && rb->params[item].out_height == h) {
              const auto t_copy = Clock::now();
              const uint8_t* src = rb->host_items[item].rgb;
              const size_t row = (size_t)w * 3;
              if (out_stride == row) memcpy(out, src, row * (size_t)h);
              else for (int y = 0; y < h; y++) memcpy((uint8_t*)out + (size_t)y * out_stride, src + (size_t)y * row, row);
              rb->rgb_consumed++;
              g_rgb_served++;
              g_cb_conversions++;
              if (g_ops_timing.on) { g_ops_timing.find_us += us_since(t_find); g_ops_timing.copy_us += us_since(t_copy); }
              return 0;
            }
          }
        }
        continue;
      }
      if (g_ops_timing.on) g_ops_timing.misses++;
      uint8_t* d = nullptr;
      const size_t st = ((size_t)pw * es + 255) & ~(size_t)255;
      if (int rc = scratch(st * ph, &d)) return rc;
      HIPDEC_CHECK_HIP(hipMemcpy2DAsync(d, st, in->plane[c], in->stride[c], (size_t)pw * es, ph, hipMemcpyHostToDevice, s));
      dp[c] = d; ds[c] = st;
    }
    if (g_ops_timing.on) g_ops_timing.find_us += us_since(t_find);
    // ---- the chain
    int chroma = in->chroma, k = 0;
    if (k < n_ops && ops[k] == HIPDEC_OP_TO_SDR) {                     // a14 on every plane
      for (int c = 0; c < 4; c++) {
        if (!dp[c]) continue;
        const int pw = (c == 0 || c == 3) ? w : cw, ph = (c == 0 || c == 3) ? h : ch;
        uint8_t* d = nullptr;
        const size_t st = ((size_t)pw + 255) & ~(size_t)255;
        if (int rc = scratch(st * ph, &d)) return rc;
        if (int rc = hipdec_color_to_sdr(dp[c], ds[c], pw, ph, bits, d, st, (void*)s)) return rc;
        g_cb_launches++;
        dp[c] = d; ds[c] = st;
      }
      bits = 8; es = 1; k++;
    }
    if (k < n_ops && ops[k] == HIPDEC_OP_BILINEAR_420_TO_444) {         // a13 on both chroma planes
      for (int c = 1; c < 3; c++) {
        uint8_t* d = nullptr;
        const size_t st = ((size_t)w * es + 255) & ~(size_t)255;
        if (int rc = scratch(st * h, &d)) return rc;
        if (int rc = hipdec_color_bilinear_420_to_444(dp[c], ds[c], w, h, bits, d, st, (void*)s)) return rc;
        g_cb_launches++;
        dp[c] = d; ds[c] = st;
      }
      chroma = 3; k++;
    }
    if (k < n_ops && ops[k] == HIPDEC_OP_BILINEAR_422_TO_444) {         // SURVEY 8 f4, on both chroma planes
      for (int c = 1; c < 3; c++) {
        uint8_t* d = nullptr;
        const size_t st = ((size_t)w * es + 255) & ~(size_t)255;
        if (int rc = scratch(st * h, &d)) return rc;
        if (int rc = hipdec_color_bilinear_422_to_444(dp[c], ds[c], w, h, bits, d, st, (void*)s)) return rc;
        g_cb_launches++;
        dp[c] = d; ds[c] = st;
      }
      chroma = 3; k++;
    }
    uint8_t* dout = (uint8_t*)out;
    size_t dout_stride = out_stride;
    if (!out_on_device) {
      dout_stride = ((size_t)w * out_bpp + 255) & ~(size_t)255;
      if (int rc = scratch(dout_stride * h, &dout)) return rc;
    }
    if (k >= n_ops) return set_error(HIPDEC_ERR_UNSUPPORTED, "color_convert: empty chain");
    // The pipeline attaches the ColorState's profile - unspecified values replaced with the sRGB defaults - to every intermediate image
    // (colorconversion.cc:475, nclx.cc:360-373): an op that is not the first of its chain reads THAT profile, not the input image's.  It matters
    // for images without (or with unspecified) matrix_coefficients: computed BT.601 coefficients instead of the rounded default constants
    hipdec_nclx later{1, 1, 13, 6, 1};
    if (nclx && nclx->has_nclx) {
      later = *nclx;
      if (later.colour_primaries == 2) later.colour_primaries = 1;
      if (later.transfer_characteristics == 2) later.transfer_characteristics = 13;
      if (later.matrix_coefficients == 2) later.matrix_coefficients = 6;
    }
    if (k > 0) nclx = &later;
    int rc = 0;
    switch (ops[k]) {
      case HIPDEC_OP_MONO_TO_RGB24_32:
        rc = hipdec_color_mono_to_rgb24(dp[0], ds[0], dp[3], ds[3], w, h, dout, dout_stride, out_chroma == 11, (void*)s); break;
      case HIPDEC_OP_420_TO_RGB24:
        rc = hipdec_color_420_to_rgb24(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, nclx, dout, dout_stride, 0, (void*)s); break;
      case HIPDEC_OP_420_TO_RGB32:
        rc = hipdec_color_420_to_rgba_alpha(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, nclx, dp[3], ds[3], 1, 1, dout, dout_stride, (void*)s); break;
      case HIPDEC_OP_YCBCR_TO_RGB:   // a10 + a11 as one pass
        if (k + 1 < n_ops && ops[k + 1] == HIPDEC_OP_TO_SDR)     // > 8-bit planes to 8-bit RGB(A): generic op at the input depth, to_sdr on R, G, B, interleave
          rc = hipdec_color_hdr_to_rgb24(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, bits, chroma, nclx, dout, dout_stride, out_chroma == 11, 0, (void*)s);
        else if (out_chroma == 12 || out_chroma == 14)
          rc = hipdec_color_ycbcr_to_rrggbb_float(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, bits, chroma, nclx, dout, dout_stride, out_chroma == 14, (void*)s);
        else if (out_chroma == 11) rc = hipdec_color_420_to_rgba_alpha(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, nclx, dp[3], ds[3], 0, chroma, dout, dout_stride, (void*)s);
        else rc = hipdec_color_ycbcr_to_rgb24_float(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, chroma, nclx, dout, dout_stride, 0, (void*)s);
        break;
      case HIPDEC_OP_420_TO_RRGGBB:
        rc = hipdec_color_420_to_rrggbb(dp[0], ds[0], dp[1], ds[1], dp[2], ds[2], w, h, bits, nclx, dout, dout_stride, out_chroma == 14, (void*)s); break;
      default: rc = set_error(HIPDEC_ERR_UNSUPPORTED, "color_convert: unplanned op %d", ops[k]); break;
    }
    if (rc) return rc;
    g_cb_launches++;
    if (!out_on_device) { if (int rc2 = copy_rows_to_host(out, out_stride, dout, dout_stride, (size_t)w * out_bpp, h, s)) return rc2; }
    else HIPDEC_CHECK_HIP(hipStreamSynchronize(s));
    g_cb_conversions++;
    return 0;
  });
}

// 'irot' / 'imir' / 'clap' of ImageItem::decode_image (libheif/image-items/image_item.cc:949-1081) over the planes of an image, on the device
// (plane kernels: transform.hip).  op HIPDEC_XF_ROTATE_CCW: args[0] = 90 / 180 / 270 (HeifPixelImage::rotate_ccw, image/pixelimage.cc:1175-1300);
// HIPDEC_XF_MIRROR: args[0] = heif_transform_mirror_direction (mirror_inplace, :1358-1430); HIPDEC_XF_CROP: args = left, right, top, bottom, the
// inclusive end points HeifPixelImage::crop takes (:1433-1530).  Where the reference first converts a subsampled image to 4:4:4 (odd sizes /
// offsets, the checks at :1187-1204, :1371-1381, :1457-1464) this returns HIPDEC_ERR_UNSUPPORTED and the caller keeps the host path.  The input
// planes are host or device pointers (host planes the decoder handed over are found device-resident); `out` brings the destination planes
// (NULL where the input has none) and receives the geometry.
int hipdec_image_transform(const hipdec_color_image* in, int op, const int* args, hipdec_color_image* out)
{
  if (!in || !out || !args || in->width <= 0 || in->height <= 0 || !in->plane[0] || in->bit_depth < 8 || in->bit_depth > 16)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: bad arguments");
  if (int rc = ensure_init()) return rc;
  return guarded("image_transform", [&]() -> int {
    const int w = in->width, h = in->height;
    const bool has_chroma = in->plane[1] && in->plane[2];
    const int chroma = has_chroma ? in->chroma : 0;
    if (has_chroma && (chroma < 1 || chroma > 3)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: chroma must be 1 (4:2:0), 2 (4:2:2) or 3 (4:4:4)");
    const bool odd_w = w & 1, odd_h = h & 1;
    int ow = w, oh = h, left = 0, top = 0;
    bool needs_444 = false;
    if (op == HIPDEC_XF_ROTATE_CCW) {
      const int a = args[0];
      if (a != 90 && a != 180 && a != 270) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: rotation must be 90, 180 or 270");
      if (chroma == 2) needs_444 = a != 180 || odd_h;
      else if (chroma == 1) needs_444 = (a == 90 && odd_w) || (a == 180 && (odd_w || odd_h)) || (a == 270 && odd_h);
      if (a != 180) { ow = h; oh = w; }
    } else if (op == HIPDEC_XF_MIRROR) {
      if (args[0] != 0 && args[0] != 1) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: mirror direction must be 0 or 1");
      if (chroma == 2) needs_444 = args[0] == 1 && odd_w;
      else if (chroma == 1) needs_444 = odd_w || odd_h;
    } else if (op == HIPDEC_XF_CROP) {
      const int l = args[0], r = args[1], t = args[2], b = args[3];
      if (l < 0 || t < 0 || r < l || b < t || r >= w || b >= h) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: invalid crop region");
      left = l; top = t; ow = r - l + 1; oh = b - t + 1;
      if (chroma == 2) needs_444 = l & 1;
      else if (chroma == 1) needs_444 = (l & 1) || (t & 1);
      // An odd window size leaves a half-covered chroma column / row at the edge.  The reference's output there is NOT the plane copy its
      // crop() reads like (measured through heif_decode_image on the compiled reference: the last chroma column / row differs from the
      // decoded plane), so those windows stay on the host rather than being claimed.
      if (!needs_444 && ((chroma == 1 || chroma == 2) && (ow & 1))) needs_444 = true;
      if (!needs_444 && chroma == 1 && (oh & 1)) needs_444 = true;
    } else return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: unknown operation %d", op);
    if (needs_444) return set_error(HIPDEC_ERR_UNSUPPORTED, "image_transform: odd size / offset of a subsampled image (the reference converts to 4:4:4 first or treats the half-covered chroma edge itself): host path");
    const size_t es = in->bit_depth > 8 ? 2 : 1;
    const int sx = (chroma == 1 || chroma == 2) ? 2 : 1, sy = chroma == 1 ? 2 : 1;
    hipStream_t s = stream_acquire_priority();
    struct Release { hipStream_t s; std::vector<std::pair<void*, size_t>> bufs;
                     ~Release() { (void)hipStreamSynchronize(s); for (auto& b : bufs) arena_release(b.first, b.second); stream_release(s); } } rel{s, {}};
    auto scratch = [&](size_t bytes, uint8_t** p) -> int {
      void* d = nullptr; size_t cap = 0;
      HIPDEC_CHECK_HIP(arena_acquire(&d, bytes ? bytes : 256, &cap));
      rel.bufs.emplace_back(d, cap); *p = (uint8_t*)d;
      return 0;
    };
    std::shared_ptr<void> keep[4];
    struct Result { int c, w, h; const uint8_t* dev; size_t dev_stride; std::shared_ptr<void> owner; };
    std::vector<Result> results;
    for (int c = 0; c < 4; c++) {
      if (!in->plane[c]) continue;
      if ((c == 1 || c == 2) && !has_chroma) continue;
      if (!out->plane[c]) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: no destination for plane %d", c);
      const bool sub = c == 1 || c == 2;
      const int pw = sub ? (w + sx - 1) / sx : w, ph = sub ? (h + sy - 1) / sy : h;
      // geometry of this plane's result (crop: HeifPixelImage::crop's plane_left .. plane_right, pixelimage.cc:1497-1500)
      int pl = left, pt = top, pow_ = ow, poh = oh;
      if (sub) {
        if (op == HIPDEC_XF_CROP) { pl = left / sx; pt = top / sy; pow_ = (left + ow - 1) / sx - pl + 1; poh = (top + oh - 1) / sy - pt + 1; }
        else if (op == HIPDEC_XF_ROTATE_CCW && args[0] != 180) { pow_ = ph; poh = pw; }
        else { pow_ = pw; poh = ph; }
      }
      if (out->stride[c] < (size_t)pow_ * es) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "image_transform: destination stride of plane %d too small", c);
      const uint8_t* dp = nullptr; size_t ds = 0;
      if (in->on_device) { dp = (const uint8_t*)in->plane[c]; ds = in->stride[c]; }
      else if (resident_find(in->plane[c], in->stride[c], pw, ph, in->bit_depth, &dp, &ds, keep[c])) g_cb_resident++;
      else {
        uint8_t* d = nullptr;
        const size_t st = ((size_t)pw * es + 255) & ~(size_t)255;
        if (int rc = scratch(st * ph, &d)) return rc;
        HIPDEC_CHECK_HIP(hipMemcpy2DAsync(d, st, in->plane[c], in->stride[c], (size_t)pw * es, ph, hipMemcpyHostToDevice, s));
        dp = d; ds = st;
      }
      uint8_t* dout = (uint8_t*)out->plane[c];
      size_t dout_stride = out->stride[c];
      if (!out->on_device) {
        dout_stride = ((size_t)pow_ * es + 255) & ~(size_t)255;
        if (g_track_planes.load(std::memory_order_relaxed)) {
          // the result stays on the device behind its host copy: the next transform or the colour conversion of this image reads it there
          void* d = nullptr; size_t cap = 0;
          HIPDEC_CHECK_HIP(arena_acquire(&d, dout_stride * poh ? dout_stride * poh : 256, &cap));
          int dev_index = 0;
          (void)hipGetDevice(&dev_index);
          std::shared_ptr<void> owner(d, [cap, dev_index](void* q) { DeviceScope scope(dev_index); arena_release(q, cap); });
          dout = (uint8_t*)d;
          results.push_back(Result{c, pow_, poh, dout, dout_stride, std::move(owner)});
        } else if (int rc = scratch(dout_stride * poh, &dout)) return rc;
      }
      int rc;
      if (op == HIPDEC_XF_ROTATE_CCW) rc = hipdec_plane_rotate_ccw(dp, ds, pw, ph, (int)es, args[0], dout, dout_stride, (void*)s);
      else if (op == HIPDEC_XF_MIRROR) rc = hipdec_plane_mirror(dp, ds, pw, ph, (int)es, args[0], dout, dout_stride, (void*)s);
      else rc = hipdec_plane_crop(dp, ds, pw, ph, (int)es, pl, pt, pow_, poh, dout, dout_stride, (void*)s);
      if (rc) return rc;
      if (!out->on_device) { if (int rc2 = copy_rows_to_host((void*)out->plane[c], out->stride[c], dout, dout_stride, (size_t)pow_ * es, poh, s)) return rc2; }
    }
    HIPDEC_CHECK_HIP(hipStreamSynchronize(s));
    for (auto& r : results) resident_note_buffer(out->plane[r.c], out->stride[r.c], r.w, r.h, in->bit_depth, r.dev, r.dev_stride, std::move(r.owner));
    out->width = ow; out->height = oh; out->chroma = in->chroma; out->bit_depth = in->bit_depth;
    g_xf_transforms++;
    return 0;
  });
}

}  // extern "C"

// ---- grid images across the GPUs of one node ------------------------------------------------------------------------------
//
// Device-side form of ImageItem_Grid::decode_full_grid_image (libheif/image-items/grid.cc:250-468: tile fan-out :405-453) and
// decode_and_paste_tile_image (:482-577, HeifPixelImage::copy_image_to image/pixelimage.cc:1115-1172), in ONE process over
// `n_devices` HIP devices:
//   * tile t = row * cols + col (the order of the 'dimg' references, grid.cc:193,319) belongs to shard t mod G; every shard
//     decodes its tiles as one batch — arena, streams and launches on its own device (DeviceScope);
//   * the one exchange step is the paste: each decoded tile plane goes straight from its device to its (x0, y0) position in the
//     canvas on the root device with one strided device-to-device copy (peer access over xGMI when the devices allow it, staged
//     by the runtime otherwise), queued on the OWNER's stream right behind the tile's decode; 1.5 bytes per pixel in total, no
//     intermediate packing and no collective — a gather whose every message lands at its final address;
//   * the root's stream waits for one event per shard, then the fused colour conversion runs once over the canvas (bilinear
//     chroma taps cross tile borders, so the colour stage sees the whole canvas).
// The Python path (libheif_amd/grid.py: one process per GPU, torch.distributed gather) stays as the multi-process test driver.
struct hipdec_grid {
  int rows = 0, cols = 0, out_w = 0, out_h = 0, tile_w = 0, tile_h = 0, bits = 8;
  int csw = 2, csh = 2;   // chroma subsampling of the tiles
  std::vector<int> devices;                       // one entry per shard; entries may repeat (several shards on one device)
  std::vector<std::unique_ptr<hipdec_batch>> shard;
  std::vector<std::vector<int>> shard_tiles;      // tile indices of shard s, in batch order
  std::vector<hipStream_t> stream;                // one per shard, on its device
  std::vector<hipEvent_t> pasted;                 // per shard: its tiles are in the canvas
  std::vector<int> transport;                     // per shard: 0 = on the root device, 1 = peer access to the root enabled (direct xGMI writes), 2 = no peer
                                                  // access (the runtime stages the copy)
  int root = 0;                                   // device of the canvas
  uint8_t* canvas = nullptr;
  size_t canvas_capacity = 0;
  size_t off[3] = {0, 0, 0}, stride[3] = {0, 0, 0};
  hipdec_image_info info{};                       // of tile 0 (colour description for the canvas)
  bool decoded = false;
  int issue_threads = 1;                          // host threads the last hipdec_grid_decode enqueued the shards from
  ~hipdec_grid()
  {
    for (size_t s = 0; s < shard.size(); s++) {
      DeviceScope scope(devices[s]);
      if (stream[s]) { (void)hipStreamSynchronize(stream[s]); stream_release(stream[s]); }
      if (pasted[s]) (void)hipEventDestroy(pasted[s]);
      shard[s].reset();
    }
    canvas_owner.reset();   // (a host plane still registered as resident keeps the buffer until its entry is consumed)
  }
  std::shared_ptr<void> canvas_owner;             // the canvas buffer; shared with the resident-plane registry (hipdec_grid_read_plane_tracked)
};

extern "C" {

int hipdec_grid_create(hipdec_grid** out, int rows, int cols, int out_width, int out_height, const void* const* tile_data,
                       const size_t* tile_sizes, const int* devices, int n_devices, uint64_t max_image_size_pixels)
{
  if (!out || rows <= 0 || cols <= 0 || rows > 256 || cols > 256 || out_width <= 0 || out_height <= 0 || !tile_data || !tile_sizes)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_create: bad arguments");
  *out = nullptr;
  if (int rc = ensure_init()) return rc;
  return guarded("grid_create", [&]() -> int {
    const int n_tiles = rows * cols;
    int visible = 0;
    HIPDEC_CHECK_HIP(hipGetDeviceCount(&visible));
    std::unique_ptr<hipdec_grid> g(new hipdec_grid());
    if (devices && n_devices > 0) g->devices.assign(devices, devices + n_devices);
    else { const int n = n_devices > 0 ? n_devices : visible; for (int d = 0; d < n; d++) g->devices.push_back(d % visible); }
    if ((int)g->devices.size() > n_tiles) g->devices.resize((size_t)n_tiles);
    for (int d : g->devices)
      if (d < 0 || d >= visible || d >= max_devices())
        return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_create: device %d is not usable (%d devices visible, at most %d supported)", d, visible, max_devices());
    const int G = (int)g->devices.size();
    g->rows = rows; g->cols = cols; g->out_w = out_width; g->out_h = out_height; g->root = g->devices[0];
    g->shard.resize((size_t)G); g->shard_tiles.resize((size_t)G); g->stream.assign((size_t)G, nullptr); g->pasted.assign((size_t)G, nullptr); g->transport.assign((size_t)G, 0);
    for (int t = 0; t < n_tiles; t++) g->shard_tiles[(size_t)(t % G)].push_back(t);
    for (int s = 0; s < G; s++) {
      DeviceScope scope(g->devices[s]);
      if (g->devices[s] != g->root) {   // direct peer copies into the canvas where the topology allows them
        int can = 0;
        g->transport[(size_t)s] = 2;
        if (hipDeviceCanAccessPeer(&can, g->devices[s], g->root) == hipSuccess && can) {
          const hipError_t pe = hipDeviceEnablePeerAccess(g->root, 0);
          if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) g->transport[(size_t)s] = 1;
          (void)hipGetLastError();
        }
      }
      std::vector<const void*> ptrs;
      std::vector<size_t> sizes;
      for (int t : g->shard_tiles[(size_t)s]) { ptrs.push_back(tile_data[t]); sizes.push_back(tile_sizes[t]); }
      hipdec_batch* b = nullptr;
      if (int rc = hipdec_batch_create(&b, (int)ptrs.size(), ptrs.data(), sizes.data(), max_image_size_pixels)) return rc;
      g->shard[(size_t)s].reset(b);
      g->stream[(size_t)s] = stream_acquire();
      HIPDEC_CHECK_HIP(hipEventCreateWithFlags(&g->pasted[(size_t)s], hipEventDisableTiming));
    }
    // every tile has the size / bit depth / chroma format of tile 0 and the output fits the tiled area (grid.cc:270-282)
    g->info = g->shard[0]->pics[0].info;
    g->tile_w = g->info.width; g->tile_h = g->info.height; g->bits = g->info.bit_depth_luma;
    for (int s = 0; s < G; s++)
      for (const auto& p : g->shard[(size_t)s]->pics)
        if (p.info.width != g->tile_w || p.info.height != g->tile_h || p.info.bit_depth_luma != g->bits || p.info.chroma_format_idc != g->info.chroma_format_idc)
          return set_error(HIPDEC_ERR_BITSTREAM, "grid_create: tiles differ in size, bit depth or chroma format");
    if (out_width > cols * g->tile_w || out_height > rows * g->tile_h)
      return set_error(HIPDEC_ERR_BITSTREAM, "grid_create: the output size exceeds the tiled area");
    g->csw = (g->info.chroma_format_idc == 1 || g->info.chroma_format_idc == 2) ? 2 : 1;   // SubWidthC / SubHeightC of the tiles (and of the canvas)
    g->csh = g->info.chroma_format_idc == 1 ? 2 : 1;
    if (g->info.chroma_format_idc && ((g->tile_w % g->csw) || (g->tile_h % g->csh))) return set_error(HIPDEC_ERR_UNSUPPORTED, "grid_create: subsampled tiles with odd dimensions");
    {
      DeviceScope scope(g->root);
      const size_t es = g->bits > 8 ? 2 : 1;
      const size_t cw = g->info.chroma_format_idc ? (size_t)(out_width + g->csw - 1) / g->csw : 0, ch = g->info.chroma_format_idc ? (size_t)(out_height + g->csh - 1) / g->csh : 0;
      size_t o = 0;
      g->stride[0] = ((size_t)out_width * es + 255) & ~(size_t)255; g->off[0] = o; o += g->stride[0] * (size_t)out_height;
      g->stride[1] = g->stride[2] = (cw * es + 255) & ~(size_t)255;
      g->off[1] = o; o += g->stride[1] * ch; g->off[2] = o; o += g->stride[2] * ch;
      HIPDEC_CHECK_HIP(arena_acquire((void**)&g->canvas, o ? o : 256, &g->canvas_capacity));
      {
        const size_t cap = g->canvas_capacity;
        const int root = g->root;
        g->canvas_owner = std::shared_ptr<void>(g->canvas, [cap, root](void* q) { DeviceScope scope(root); arena_release(q, cap); });
      }
    }
    *out = g.release();
    return 0;
  });
}

void hipdec_grid_free(hipdec_grid* g) { delete g; }

int hipdec_grid_info(const hipdec_grid* g, hipdec_image_info* info, int* n_shards)
{
  if (!g || !info) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_info: bad arguments");
  *info = g->info;
  info->width = g->out_w; info->height = g->out_h;
  info->chroma_width = g->info.chroma_format_idc ? (g->out_w + g->csw - 1) / g->csw : 0; info->chroma_height = g->info.chroma_format_idc ? (g->out_h + g->csh - 1) / g->csh : 0;
  info->coded_width = g->cols * g->tile_w; info->coded_height = g->rows * g->tile_h;
  size_t bytes = 0; int subs = 0;
  for (const auto& b : g->shard) for (const auto& p : b->pics) { bytes += p.info.bitstream_bytes; subs += p.info.num_substreams; }
  info->bitstream_bytes = bytes; info->num_substreams = subs;
  if (n_shards) *n_shards = (int)g->shard.size();
  return 0;
}

int hipdec_grid_transport(const hipdec_grid* g, int* local_shards, int* peer_shards, int* staged_shards)
{
  if (!g) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_transport: NULL grid");
  int n[3] = {0, 0, 0};
  for (int t : g->transport) n[t < 0 || t > 2 ? 2 : t]++;
  if (local_shards) *local_shards = n[0];
  if (peer_shards) *peer_shards = n[1];
  if (staged_shards) *staged_shards = n[2];
  return 0;
}

int hipdec_grid_decode(hipdec_grid* g)
{
  if (!g) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_decode: NULL grid");
  return guarded("grid_decode", [&]() -> int {
    const size_t es = g->bits > 8 ? 2 : 1;
    const int ncomp = g->info.chroma_format_idc ? 3 : 1;
    auto issue = [&](size_t s) -> int {
      DeviceScope scope(g->devices[s]);
      hipdec_batch* b = g->shard[s].get();
      if (int rc = hipdec_batch_run(b, (void*)g->stream[s])) return rc;
      (void)follow_stream(b, (void*)g->stream[s]);   // with stage overlap the pixel stages run on the post stream: the pastes below wait for them
      for (size_t i = 0; i < g->shard_tiles[s].size(); i++) {
        const int t = g->shard_tiles[s][i];
        const int x0 = (t % g->cols) * g->tile_w, y0 = (t / g->cols) * g->tile_h;
        const int w = std::min(g->tile_w, g->out_w - x0), h = std::min(g->tile_h, g->out_h - y0);   // clipped to the output (pixelimage.cc:1130-1160)
        if (w <= 0 || h <= 0) continue;
        const PicParams& P = b->params[i];
        for (int c = 0; c < ncomp; c++) {
          const size_t sw = c ? (size_t)g->csw : 1, sh = c ? (size_t)g->csh : 1;
          const size_t pw = ((size_t)w + sw - 1) / sw, ph = ((size_t)h + sh - 1) / sh;
          const size_t px = (size_t)x0 / sw, py = (size_t)y0 / sh;
          HIPDEC_CHECK_HIP(hipMemcpy2DAsync(g->canvas + g->off[c] + py * g->stride[c] + px * es, g->stride[c], b->arena + P.off_out[c], P.out_stride[c],
                                            pw * es, ph, hipMemcpyDefault, g->stream[s]));   // (kind from the pointers: the canvas may sit on another device)
        }
      }
      b->mark_done(g->stream[s]);
      HIPDEC_CHECK_HIP(hipEventRecord(g->pasted[s], g->stream[s]));
      return 0;
    };
    // One host thread per device (ADVICE / VERDICT round 4): a shard's launch set - upload wait, six kernels, its tiles' pastes - is enqueued by its
    // own thread, so the devices start together instead of one after the other behind a single thread's launch overhead.  (A thread's error text
    // is thread-local: it is carried back to the caller.)
    if (g->shard.size() <= 1 || getenv("HIPDEC_GRID_SERIAL_ISSUE")) {
      for (size_t s = 0; s < g->shard.size(); s++) if (int rc = issue(s)) return rc;
    } else {
      std::vector<int> rcs(g->shard.size(), 0);
      std::vector<std::string> msgs(g->shard.size());
      std::vector<std::thread> th;
      for (size_t s = 1; s < g->shard.size(); s++)
        th.emplace_back([&, s]() { try { rcs[s] = issue(s); } catch (...) { rcs[s] = set_error(HIPDEC_ERR_MEMORY, "grid_decode: shard %zu: host failure", s); } if (rcs[s]) msgs[s] = hipdec_last_error(); });
      rcs[0] = issue(0);
      if (rcs[0]) msgs[0] = hipdec_last_error();
      for (auto& t : th) t.join();
      g->issue_threads = (int)g->shard.size();
      for (size_t s = 0; s < g->shard.size(); s++) if (rcs[s]) return set_error(rcs[s], "%s", msgs[s].c_str());
    }
    {
      DeviceScope scope(g->root);   // whatever the caller queues on the root's stream next sees the whole canvas
      for (size_t s = 0; s < g->shard.size(); s++) HIPDEC_CHECK_HIP(hipStreamWaitEvent(default_stream(), g->pasted[s], 0));
    }
    g->decoded = true;
    return 0;
  });
}

int hipdec_grid_wait(hipdec_grid* g)
{
  if (!g || !g->decoded) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_wait: nothing was decoded");
  for (size_t s = 0; s < g->shard.size(); s++) {
    DeviceScope scope(g->devices[s]);
    if (int rc = hipdec_batch_status(g->shard[s].get())) return rc;       // device-side decode errors are loud, per shard
    HIPDEC_CHECK_HIP(hipEventSynchronize(g->pasted[s]));
  }
  DeviceScope scope(g->root);
  HIPDEC_CHECK_HIP(hipStreamSynchronize(default_stream()));
  return 0;
}

int hipdec_grid_canvas_plane(hipdec_grid* g, int c, const void** dptr, size_t* stride, int* device)
{
  if (!g || c < 0 || c > 2 || !dptr || !stride) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_canvas_plane: bad arguments");
  if (c > 0 && !g->info.chroma_format_idc) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_canvas_plane: monochrome grid has no chroma planes");
  *dptr = g->canvas + g->off[c]; *stride = g->stride[c];
  if (device) *device = g->root;
  return 0;
}

int hipdec_grid_read_plane(hipdec_grid* g, int c, void* dst_host, size_t dst_stride)
{
  if (!g || !g->decoded || c < 0 || c > 2 || !dst_host) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_read_plane: bad arguments");
  if (int rc = hipdec_grid_wait(g)) return rc;
  DeviceScope scope(g->root);
  const size_t es = g->bits > 8 ? 2 : 1;
  const size_t sw = c ? (size_t)g->csw : 1, sh = c ? (size_t)g->csh : 1;
  const size_t w = ((size_t)g->out_w + sw - 1) / sw, h = ((size_t)g->out_h + sh - 1) / sh;
  if (dst_stride < w * es) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_read_plane: dst_stride %zu is smaller than a row of %zu bytes", dst_stride, w * es);
  return copy_rows_to_host(dst_host, dst_stride, g->canvas + g->off[c], g->stride[c], w * es, (int)h, default_stream());
}

// hipdec_grid_read_plane + the host plane registered as device-resident (the canvas stays alive behind it): the colour conversion of the
// composed image then reads the canvas where it is instead of uploading it again
int hipdec_grid_read_plane_tracked(hipdec_grid* g, int c, void* dst_host, size_t dst_stride)
{
  if (int rc = hipdec_grid_read_plane(g, c, dst_host, dst_stride)) return rc;
  return guarded("grid_read_plane", [&]() -> int {
    DeviceScope scope(g->root);
    const int sw = c ? g->csw : 1, sh = c ? g->csh : 1;
    resident_note_buffer(dst_host, dst_stride, (g->out_w + sw - 1) / sw, (g->out_h + sh - 1) / sh, g->bits, g->canvas + g->off[c], g->stride[c], g->canvas_owner);
    if (c == 0) g_grid_canvases++;
    return 0;
  });
}

void hipdec_image_ops_stats(uint64_t* transforms, uint64_t* grid_canvases)
{
  if (transforms) *transforms = g_xf_transforms.load();
  if (grid_canvases) *grid_canvases = g_grid_canvases.load();
}

int hipdec_grid_to_rgb(hipdec_grid* g, int out_chroma, int upsampling, int only_preferred, void* out, size_t out_stride, int out_on_device)
{
  if (!g || !g->decoded || !out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_to_rgb: bad arguments");
  if (!g->info.chroma_format_idc) return set_error(HIPDEC_ERR_UNSUPPORTED, "grid_to_rgb: monochrome grid");
  if (int rc = hipdec_grid_wait(g)) return rc;
  DeviceScope scope(g->root);
  hipdec_color_image img{};
  img.width = g->out_w; img.height = g->out_h; img.chroma = g->info.chroma_format_idc; img.bit_depth = g->bits; img.on_device = 1;
  for (int c = 0; c < 3; c++) { img.plane[c] = g->canvas + g->off[c]; img.stride[c] = g->stride[c]; }
  hipdec_nclx nclx{1, g->info.colour_primaries, g->info.transfer_characteristics, g->info.matrix_coeffs, g->info.full_range_flag};
  return hipdec_color_convert(&img, &nclx, out_chroma, upsampling, only_preferred, out, out_stride, out_on_device);
}

}  // extern "C"
