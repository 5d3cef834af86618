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
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>


using namespace hipdec;

constexpr int kEv = 8;   // events per timing slot: start, parse, residual, recon, deblock, sao | colour begin, colour end

struct hipdec_batch : BatchLayout {
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
  std::vector<uint8_t> colour_timed;   // per slot: the colour stage of that run was recorded
  uint64_t runs = 0;
  hipStream_t last_stream = nullptr;
  bool ran = false;
  bool retired = false;         // its arena went to another batch (hipdec_batch_create_recycling): only status / timing / free remain
  ColorBatchState color;        // parameter blocks of hipdec_batch_to_rgb_all
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
    if (arena || staging) (void)wait();   // nothing of this batch may still be running when the arena is recycled
    release_staging();
    color_batch_state_free(color);
    if (arena) arena_release(arena, arena_capacity);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
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
int build_batch(hipdec_batch& b, int n, const void* const* data, const size_t* sizes, uint64_t max_pixels, hipdec_batch* recycle = nullptr)
{
  std::string err;
  int rc = layout_batch_plan(b, n, data, sizes, max_pixels, err);
  if (rc != HIPDEC_OK) return set_error(rc, "%s", err.c_str());
  HIPDEC_CHECK_HIP(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
  b.host_status = status_slot_acquire();
  if (!b.host_status) return set_error(HIPDEC_ERR_MEMORY, "batch_create: more than 4096 live batches (no pinned status slot left)");
  *b.host_status = 0;
  // a retired batch of the same shape hands its arena over (hipdec_batch_create_recycling): the upload is ordered behind
  // everything that batch still has in flight, so one arena serves a stream of batches
  hipEvent_t after = nullptr;
  if (recycle && recycle->arena && recycle->arena_capacity >= b.arena_size) {
    b.arena = recycle->arena; b.arena_capacity = recycle->arena_capacity;
    recycle->arena = nullptr; recycle->arena_capacity = 0;
    if (recycle->done_recorded) after = recycle->done;
    recycle->retired = true;
  }
  static const bool sync_upload = getenv("HIPDEC_SYNC_UPLOAD") != nullptr;   // profiling knob: one stream, no cross-stream event waits
  if (b.upload_size > (size_t(4) << 20) && !sync_upload) {
    HIPDEC_CHECK_HIP(pinned_acquire(&b.staging, b.upload_size, &b.staging_capacity));
    layout_batch_fill(b, data, sizes, (uint8_t*)b.staging);
    if (!b.arena) HIPDEC_CHECK_HIP(arena_acquire((void**)&b.arena, b.arena_size, &b.arena_capacity));
    HIPDEC_CHECK_HIP(hipEventCreateWithFlags(&b.uploaded, hipEventDisableTiming));
    hipStream_t us = upload_stream();
    if (after) HIPDEC_CHECK_HIP(hipStreamWaitEvent(us, after, 0));
    HIPDEC_CHECK_HIP(hipMemcpyAsync(b.arena, b.staging, b.upload_size, hipMemcpyHostToDevice, us));
    HIPDEC_CHECK_HIP(hipEventRecord(b.uploaded, us));
  } else {
    std::vector<uint8_t> host(b.upload_size);
    layout_batch_fill(b, data, sizes, host.data());
    if (!b.arena) HIPDEC_CHECK_HIP(arena_acquire((void**)&b.arena, b.arena_size, &b.arena_capacity));
    if (after) HIPDEC_CHECK_HIP(hipEventSynchronize(after));
    HIPDEC_CHECK_HIP(hipMemcpy(b.arena, host.data(), b.upload_size, hipMemcpyHostToDevice));
  }
  b.ev.assign(kEv, nullptr);
  b.colour_timed.assign(1, 0);
  for (auto& e : b.ev) HIPDEC_CHECK_HIP(hipEventCreate(&e));
  return 0;
}

int launch_all(hipdec_batch& b, hipStream_t s)
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
    uint32_t waves = getenv("HIPDEC_POOL_WAVES") ? (uint32_t)atoi(getenv("HIPDEC_POOL_WAVES")) : parse_wave_budget();
    waves = waves > b.num_subs ? b.num_subs : waves;
    pa.num_waves = waves < 1 ? 1 : waves;
  }
  // a row hands its wave back after every CTB: the ready queue then advances all rows breadth-first and rows rarely run into
  // the row above (measured: parse 698 -> 591 ms against "run until blocked")
  pa.yield_ctbs = getenv("HIPDEC_POOL_YIELD") ? (uint32_t)atoi(getenv("HIPDEC_POOL_YIELD")) : 1;
  pa.wake_hyst = getenv("HIPDEC_POOL_HYST") ? (uint32_t)atoi(getenv("HIPDEC_POOL_HYST")) : 0u;
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
  HIPDEC_CHECK_HIP(hipMemsetAsync(b.arena + b.off_ctrl, 0, b.ctrl_size, s));
  HIPDEC_CHECK_HIP(hipEventRecord(ev[0], s));
  if (int rc = step("memset")) return rc;
  launch_parse(pa, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[1], s));
  if (int rc = step("parse")) return rc;
  static const bool parse_only = getenv("HIPDEC_DEBUG_PARSE_ONLY") != nullptr;   // tuning knob: isolate the CABAC kernel
  if (!parse_only) launch_residual(fa, n, b.max_ctbs, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[2], s));
  if (int rc = step("residual")) return rc;
  if (!parse_only) launch_recon(ra, b.wide, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[3], s));
  if (int rc = step("recon")) return rc;
  if (!parse_only) launch_deblock(fa, n, b.max_w, b.max_h, b.wide, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[4], s));
  if (int rc = step("deblock")) return rc;
  if (!parse_only) launch_sao(fa, n, b.max_ow, b.max_oh, b.wide, s);
  HIPDEC_CHECK_HIP(hipEventRecord(ev[5], s));
  if (int rc = step("sao")) return rc;
  HIPDEC_CHECK_HIP(hipGetLastError());
  HIPDEC_CHECK_HIP(hipMemcpyAsync(b.host_status, b.arena + b.off_status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  b.mark_done(s);
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

int copy_plane_d2h(const hipdec_batch& b, size_t off, uint32_t stride, int w, int h, void* dst, size_t dst_stride)
{
  const size_t es = b.wide ? 2 : 1;
  if (w <= 0 || h <= 0) return 0;
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
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_run: the batch's arena was handed to another batch");
  if (int rc = ensure_init()) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  b->last_stream = s;
  b->ran = true;
  return launch_all(*b, s);
}

int hipdec_batch_status(hipdec_batch* b)
{
  if (!b || !b->ran) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "batch_status: batch has not been run");
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
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "pack_item: the batch's arena was handed to another batch");
  if (dst_bytes < hipdec_batch_item_packed_bytes(b, i)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "pack_item: destination too small");
  const PicParams& P = b->params[i];
  const size_t es = b->wide ? 2 : 1;
  hipStream_t s = stream ? (hipStream_t)stream : (b->last_stream ? b->last_stream : default_stream());
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
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_rgb: the batch's arena was handed to another batch");
  const PicParams& P = b->params[i];
  const hipdec_image_info& I = b->pics[i].info;
  if (!P.chroma_format_idc) return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: monochrome input");
  // the decoder reports the VUI colour description exactly as the libde265 plugin would attach it
  hipdec_nclx nclx{1, I.colour_primaries, I.transfer_characteristics, I.matrix_coeffs, I.full_range_flag};
  const uint8_t* y = b->arena + P.off_out[0]; const uint8_t* cb = b->arena + P.off_out[1]; const uint8_t* cr = b->arena + P.off_out[2];
  void* s = stream ? stream : (void*)b->last_stream;
  struct MarkDone {   // whatever is enqueued below belongs to this batch (hipdec_batch_status / free wait for it)
    hipdec_batch* b; hipStream_t s;
    ~MarkDone() { b->mark_done(s ? s : default_stream()); }
  } mark{b, (hipStream_t)s};
  if (out_chroma == 10 || out_chroma == 11) {
    if (b->wide) return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: 8-bit interleaved output from >8-bit planes needs hipdec_color_to_sdr first");
    // planner rule: integer op only for full range and a matrix it accepts
    const int m = I.matrix_coeffs == 2 ? 6 : I.matrix_coeffs;
    if (I.full_range_flag && m != 0 && m != 8)
      return hipdec_color_420_to_rgb24(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, &nclx, out_dev,
                                       out_stride, out_chroma == 11, s);
    return hipdec_color_ycbcr_to_rgb24_float(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, 1, &nclx,
                                             out_dev, out_stride, out_chroma == 11, s);
  }
  if (out_chroma == 12 || out_chroma == 14) {
    if (!b->wide) return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: RRGGBB output needs >8-bit planes");
    return hipdec_color_420_to_rrggbb(y, P.out_stride[0], cb, P.out_stride[1], cr, P.out_stride[2], P.out_width, P.out_height, I.bit_depth_luma,
                                      &nclx, out_dev, out_stride, out_chroma == 14, s);
  }
  return set_error(HIPDEC_ERR_UNSUPPORTED, "to_rgb: unsupported output chroma %d", out_chroma);
}

int hipdec_batch_to_rgb_all(hipdec_batch* b, int out_chroma, void* const* outs_dev, const size_t* out_strides, void* stream)
{
  if (!b || !outs_dev || !out_strides) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_rgb_all: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : (b->last_stream ? b->last_stream : default_stream());
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

int hipdec_batch_timing_slots(hipdec_batch* b, int slots)
{
  if (!b || slots < 1 || slots > 4096) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "timing_slots: bad arguments");
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
  if (b->retired) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_tap: the batch's arena was handed to another batch");
  (void)which;  // the reconstruction buffer holds the deblocked picture after a full run
  const PicParams& P = b->params[i];
  return copy_plane_d2h(*b, P.off_rec[c], P.rec_stride[c], c ? P.cwidth : P.width, c ? P.cheight : P.height, dst, dst_stride);
}

int hipdec_batch_read_maps(hipdec_batch* b, int i, uint8_t* log2_tb, uint8_t* log2_cb, uint8_t* intra_luma, uint8_t* intra_chroma, int8_t* qp_y,
                           uint8_t* flags, size_t map_elems)
{
  if (!b || i < 0 || i >= (int)b->pics.size()) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_maps: bad arguments");
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
  std::vector<uint8_t> data;
  int strict = 0;
  uint64_t max_pixels = 0;
  std::shared_ptr<hipdec_batch> batch;   // shared with the other instances decoded in the same launch
  int item = 0;                          // this instance's picture inside `batch`
  bool decoded = false;
  bool counted = false;                  // included in Coalescer::armed
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
  uint64_t n_requests = 0, n_launch_sets = 0, n_shared = 0;   // statistics (hipdec_decoder_coalesce_stats)
} g_co;

long coalesce_window_us()
{
  if (g_co.window_us < 0) {
    const char* e = std::getenv("HIPDEC_COALESCE_WINDOW_US");   // 0 disables coalescing
    g_co.window_us = e ? std::max(0L, std::atol(e)) : 2000;
    if (const char* q = std::getenv("HIPDEC_COALESCE_QUIET_US")) g_co.quiet_us = std::max(1L, std::atol(q));
  }
  return g_co.window_us;
}

// one decoder in a batch of its own: the reference behaviour, and the fallback that gives every request its own
// error when a shared batch could not be built or failed on the device
void run_single(DecodeRequest& r, hipStream_t s)
{
  hipdec_decoder* d = r.d;
  const void* ptrs[1] = {d->data.data()};
  const size_t sizes[1] = {d->data.size()};
  hipdec_batch* b = nullptr;
  r.rc = hipdec_batch_create(&b, 1, ptrs, sizes, d->max_pixels);
  if (!r.rc) {
    r.rc = hipdec_batch_run(b, (void*)s);
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

void run_group(std::vector<DecodeRequest*>& group, hipStream_t s)
{
  if (group.size() == 1) { run_single(*group[0], s); return; }
  std::vector<const void*> ptrs;
  std::vector<size_t> sizes;
  for (auto* r : group) { ptrs.push_back(r->d->data.data()); sizes.push_back(r->d->data.size()); }
  hipdec_batch* b = nullptr;
  int rc = hipdec_batch_create(&b, (int)group.size(), ptrs.data(), sizes.data(), group[0]->d->max_pixels);
  if (!rc) {
    rc = hipdec_batch_run(b, (void*)s);
    if (!rc) rc = hipdec_batch_status(b);
    else (void)hipStreamSynchronize(s);
    b->last_stream = nullptr;
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
  run_group(lo, s);
  run_group(hi, s);
}

void run_requests(std::vector<DecodeRequest*>& take)
{
  hipStream_t s = stream_acquire();   // own stream per launch set: batches of different leaders overlap on the GPU
  std::vector<bool> used(take.size(), false);
  for (size_t i = 0; i < take.size(); i++) {
    if (used[i]) continue;
    std::vector<DecodeRequest*> group;   // security limits are per instance: only equal limits share a batch
    for (size_t j = i; j < take.size(); j++)
      if (!used[j] && take[j]->d->max_pixels == take[i]->d->max_pixels) { used[j] = true; group.push_back(take[j]); }
    run_group(group, s);
  }
  stream_release(s);
}

void uncount(hipdec_decoder* d)   // g_co.mu held
{
  if (d->counted) { d->counted = false; g_co.armed--; }
}

}  // namespace

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
  delete d;
}

void hipdec_decoder_set_strict(hipdec_decoder* d, int strict) { if (d) d->strict = strict; }

int hipdec_decoder_push_data(hipdec_decoder* d, const void* data, size_t size)
{
  if (!d || (!data && size)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "push_data: bad arguments");
  if (d->decoded) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "push_data after decode (heif_plugin.h:113-115 forbids it)");
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
  return guarded("push_data", [&]() -> int { d->data.insert(d->data.end(), p, p + size); return 0; });
}

static int decoder_decode_impl(hipdec_decoder* d, hipdec_image_info* info);
int hipdec_decoder_decode(hipdec_decoder* d, hipdec_image_info* info)
{
  return guarded("decode", [&]() -> int { return decoder_decode_impl(d, info); });
}
static int decoder_decode_impl(hipdec_decoder* d, hipdec_image_info* info)
{
  if (!d) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "decode: NULL decoder");
  if (d->decoded) return set_error(HIPDEC_ERR_NO_IMAGE, "no further image");
  if (d->data.empty()) return set_error(HIPDEC_ERR_NO_IMAGE, "no data was pushed");
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
      const auto t0 = Clock::now();
      const auto deadline = t0 + std::chrono::microseconds(window);
      const bool overlapping = g_co.last_overlap.time_since_epoch().count() != 0 &&
                               t0 - g_co.last_overlap < std::chrono::milliseconds(250);
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
      std::vector<DecodeRequest*> take;
      take.swap(g_co.pending);
      for (auto* r : take) { r->taken = true; uncount(r->d); }
      g_co.in_flight += (int)take.size();
      g_co.collecting = false;
      g_co.cv.notify_all();
      lk.unlock();
      run_requests(take);
      lk.lock();
      g_co.in_flight -= (int)take.size();
      for (auto* r : take) r->done = true;
      g_co.cv.notify_all();
    }
  }
  if (req.rc) return set_error(req.rc, "%s", req.err.c_str());
  d->decoded = true;
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
  if (!d || !d->decoded) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "read_plane: nothing decoded");
  return hipdec_batch_read_plane(d->batch.get(), d->item, c, dst, dst_stride);
}

int hipdec_decoder_device_plane(hipdec_decoder* d, int c, const void** dptr, size_t* stride)
{
  if (!d || !d->decoded) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "device_plane: nothing decoded");
  return hipdec_batch_device_plane(d->batch.get(), d->item, c, dptr, stride);
}

}  // extern "C"
