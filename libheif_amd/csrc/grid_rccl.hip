// grid_rccl.hip — grid images with the tiles sharded over one PROCESS PER GPU and gathered with RCCL over xGMI.
//
// SPMD form of ImageItem_Grid::decode_full_grid_image + decode_and_paste_tile_image (libheif/image-items/grid.cc:250-468, :482-577;
// HeifPixelImage::copy_image_to, libheif/image/pixelimage.cc:1115-1172), beside the one-process form hipdec_grid_* (decoder.hip):
//   * tile t = row * cols + col (order of the 'dimg' references, grid.cc:193,319) belongs to rank t mod nranks; every rank decodes
//     its tiles as ONE hipdec batch on its own GPU;
//   * the one exchange step is a gather to rank 0: every other rank packs its decoded tiles (Y | Cb | Cr, tight rows) into one send
//     buffer, rank 0 posts one ncclRecv per peer, all inside one ncclGroupStart / ncclGroupEnd on the decode stream (ragged shards
//     need no padding; 1.5 bytes per pixel for 8-bit 4:2:0, one message per link);
//   * rank 0 pastes its own tiles straight from its batch and the received ones from the receive buffer to (col * tile_w,
//     row * tile_h), clipped to the output size, and runs the colour conversion once over the canvas.
// librccl is loaded at run time (dlopen), so the plugin has no link-time dependency on it: a host without RCCL gets
// HIPDEC_ERR_UNSUPPORTED from these entry points and everything else works.
#include "hipdec_internal.h"
// Build time needs RCCL's TYPES only (the functions come from dlopen): taken from its header where the toolchain has one, otherwise the handful this
// file uses is declared here with the values of the NCCL 2.x / RCCL ABI (opaque communicator, 128-byte unique id, enum codes).
#if defined(__has_include) && __has_include(<rccl/rccl.h>) && !defined(HIPDEC_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}
#endif
#include <dlfcn.h>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace hipdec;

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("HIPDEC_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) return;
    auto sym = [&](const char* n) { return dlsym(api.handle, n); };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv && api.GroupStart && api.GroupEnd && api.AllReduce && api.GetErrorString;
  });
  return api;
}

int need_rccl()
{
  if (!rccl().ok) return set_error(HIPDEC_ERR_UNSUPPORTED, "RCCL is not available (librccl.so.1 could not be loaded; HIPDEC_RCCL_LIBRARY names another path)");
  return 0;
}

// ncclGroupStart / ncclGroupEnd as a scope: an error return between them still closes the group, so that the peers - which are inside the same
// collective - are not left waiting for this rank's half of it
struct RcclGroup {
  bool open = false;
  ncclResult_t start() { const ncclResult_t r = rccl().GroupStart(); open = r == ncclSuccess; return r; }
  ncclResult_t end() { open = false; return rccl().GroupEnd(); }
  ~RcclGroup() { if (open) (void)rccl().GroupEnd(); }
};

#define HIPDEC_CHECK_NCCL(expr)                                                                                 \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess) return set_error(HIPDEC_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
  } while (0)

}  // namespace

struct hipdec_grid_rccl {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  int rows = 0, cols = 0, out_w = 0, out_h = 0, tile_w = 0, tile_h = 0, bits = 8, chroma = 1;
  int csw = 2, csh = 2;
  int device = 0;
  std::vector<int> mine;                     // tile indices of this rank, in batch / message order
  hipdec_batch* batch = nullptr;             // this rank's tiles (NULL: the rank owns none)
  hipStream_t stream = nullptr;
  size_t tile_bytes = 0;                     // one packed tile: Y | Cb | Cr, tight rows
  uint8_t* send = nullptr; size_t send_capacity = 0;     // ranks != 0: their packed tiles
  uint8_t* recv = nullptr; size_t recv_capacity = 0;     // rank 0: the peers' tiles, peer p's at recv_off[p]
  std::vector<size_t> recv_off;
  uint8_t* canvas = nullptr; size_t canvas_capacity = 0; // rank 0
  size_t off[3] = {0, 0, 0}, stride[3] = {0, 0, 0};
  hipdec_image_info info{};                  // of this rank's first tile (rank 0: tile 0, the canvas' colour description)
  bool decoded = false;
  bool attempted = false;                    // a decode posted this rank's part of the exchange (whatever its own result): wait() must join the status all-reduce
  int64_t* status_dev = nullptr; size_t status_capacity = 0;   // the 8 bytes of wait()'s all-reduce, held for the object's life: no allocation can fail there
  bool status_known = false; int status_rc = 0; std::string status_msg;   // wait()'s result for the last decode (the exchange runs once per decode)
  std::string queue_msg;
  int queue_rc = 0;                          // this rank could not queue its shard's decode (its tiles are undefined): reported to every rank by wait()
  ~hipdec_grid_rccl()
  {
    DeviceScope scope(device);
    if (stream) { (void)hipStreamSynchronize(stream); stream_release(stream); }
    if (batch) hipdec_batch_free(batch);
    if (send) arena_release(send, send_capacity);
    if (recv) arena_release(recv, recv_capacity);
    if (canvas) arena_release(canvas, canvas_capacity);
    if (status_dev) arena_release(status_dev, status_capacity);
  }
};

extern "C" {

int hipdec_rccl_available(void) { return rccl().ok ? 1 : 0; }

int hipdec_rccl_unique_id(void* id_out)
{
  if (!id_out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "rccl_unique_id: NULL buffer");
  if (int rc = need_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == HIPDEC_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  HIPDEC_CHECK_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int hipdec_rccl_comm_create(void** comm_out, int nranks, int rank, const void* unique_id)
{
  if (!comm_out || !unique_id || nranks <= 0 || rank < 0 || rank >= nranks) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "rccl_comm_create: bad arguments");
  *comm_out = nullptr;
  if (int rc = need_rccl()) return rc;
  if (int rc = ensure_init()) return rc;       // the communicator lives on the device hipdec_init() selected
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  HIPDEC_CHECK_NCCL(rccl().CommInitRank(&comm, nranks, id, rank));
  *comm_out = (void*)comm;
  return 0;
}

void hipdec_rccl_comm_destroy(void* comm)
{
  if (comm && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)comm);
}

int hipdec_grid_create_rccl(hipdec_grid_rccl** out, void* comm, int rank, int nranks, int rows, int cols, int out_width, int out_height,
                            const void* const* tile_data, const size_t* tile_sizes, uint64_t max_image_size_pixels)
{
  if (!out || !comm || nranks <= 0 || rank < 0 || rank >= nranks || rows <= 0 || cols <= 0 || rows > 256 || cols > 256 || out_width <= 0 || out_height <= 0 ||
      !tile_data || !tile_sizes)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_create_rccl: bad arguments");
  *out = nullptr;
  if (int rc = need_rccl()) return rc;
  if (int rc = ensure_init()) return rc;
  return guarded("grid_create_rccl", [&]() -> int {
    const int n_tiles = rows * cols;
    std::unique_ptr<hipdec_grid_rccl> g(new hipdec_grid_rccl());
    g->comm = (ncclComm_t)comm; g->rank = rank; g->nranks = nranks; g->device = active_device();
    g->rows = rows; g->cols = cols; g->out_w = out_width; g->out_h = out_height;
    for (int t = rank; t < n_tiles; t += nranks) g->mine.push_back(t);
    g->stream = stream_acquire();
    // geometry word every rank agrees on before anything is exchanged: {tile bytes, tile width, tile height, bits, chroma}; ranks without tiles
    // contribute the neutral element of both reductions
    int64_t geo[5] = {0, 0, 0, 0, 0};
    int local_rc = 0;
    if (!g->mine.empty()) {
      std::vector<const void*> ptrs;
      std::vector<size_t> sizes;
      for (int t : g->mine) {
        if (!tile_data[t] || !tile_sizes[t]) { local_rc = set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_create_rccl: rank %d owns tile %d but was given no data for it", rank, t); break; }
        ptrs.push_back(tile_data[t]); sizes.push_back(tile_sizes[t]);
      }
      if (!local_rc) local_rc = hipdec_batch_create(&g->batch, (int)ptrs.size(), ptrs.data(), sizes.data(), max_image_size_pixels);
      if (!local_rc) {
        (void)hipdec_batch_info(g->batch, 0, &g->info);
        for (int i = 0; i < (int)g->mine.size() && !local_rc; i++) {
          hipdec_image_info ii{};
          (void)hipdec_batch_info(g->batch, i, &ii);
          if (ii.width != g->info.width || ii.height != g->info.height || ii.bit_depth_luma != g->info.bit_depth_luma || ii.chroma_format_idc != g->info.chroma_format_idc)
            local_rc = set_error(HIPDEC_ERR_BITSTREAM, "grid_create_rccl: tiles differ in size, bit depth or chroma format");
        }
        geo[0] = (int64_t)hipdec_batch_item_packed_bytes(g->batch, 0);
        geo[1] = g->info.width; geo[2] = g->info.height; geo[3] = g->info.bit_depth_luma; geo[4] = g->info.chroma_format_idc;
      }
    }
    // one small all-reduce pair (max and min over {geometry, -error}) so that every rank learns whether all ranks built their shard and agree on
    // the tile geometry; a rank that failed still takes part, otherwise the others would wait for it in the gather
    {
      int64_t* d = nullptr;
      size_t cap = 0;
      HIPDEC_CHECK_HIP(arena_acquire((void**)&d, 256, &cap));
      struct Rel { void* p; size_t c; ~Rel() { arena_release(p, c); } } rel{d, cap};
      int64_t h[12];
      for (int k = 0; k < 5; k++) { h[k] = g->mine.empty() ? INT64_MIN : geo[k]; h[6 + k] = g->mine.empty() ? INT64_MAX : geo[k]; }
      h[5] = local_rc ? 1 : 0; h[11] = 0;
      HIPDEC_CHECK_HIP(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, g->stream));
      {
        RcclGroup grp;
        HIPDEC_CHECK_NCCL(grp.start());
        HIPDEC_CHECK_NCCL(rccl().AllReduce(d, d, 6, ncclInt64, ncclMax, g->comm, g->stream));
        HIPDEC_CHECK_NCCL(rccl().AllReduce(d + 6, d + 6, 6, ncclInt64, ncclMin, g->comm, g->stream));
        HIPDEC_CHECK_NCCL(grp.end());
      }
      HIPDEC_CHECK_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, g->stream));
      HIPDEC_CHECK_HIP(hipStreamSynchronize(g->stream));
      if (local_rc) return local_rc;
      if (h[5]) return set_error(HIPDEC_ERR_BITSTREAM, "grid_create_rccl: another rank could not build its shard");
      for (int k = 0; k < 5; k++)
        if (h[k] != h[6 + k]) return set_error(HIPDEC_ERR_BITSTREAM, "grid_create_rccl: the ranks' tiles differ in size, bit depth or chroma format");
      g->tile_bytes = (size_t)h[0]; g->tile_w = (int)h[1]; g->tile_h = (int)h[2]; g->bits = (int)h[3]; g->chroma = (int)h[4];
    }
    if (out_width > cols * g->tile_w || out_height > rows * g->tile_h) return set_error(HIPDEC_ERR_BITSTREAM, "grid_create_rccl: the output size exceeds the tiled area");
    g->csw = (g->chroma == 1 || g->chroma == 2) ? 2 : 1;
    g->csh = g->chroma == 1 ? 2 : 1;
    if (g->chroma && ((g->tile_w % g->csw) || (g->tile_h % g->csh))) return set_error(HIPDEC_ERR_UNSUPPORTED, "grid_create_rccl: subsampled tiles with odd dimensions");
    if (rank != 0) {
      if (!g->mine.empty()) HIPDEC_CHECK_HIP(arena_acquire((void**)&g->send, g->tile_bytes * g->mine.size(), &g->send_capacity));
    } else {
      g->recv_off.assign((size_t)nranks, 0);
      size_t o = 0;
      for (int p = 1; p < nranks; p++) {
        g->recv_off[(size_t)p] = o;
        const size_t n_p = p < n_tiles ? (size_t)((n_tiles - 1 - p) / nranks + 1) : 0;
        o += n_p * g->tile_bytes;
      }
      if (o) HIPDEC_CHECK_HIP(arena_acquire((void**)&g->recv, o, &g->recv_capacity));
      const size_t es = g->bits > 8 ? 2 : 1;
      const size_t cw = g->chroma ? (size_t)(out_width + g->csw - 1) / g->csw : 0, ch = g->chroma ? (size_t)(out_height + g->csh - 1) / g->csh : 0;
      size_t c = 0;
      g->stride[0] = ((size_t)out_width * es + 255) & ~(size_t)255; g->off[0] = c; c += g->stride[0] * (size_t)out_height;
      g->stride[1] = g->stride[2] = (cw * es + 255) & ~(size_t)255;
      g->off[1] = c; c += g->stride[1] * ch; g->off[2] = c; c += g->stride[2] * ch;
      HIPDEC_CHECK_HIP(arena_acquire((void**)&g->canvas, c ? c : 256, &g->canvas_capacity));
    }
    if (nranks > 1) HIPDEC_CHECK_HIP(arena_acquire((void**)&g->status_dev, 256, &g->status_capacity));
    *out = g.release();
    return 0;
  });
}

void hipdec_grid_rccl_free(hipdec_grid_rccl* g) { delete g; }

// pastes one packed tile (or the planes of a batch item) at its position in the canvas, clipped to the output size (pixelimage.cc:1130-1160)
static int paste_tile(hipdec_grid_rccl* g, int t, const uint8_t* const src[3], const size_t src_stride[3])
{
  const size_t es = g->bits > 8 ? 2 : 1;
  const int x0 = (t % g->cols) * g->tile_w, y0 = (t / g->cols) * g->tile_h;
  const int w = std::min(g->tile_w, g->out_w - x0), h = std::min(g->tile_h, g->out_h - y0);
  if (w <= 0 || h <= 0) return 0;
  for (int c = 0; c < (g->chroma ? 3 : 1); c++) {
    const size_t sw = c ? (size_t)g->csw : 1, sh = c ? (size_t)g->csh : 1;
    const size_t pw = ((size_t)w + sw - 1) / sw, ph = ((size_t)h + sh - 1) / sh, px = (size_t)x0 / sw, py = (size_t)y0 / sh;
    HIPDEC_CHECK_HIP(hipMemcpy2DAsync(g->canvas + g->off[c] + py * g->stride[c] + px * es, g->stride[c], src[c], src_stride[c], pw * es, ph, hipMemcpyDeviceToDevice, g->stream));
  }
  return 0;
}

int hipdec_grid_rccl_decode(hipdec_grid_rccl* g)
{
  if (!g) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_rccl_decode: NULL grid");
  return guarded("grid_rccl_decode", [&]() -> int {
    DeviceScope scope(g->device);
    const size_t es = g->bits > 8 ? 2 : 1;
    const size_t ysz = (size_t)g->tile_w * g->tile_h * es, cw = g->chroma ? (size_t)g->tile_w / g->csw : 0, chh = g->chroma ? (size_t)g->tile_h / g->csh : 0, csz = cw * chh * es;
    // a rank whose decode could not be queued still posts its send / receives: the peers must not be left waiting (its tiles are then undefined and
    // its error is what hipdec_grid_rccl_wait reports on that rank)
    int rc = 0;
    if (g->batch) {
      rc = hipdec_batch_run(g->batch, (void*)g->stream);
      if (!rc && g->rank != 0)
        for (int i = 0; i < (int)g->mine.size() && !rc; i++) rc = hipdec_batch_pack_item(g->batch, i, g->send + (size_t)i * g->tile_bytes, g->tile_bytes, (void*)g->stream);
    }
    const int n_tiles = g->rows * g->cols;
    if (g->nranks > 1) {
      RcclGroup grp;
      HIPDEC_CHECK_NCCL(grp.start());
      if (g->rank != 0) {
        if (!g->mine.empty()) HIPDEC_CHECK_NCCL(rccl().Send(g->send, g->tile_bytes * g->mine.size(), ncclUint8, 0, g->comm, g->stream));
      } else {
        for (int p = 1; p < g->nranks && p < n_tiles; p++) {
          const size_t n_p = (size_t)((n_tiles - 1 - p) / g->nranks + 1);
          HIPDEC_CHECK_NCCL(rccl().Recv(g->recv + g->recv_off[(size_t)p], n_p * g->tile_bytes, ncclUint8, p, g->comm, g->stream));
        }
      }
      HIPDEC_CHECK_NCCL(grp.end());
    }
    // from here on this rank has posted its part of the exchange: whatever happens next, wait() joins the status all-reduce (a rank that returned
    // early used to skip it and leave its peers inside the collective - ADVICE round 5); a failure of the paste below is this rank's status too
    g->attempted = true; g->decoded = false; g->status_known = false;
    if (!rc && g->rank == 0) {
      if (g->batch) (void)batch_follow_stream(g->batch, g->stream);   // (with stage overlap the pixel stages ran on the post stream)
      for (int i = 0; i < (int)g->mine.size() && !rc; i++) {   // own tiles: straight from the batch's output planes
        const uint8_t* src[3] = {nullptr, nullptr, nullptr};
        size_t ss[3] = {0, 0, 0};
        for (int c = 0; c < (g->chroma ? 3 : 1) && !rc; c++) {
          const void* p = nullptr;
          rc = hipdec_batch_device_plane(g->batch, i, c, &p, &ss[c]);
          src[c] = (const uint8_t*)p;
        }
        if (!rc) rc = paste_tile(g, g->mine[(size_t)i], src, ss);
      }
      for (int p = 1; p < g->nranks && p < n_tiles && !rc; p++) {
        int slot = 0;
        for (int t = p; t < n_tiles && !rc; t += g->nranks, slot++) {
          const uint8_t* base = g->recv + g->recv_off[(size_t)p] + (size_t)slot * g->tile_bytes;
          const uint8_t* src[3] = {base, base + ysz, base + ysz + csz};
          const size_t ss[3] = {(size_t)g->tile_w * es, cw * es, cw * es};
          rc = paste_tile(g, t, src, ss);
        }
      }
    }
    g->queue_rc = rc;
    g->queue_msg = rc ? hipdec_last_error() : "";
    g->decoded = rc == 0;
    return rc;
  });
}

// Every rank that called decode() calls wait() - also a rank whose decode() FAILED (its peers are inside the status all-reduce and would wait for it
// for ever): the Python host does that in a finally block, a C host must do the same.
int hipdec_grid_rccl_wait(hipdec_grid_rccl* g)
{
  if (!g || !g->attempted) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_rccl_wait: nothing was decoded");
  DeviceScope scope(g->device);
  if (g->status_known) {   // (rank 0 waits again inside to_rgb / read_plane)
    if (hipStreamSynchronize(g->stream) != hipSuccess) return set_error(HIPDEC_ERR_DEVICE, "grid_rccl_wait: the stream failed");
    return g->status_rc ? set_error(g->status_rc, "%s", g->status_msg.c_str()) : 0;
  }
  auto done = [&](int rc, const std::string& msg) { g->status_known = true; g->status_rc = rc; g->status_msg = msg; return rc ? set_error(rc, "%s", msg.c_str()) : 0; };
  int local = g->queue_rc;
  std::string local_msg = g->queue_msg;
  // local HIP failures are FOLDED into the reduced value instead of returning in front of the collective
  if (hipStreamSynchronize(g->stream) != hipSuccess && !local) { local = HIPDEC_ERR_DEVICE; local_msg = "grid_rccl_wait: the stream failed"; }
  if (!local && g->batch) { local = hipdec_batch_status(g->batch); if (local) local_msg = hipdec_last_error(); }   // device-side decode errors of this rank's shard
  // Every rank learns whether EVERY shard decoded (one 8-byte all-reduce): rank 0 has pasted whatever the peers sent, and must not hand out a canvas
  // with the undefined tiles of a rank whose shard failed.  All ranks call wait(), so the collective is matched.
  if (g->nranks > 1) {
    int64_t h = local ? 1 + g->rank : 0;
    bool hip_ok = hipMemcpyAsync(g->status_dev, &h, sizeof(h), hipMemcpyHostToDevice, g->stream) == hipSuccess;
    const bool nccl_ok = rccl().AllReduce(g->status_dev, g->status_dev, 1, ncclInt64, ncclMax, g->comm, g->stream) == ncclSuccess;
    int64_t r = 0;
    hip_ok = hip_ok && hipMemcpyAsync(&r, g->status_dev, sizeof(r), hipMemcpyDeviceToHost, g->stream) == hipSuccess;
    hip_ok = hipStreamSynchronize(g->stream) == hipSuccess && hip_ok;
    if (local) return done(local, local_msg);
    if (!nccl_ok || !hip_ok) return done(HIPDEC_ERR_DEVICE, "grid_rccl_wait: the status exchange failed");
    if (r) return done(HIPDEC_ERR_BITSTREAM, "grid_rccl_wait: the shard of rank " + std::to_string((int)r - 1) + " failed to decode: the canvas is incomplete");
    return done(0, "");
  }
  return done(local, local_msg);
}

int hipdec_grid_rccl_canvas_plane(hipdec_grid_rccl* g, int c, const void** dptr, size_t* stride)
{
  if (!g || c < 0 || c > 2 || !dptr || !stride || g->rank != 0 || (c > 0 && !g->chroma)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_rccl_canvas_plane: bad arguments (rank 0 owns the canvas)");
  *dptr = g->canvas + g->off[c]; *stride = g->stride[c];
  return 0;
}

int hipdec_grid_rccl_read_plane(hipdec_grid_rccl* g, int c, void* dst_host, size_t dst_stride)
{
  if (!g || !g->decoded || c < 0 || c > 2 || !dst_host || g->rank != 0 || (c > 0 && !g->chroma)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_rccl_read_plane: bad arguments (rank 0 owns the canvas)");
  if (int rc = hipdec_grid_rccl_wait(g)) return rc;
  DeviceScope scope(g->device);
  const size_t es = g->bits > 8 ? 2 : 1;
  const size_t sw = c ? (size_t)g->csw : 1, sh = c ? (size_t)g->csh : 1;
  const size_t w = ((size_t)g->out_w + sw - 1) / sw, h = ((size_t)g->out_h + sh - 1) / sh;
  HIPDEC_CHECK_HIP(hipMemcpy2D(dst_host, dst_stride, g->canvas + g->off[c], g->stride[c], w * es, h, hipMemcpyDeviceToHost));
  return 0;
}

int hipdec_grid_rccl_to_rgb(hipdec_grid_rccl* g, int out_chroma, int upsampling, int only_preferred, void* out, size_t out_stride, int out_on_device)
{
  if (!g || !g->decoded || !out || g->rank != 0) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "grid_rccl_to_rgb: bad arguments (rank 0 owns the canvas)");
  if (!g->chroma) return set_error(HIPDEC_ERR_UNSUPPORTED, "grid_rccl_to_rgb: monochrome grid");
  if (int rc = hipdec_grid_rccl_wait(g)) return rc;
  DeviceScope scope(g->device);
  hipdec_color_image img{};
  img.width = g->out_w; img.height = g->out_h; img.chroma = g->chroma; img.bit_depth = g->bits; img.on_device = 1;
  for (int c = 0; c < 3; c++) { img.plane[c] = g->canvas + g->off[c]; img.stride[c] = g->stride[c]; }
  hipdec_nclx nclx{1, g->info.colour_primaries, g->info.transfer_characteristics, g->info.matrix_coeffs, g->info.full_range_flag};
  return hipdec_color_convert(&img, &nclx, out_chroma, upsampling, only_preferred, out, out_stride, out_on_device);
}

}  // extern "C"
