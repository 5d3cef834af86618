// batch_layout.h — pure-host layout of one decode batch in its HBM arena (no HIP calls: shared by the
// device path in decoder.hip and by the CPU-test emulation in tests/emu).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include "hevc_headers.h"

namespace hipdec {

struct BatchLayout {
  std::vector<ParsedPicture> pics;
  std::vector<PicParams> params;
  size_t arena_size = 0, upload_size = 0;
  size_t off_pics = 0, off_subs = 0, off_rows = 0, off_waves = 0, off_ctrl = 0, ctrl_size = 0;
  size_t off_progress = 0, off_ctx = 0, off_row_progress = 0, off_ticket = 0, off_status = 0;
  size_t off_waitneed = 0, off_resume_k = 0, off_queue = 0, off_qctl = 0, off_saved = 0;
  uint32_t queue_cap = 0, pool = 0, pool_waves = 0;
  size_t off_rwaves = 0;
  uint32_t num_rwaves = 0;
  uint32_t num_subs = 0, num_rows = 0, num_waves = 0;
  std::vector<ParseWave> parse_waves;   // plan -> fill
  std::vector<ReconWave> recon_waves;
  bool wide = false;  // samples wider than 8 bit -> uint16 planes
  bool any_inter = false;   // the batch holds a P picture: the parser build with the inter syntax, k_motion and k_mc run
  int max_w = 0, max_h = 0, max_ow = 0, max_oh = 0, max_ctbs = 0;
};

// Parses n items (host worker threads for large batches) and lays the arena out ([upload region][control words][device-only
// buffers]).  Returns a hipdec_status; `err` holds the message.
// seqs: per item the sequence context of its decoder instance (reference pictures, POC state), or nullptr / an array of nullptrs for stills
int layout_batch_plan(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels,
                      std::string& err, const SeqContext* const* seqs = nullptr);
// Writes the upload region (b.upload_size bytes: descriptors, tables, the bitstreams as pushed) into `dst`, e.g. a pinned
// staging buffer.  The wave tables built by the plan are released afterwards.
void layout_batch_fill(BatchLayout& b, const void* const* data, const size_t* sizes, uint8_t* dst);
// plan + fill into a vector (CPU-test emulation, small batches)
int layout_batch(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels,
                 std::vector<uint8_t>& host_image, std::string& err, const SeqContext* const* seqs = nullptr);

}  // namespace hipdec
