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
  // ---- chains (sequence tracks with look-ahead): the items are consecutive samples of a track in decoding order - of ONE track, or of
  //      several tracks whose decoder instances asked at the same time (`tracks`).  The CABAC parser and the residual kernel run over all of them at
  //      once (parsing needs nothing of another picture: the parser emits MotionSyntax records, k_motion derives the vectors); motion derivation,
  //      prediction, reconstruction and the in-loop filters are launched step by step in dependency order, so an item's reference pictures -
  //      earlier items of the same batch among them - are complete when it starts
  bool chain = false;
  struct ChainItem {
    uint32_t first_rwave = 0, num_rwaves = 0;   // the picture's reconstruction wavefronts in the ReconWave table
    size_t off_full_pic = 0;                    // != 0: a PicParams copy without the conformance window in the upload region; one more SAO pass with it
                                                // writes the whole coded picture (reference pictures are addressed in coded coordinates)
    size_t off_full[3] = {0, 0, 0};
    uint32_t full_stride[3] = {0, 0, 0};
  };
  std::vector<ChainItem> chain_items;
  // Steps: within a track, maximal runs of consecutive samples without a dependency INSIDE the run; step k of the batch is step k of every track,
  // launched together.  Pixel steps break where a sample predicts from an earlier sample of the run (an intra-only track is one step: a plain
  // batch; the non-reference B pictures between two anchors share one); motion steps break only where a sample's COLLOCATED picture (temporal
  // candidates, 8.5.3.2.8) is in the run - merge / AMVP derivation reads nothing else of another picture - and run on a stream of their own
  // beside the pixel steps of earlier pictures.  Items are ordered by (pixel step, track, decoding order): a pixel step is a range of items; a
  // motion step is a range of the RowDesc table, which is ordered by (motion step, item).
  struct ChainStep {
    int first = 0, count = 0;                   // pixel steps: the items
    uint32_t first_rwave = 0, num_rwaves = 0;   // pixel steps: their reconstruction wavefronts
    uint32_t first_row = 0, num_rows = 0;       // motion steps: their rows in the RowDesc table
    int max_w = 0, max_h = 0, max_ow = 0, max_oh = 0;
    bool any_inter = false;
    int motion_need = 0;                        // pixel steps: motion steps [0, motion_need) hold the motion fields of the step's pictures
  };
  std::vector<ChainStep> pixel_steps, motion_steps;
  std::vector<int> pixel_step_of, motion_step_of;   // item -> index into pixel_steps / motion_steps
  std::vector<int> src_index;   // item -> index into the caller's data[] / sizes[] (RASL pictures that 8.3.3 drops are no items); empty: identity
  struct ChainTrack {
    SeqContext seq_after;       // the track's sequence state behind its last item; pictures of this batch carry RefPicture::batch_item until chain_resolve()
    std::vector<int> items;     // the track's items in decoding order
    std::vector<int> samples;   // ... and which of the track's samples each one is (index into ITS data[] / sizes[])
  };
  std::vector<ChainTrack> tracks;
  // ticket words of the per-step launches (dwords from off_ticket): [chain_ticket(k)] reconstruction of pixel step k, [+ 1] motion of motion step k
  static uint32_t chain_ticket(int k) { return 64u + 2u * (uint32_t)k; }
  int src(int i) const { return src_index.empty() ? i : src_index[(size_t)i]; }
};

// Parses n items (host worker threads for large batches) and lays the arena out ([upload region][control words][device-only
// buffers]).  Returns a hipdec_status; `err` holds the message.
// seqs: per item the sequence context of its decoder instance (reference pictures, POC state), or nullptr / an array of nullptrs for stills
int layout_batch_plan(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels,
                      std::string& err, const SeqContext* const* seqs = nullptr);
// The same for a chain: n consecutive samples of one track whose sequence state is `seq` (parsed one after the other, each committed to a working
// copy before the next is parsed).  Samples 8.3.3 drops are left out (b.src_index maps items to inputs).  b.pics may end up EMPTY (all dropped).
int layout_batch_plan_chain(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels,
                            std::string& err, const SeqContext& seq);
// Several tracks' chains in ONE batch: track t's samples are data[first[t]] .. data[first[t] + count[t] - 1] with sequence state *seqs[t].
// On failure *bad_track (if given) names the track whose sample was refused, or -1 when the combination was (8-bit beside 10-bit tracks).
int layout_batch_plan_chains(BatchLayout& b, int n_tracks, const int* first, const int* count, const void* const* data, const size_t* sizes,
                             uint64_t max_image_size_pixels, std::string& err, const SeqContext* const* seqs, int* bad_track = nullptr);
// Writes the upload region (b.upload_size bytes: descriptors, tables, the bitstreams as pushed) into `dst`, e.g. a pinned
// staging buffer.  The wave tables built by the plan are released afterwards.
// arena_base: the device address the arena will live at - needed by chains only (reference pictures inside the batch are addressed absolutely)
void layout_batch_fill(BatchLayout& b, const void* const* data, const size_t* sizes, uint8_t* dst, uint64_t arena_base = 0);
// reference picture of item j of a chain as later pictures (of this batch or of later ones) address it
RefPicture chain_ref_picture(const BatchLayout& b, int j, uint64_t arena_base);
// b.tracks[track].seq_after with the batch's own pictures addressed absolutely; `own` receives the POCs of those entries (their memory is this batch's arena)
void chain_resolve(BatchLayout& b, uint64_t arena_base, std::vector<int>& own, int track = 0);
// plan + fill into a vector (CPU-test emulation, small batches)
int layout_batch(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_image_size_pixels,
                 std::vector<uint8_t>& host_image, std::string& err, const SeqContext* const* seqs = nullptr);

}  // namespace hipdec
