// kernels.h — kernel argument blocks and the host-side launchers each kernel translation unit exports.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "batch_layout.h"

namespace hipdec {


struct ReconArgs {
  const PicParams* pics;
  const ReconWave* waves;
  uint32_t num_waves;
  uint8_t* arena;
  uint32_t* row_progress;  // per (batch row, component): CTBs completed
  uint32_t* ticket;
  int32_t* status;
  uint32_t inter_from_plane = 0;   // P / B pictures: k_mc has added the residuals of the inter coded units to its prediction (launch_mc(..., add_residual)): the rec
                                   // plane holds them complete, a CTB's tile is loaded whole and the wavefront only reconstructs the intra blocks
};

struct FilterArgs {
  const PicParams* pics;
  uint8_t* arena;
  const int32_t* status;   // batch status word: a failed parse leaves garbage maps, later stages skip the batch
};

// k_motion (inter_kernels.hip): one wave per CTB row of the batch (rows of intra pictures return at once)
struct MotionArgs {
  const PicParams* pics;
  const RowDesc* rows;
  uint32_t num_rows;
  uint8_t* arena;
  uint32_t* row_progress;  // per (batch row, slot): slot 2 is the motion wavefront's
  uint32_t* ticket;
  int32_t* status;
};

void launch_parse(const ParseArgs& a, hipStream_t s);
void launch_parse_throughput(const ParseArgs& a, hipStream_t s);   // parse_kernel_tp.hip: k_parse_occ8, LDS-resident contexts (4:0:0 / 4:2:0 intra batches)
void launch_parse_inter(const ParseArgs& a, hipStream_t s);   // parse_kernel_inter.hip: batches with P pictures (sequence tracks)
void launch_motion(const MotionArgs& a, hipStream_t s);       // P pictures: MotionSyntax -> motion field (merge / AMVP derivation)
// motion-compensated prediction into the rec planes; add_residual: plus the residual of every inter coded sample (then k_recon runs with inter_from_plane)
void launch_mc(const FilterArgs& a, int n_pics, int max_w, int max_h, bool wide, hipStream_t s, bool add_residual = false);
// Where do the residuals of inter coded units get added?  In k_mc (fully parallel over the picture), so that the reconstruction wavefront - a serial walk over a
// CTB's blocks, 2 CTBs behind the row above - only deals with the intra blocks of a P / B picture.  HIPDEC_INTER_RECON_PER_BLOCK=1: the form before (k_recon
// copies every inter block out of the plane and adds its residual), kept for A/B measurements.
inline bool inter_residual_in_mc() { static const bool per_block = getenv("HIPDEC_INTER_RECON_PER_BLOCK") != nullptr; return !per_block; }
void launch_parse_general(const ParseArgs& a, bool throughput, hipStream_t s);   // parse_kernel_general.hip: batches with 4:2:2 / 4:4:4 pictures
void launch_residual(const FilterArgs& a, int n_pics, int max_ctbs, bool general_chroma /* the batch holds 4:2:2 / 4:4:4 pictures */, hipStream_t s);
void launch_recon(const ReconArgs& a, bool wide, hipStream_t s, bool inter = false /* the batch holds P pictures */);
// one_pass: the batch holds intra pictures with 4:0:0 / 4:2:0 sampling only - both edge directions in one pass (k_deblock_fused)
void launch_deblock(const FilterArgs& a, int n_pics, int max_w, int max_h, bool wide, hipStream_t s, bool one_pass = false);
// may_keep = false: no picture of the batch has lossless CUs or unfiltered PCM units (8.7.3 "samples stay as they are"): the kernel variant without those paths
// restricted = false: every picture has PicParams::sao_free_neighbours (no slice / tile boundary restricts the edge-offset neighbours)
void launch_sao(const FilterArgs& a, int n_pics, int max_out_w, int max_out_h, bool wide, hipStream_t s, bool may_keep = true, bool restricted = true);
// SAO + crop with the RGB24 emission fused into the store path (8-bit 4:2:0 only); color_params_dev: one colordev::ColorParams per picture
// host_pics / host_color_params: the host's copies of the n_pics parameter blocks (may be null): which of the two kernels the batch needs
void launch_sao_rgb(const FilterArgs& a, const void* color_params_dev, int n_pics, int max_out_w, int max_out_h, hipStream_t s, bool may_keep = true, bool restricted = true,
                    const PicParams* host_pics = nullptr, const void* host_color_params = nullptr);

// Chain batches (BatchLayout::chain; used by decoder.hip:launch_all and by the CPU-test emulation).  The motion fields of motion step k: k_motion over
// the CTB rows of the step's pictures (rows of intra pictures return at once).  Needs the parser's output and the motion fields of earlier steps only.
inline void launch_chain_motion(const BatchLayout& L, uint8_t* arena, int k, hipStream_t s)
{
  const BatchLayout::ChainStep& st = L.motion_steps[(size_t)k];
  if (!st.any_inter) return;
  MotionArgs ma{(const PicParams*)(arena + L.off_pics), (const RowDesc*)(arena + L.off_rows) + st.first_row, st.num_rows, arena,
                (uint32_t*)(arena + L.off_row_progress), (uint32_t*)(arena + L.off_ticket) + BatchLayout::chain_ticket(k) + 1, (int32_t*)(arena + L.off_status)};
  launch_motion(ma, s);
}
// All motion steps as ONE launch: the RowDesc table is ordered by (motion step, item, row) and k_motion hands rows out by ticket, so a picture's
// collocated picture - an item of an earlier motion step - holds earlier tickets; the kernel waits for that picture's row progress CTB by CTB
// (RefFrame::progress_row), and the pictures of a chain follow each other at a 2-CTB distance instead of one launch per step.
inline void launch_chain_motion_all(const BatchLayout& L, uint8_t* arena, hipStream_t s)
{
  if (!L.any_inter) return;
  MotionArgs ma{(const PicParams*)(arena + L.off_pics), (const RowDesc*)(arena + L.off_rows), L.num_rows, arena,
                (uint32_t*)(arena + L.off_row_progress), (uint32_t*)(arena + L.off_ticket) + BatchLayout::chain_ticket(0) + 1, (int32_t*)(arena + L.off_status)};
  launch_motion(ma, s);
}
// The pixel stages of pixel step k: motion-compensated prediction, reconstruction, deblocking, SAO of the step's pictures together - and for a
// picture with a conformance window one more SAO pass that writes the whole coded picture for the pictures that predict from it.  Stream order
// makes every earlier step complete; the motion fields of the step's pictures must be (the caller orders launch_chain_motion before it).
inline void launch_chain_pixels(const BatchLayout& L, uint8_t* arena, int k, hipStream_t s)
{
  const BatchLayout::ChainStep& st = L.pixel_steps[(size_t)k];
  const PicParams* pics = (const PicParams*)(arena + L.off_pics);
  int32_t* status = (int32_t*)(arena + L.off_status);
  FilterArgs fa{pics + st.first, arena, status};
  if (st.any_inter) launch_mc(fa, st.count, st.max_w, st.max_h, L.wide, s, inter_residual_in_mc());
  ReconArgs ra{pics, (const ReconWave*)(arena + L.off_rwaves) + st.first_rwave, st.num_rwaves, arena, (uint32_t*)(arena + L.off_row_progress),
               (uint32_t*)(arena + L.off_ticket) + BatchLayout::chain_ticket(k), status, inter_residual_in_mc() ? 1u : 0u};
  launch_recon(ra, L.wide, s, L.any_inter);
  launch_deblock(fa, st.count, st.max_w, st.max_h, L.wide, s);
  bool may_keep = false, restricted = false;
  for (int i = st.first; i < st.first + st.count; i++) {
    const PicParams& P = L.params[(size_t)i];
    if (P.transquant_bypass_enabled || (P.pcm_enabled && P.pcm_loop_filter_disabled)) may_keep = true;
    if (!P.sao_free_neighbours) restricted = true;
  }
  launch_sao(fa, st.count, st.max_ow, st.max_oh, L.wide, s, may_keep, restricted);
  for (int i = st.first; i < st.first + st.count; i++) {
    const BatchLayout::ChainItem& ci = L.chain_items[(size_t)i];
    if (!ci.off_full_pic) continue;
    const PicParams& P = L.params[(size_t)i];
    FilterArgs full{(const PicParams*)(arena + ci.off_full_pic), arena, status};
    launch_sao(full, 1, P.width, P.height, L.wide, s, may_keep, restricted);
  }
}

}  // namespace hipdec
