// kernels.h — kernel argument blocks and the host-side launchers each kernel translation unit exports.
#pragma once
#include <hip/hip_runtime.h>
#include "hevc_device.h"

namespace hipdec {


struct ReconArgs {
  const PicParams* pics;
  const ReconWave* waves;
  uint32_t num_waves;
  uint8_t* arena;
  uint32_t* row_progress;  // per (batch row, component): CTBs completed
  uint32_t* ticket;
  int32_t* status;
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
void launch_parse_inter(const ParseArgs& a, hipStream_t s);   // parse_kernel_inter.hip: batches with P pictures (sequence tracks)
void launch_motion(const MotionArgs& a, hipStream_t s);       // P pictures: MotionSyntax -> motion field (merge / AMVP derivation)
void launch_mc(const FilterArgs& a, int n_pics, int max_w, int max_h, bool wide, hipStream_t s);   // motion-compensated prediction into the rec planes
void launch_parse_general(const ParseArgs& a, bool throughput, hipStream_t s);   // parse_kernel_general.hip: batches with 4:2:2 / 4:4:4 pictures
void launch_residual(const FilterArgs& a, int n_pics, int max_ctbs, bool general_chroma /* the batch holds 4:2:2 / 4:4:4 pictures */, hipStream_t s);
void launch_recon(const ReconArgs& a, bool wide, hipStream_t s, bool inter = false /* the batch holds P pictures */);
void launch_deblock(const FilterArgs& a, int n_pics, int max_w, int max_h, bool wide, hipStream_t s);
// may_keep = false: no picture of the batch has lossless CUs or unfiltered PCM units (8.7.3 "samples stay as they are"): the kernel variant without those paths
// restricted = false: every picture has PicParams::sao_free_neighbours (no slice / tile boundary restricts the edge-offset neighbours)
void launch_sao(const FilterArgs& a, int n_pics, int max_out_w, int max_out_h, bool wide, hipStream_t s, bool may_keep = true, bool restricted = true);
// SAO + crop with the RGB24 emission fused into the store path (8-bit 4:2:0 only); color_params_dev: one colordev::ColorParams per picture
void launch_sao_rgb(const FilterArgs& a, const void* color_params_dev, int n_pics, int max_out_w, int max_out_h, hipStream_t s, bool may_keep = true, bool restricted = true);

}  // namespace hipdec
