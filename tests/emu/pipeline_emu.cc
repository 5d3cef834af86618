// pipeline_emu.cc — CPU-TEST-ONLY: runs the kernels AFTER the parser (residual, reconstruction, deblocking, SAO; the very
// sources of libheif_amd/csrc/*.hip, compiled for the host against tests/emu/shim/hip/hip_runtime.h) on a batch that
// emu_run_parse() has parsed, with the same argument blocks as decoder.hip:launch_all.  NOT part of the product.
#include <cstdlib>
#include <cstring>
#include "emu_batch.h"
#include "kernels.h"
#include "hipdec_internal.h"
#include <vector>

using namespace hipdec;

extern "C" {

// stages: bit 0 residual, 1 recon, 2 deblock, 3 sao, 4 sao with the RGB24 emission fused into its store path (then `rgb` receives the
// interleaved rows of every item, tightly packed one after the other).  Returns the device status word.
int emu_run_pipeline_rgb(EmuBatch* b, int stages, uint8_t* rgb);
int emu_run_pipeline(EmuBatch* b, int stages) { return emu_run_pipeline_rgb(b, stages, nullptr); }
int emu_run_pipeline_rgb(EmuBatch* b, int stages, uint8_t* rgb)
{
  uint8_t* a = b->arena.data();
  const BatchLayout& L = b->L;
  const int n = (int)L.params.size();
  ReconArgs ra{(const PicParams*)(a + L.off_pics), (const ReconWave*)(a + L.off_rwaves), L.num_rwaves, a,
               (uint32_t*)(a + L.off_row_progress), (uint32_t*)(a + L.off_ticket) + 1, (int32_t*)(a + L.off_status)};
  FilterArgs fa{(const PicParams*)(a + L.off_pics), a, (const int32_t*)(a + L.off_status)};
  bool general = false;
  for (const PicParams& P : L.params) if (P.chroma_format_idc >= 2) general = true;
  if (stages & 1) launch_residual(fa, n, L.max_ctbs, general, nullptr);
  if ((stages & 2) && L.any_inter) {   // P pictures: motion field, motion-compensated prediction (as decoder.hip:launch_all)
    MotionArgs ma{(const PicParams*)(a + L.off_pics), (const RowDesc*)(a + L.off_rows), L.num_rows, a, (uint32_t*)(a + L.off_row_progress),
                  (uint32_t*)(a + L.off_ticket) + 2, (int32_t*)(a + L.off_status)};
    launch_motion(ma, nullptr);
    launch_mc(fa, n, L.max_w, L.max_h, L.wide, nullptr, inter_residual_in_mc());
    ra.inter_from_plane = inter_residual_in_mc() ? 1u : 0u;
  }
  if (stages & 2) launch_recon(ra, L.wide, nullptr, L.any_inter);
  if (stages & 4) launch_deblock(fa, n, L.max_w, L.max_h, L.wide, nullptr, !L.any_inter && !general);   // (as decoder.hip:launch_all)
  bool may_keep = false, restricted = false;   // as decoder.hip:launch_all picks the kernel variant
  for (const PicParams& P : L.params) {
    if (P.transquant_bypass_enabled || (P.pcm_enabled && P.pcm_loop_filter_disabled)) may_keep = true;
    if (!P.sao_free_neighbours) restricted = true;
  }
  if (stages & 8) launch_sao(fa, n, L.max_ow, L.max_oh, L.wide, nullptr, may_keep, restricted);
  if ((stages & 16) && rgb) {
    // the parameter blocks come from the product's own entry points in capture mode, as in hipdec_batch_run_rgb (decoder.hip)
    static ColorBatchState st;
    color_capture_begin();
    uint8_t* dst = rgb;
    for (int i = 0; i < n; i++) {
      const PicParams& P = L.params[i];
      const hipdec_image_info& I = L.pics[i].info;
      hipdec_nclx nclx{1, I.colour_primaries, I.transfer_characteristics, I.matrix_coeffs, I.full_range_flag};
      const int m = I.matrix_coeffs == 2 ? 6 : I.matrix_coeffs;
      int rc;
      if (I.full_range_flag && m != 0 && m != 8)
        rc = hipdec_color_420_to_rgb24(a + P.off_out[0], P.out_stride[0], a + P.off_out[1], P.out_stride[1], a + P.off_out[2], P.out_stride[2], P.out_width,
                                       P.out_height, &nclx, dst, (size_t)P.out_width * 3, 0, nullptr);
      else
        rc = hipdec_color_ycbcr_to_rgb24_float(a + P.off_out[0], P.out_stride[0], a + P.off_out[1], P.out_stride[1], a + P.off_out[2], P.out_stride[2],
                                               P.out_width, P.out_height, 1, &nclx, dst, (size_t)P.out_width * 3, 0, nullptr);
      if (rc) { color_capture_abort(); return -100 + rc; }
      dst += (size_t)P.out_width * P.out_height * 3;
    }
    const void* dev = nullptr; int variant = -1, count = 0;
    if (color_capture_take(st, nullptr, &dev, &variant, &count) || variant != color_variant_rgb24_u8()) return -200;
    launch_sao_rgb(fa, dev, n, L.max_ow, L.max_oh, nullptr, may_keep, restricted);
  }
  b->status = *(int32_t*)(a + L.off_status);
  return b->status;
}

// a chain batch (emu_seq_create_chain) behind emu_run_parse(): the residual kernel over every picture, then the pixel stages picture by picture in
// decoding order - the launch sequence of decoder.hip:launch_all for BatchLayout::chain
int emu_run_pipeline_chain(EmuBatch* b)
{
  uint8_t* a = b->arena.data();
  const BatchLayout& L = b->L;
  const int n = (int)L.params.size();
  if (!L.chain) return -1;
  if (!n) return 0;
  FilterArgs fa{(const PicParams*)(a + L.off_pics), a, (const int32_t*)(a + L.off_status)};
  bool general = false;
  for (const PicParams& P : L.params) if (P.chroma_format_idc >= 2) general = true;
  launch_residual(fa, n, L.max_ctbs, general, nullptr);
  if (!getenv("HIPDEC_CHAIN_MOTION_STEPS")) {
    // as the product: the motion fields of all pictures with ONE launch (rows by ticket in (motion step, item) order), then the pixel steps
    launch_chain_motion_all(L, a, nullptr);
    for (int k = 0; k < (int)L.pixel_steps.size(); k++) launch_chain_pixels(L, a, k, nullptr);
  } else {
    // (the A/B form: every motion step a pixel step needs, then the pixel step)
    int motion_done = 0;
    for (int k = 0; k < (int)L.pixel_steps.size(); k++) {
      const BatchLayout::ChainStep& st = L.pixel_steps[(size_t)k];
      const int need = st.motion_need;
      for (; motion_done < need; motion_done++) launch_chain_motion(L, a, motion_done, nullptr);
      launch_chain_pixels(L, a, k, nullptr);
    }
  }
  b->status = *(int32_t*)(a + L.off_status);
  return b->status;
}

// step structure of a chain batch: number of pixel steps / motion steps (batch_layout.h)
int emu_chain_steps(EmuBatch* b, int* pixel_steps, int* motion_steps)
{
  if (!b->L.chain) return -1;
  *pixel_steps = (int)b->L.pixel_steps.size(); *motion_steps = (int)b->L.motion_steps.size();
  return 0;
}

// The picture of batch `b` (one item) was decoded: it enters the sequence's DPB as a reference picture - its output planes when they ARE the
// coded picture, else an uncropped copy made by running the SAO kernel once more without the conformance window (what the product's decoder
// does lazily, decoder.hip) - and the DPB drops what the picture's RPS no longer names.
int emu_seq_commit(EmuSeq* q, EmuBatch* b)
{
  const BatchLayout& L = b->L;
  const PicParams& P = L.params[0];
  const ParsedPicture& pp = L.pics[0];
  uint8_t* a = b->arena.data();
  RefPicture rp;
  rp.poc = pp.poc;
  rp.width = P.width; rp.height = P.height; rp.chroma_format_idc = P.chroma_format_idc; rp.bit_depth_luma = P.bit_depth_luma; rp.bit_depth_chroma = P.bit_depth_chroma;
  rp.log2_ctb = P.log2_ctb;
  const bool cropped = P.out_width != P.width || P.out_height != P.height || P.crop_x || P.crop_y;
  if (!cropped) {
    for (int c = 0; c < 3; c++) { rp.plane[c] = (uint64_t)(uintptr_t)(a + P.off_out[c]); rp.stride[c] = P.out_stride[c]; }
  } else {
    const size_t es = L.wide ? 2 : 1;
    auto* buf = new std::vector<uint8_t>();
    size_t off[3], total = 0;
    uint32_t stride[3];
    for (int c = 0; c < 3; c++) {
      const size_t w = c ? P.cwidth : P.width, h = c ? P.cheight : P.height;
      stride[c] = (uint32_t)(((w ? w : 1) * es + 63) / 64 * 64);
      off[c] = total; total += (size_t)stride[c] * (h ? h : 1) + 256;
    }
    buf->assign(total, 0);
    q->full_frames.push_back(buf);
    std::vector<PicParams> tmp(1, P);
    PicParams& F = tmp[0];
    F.crop_x = F.crop_y = 0; F.out_width = P.width; F.out_height = P.height; F.out_cwidth = P.cwidth; F.out_cheight = P.cheight;
    for (int c = 0; c < 3; c++) { F.off_out[c] = (uint64_t)(uintptr_t)(buf->data() + off[c]) - (uint64_t)(uintptr_t)a; F.out_stride[c] = stride[c]; }
    FilterArgs fa{tmp.data(), a, (const int32_t*)(a + L.off_status)};
    bool may_keep = P.transquant_bypass_enabled || (P.pcm_enabled && P.pcm_loop_filter_disabled), restricted = !P.sao_free_neighbours;
    launch_sao(fa, 1, P.width, P.height, L.wide, nullptr, may_keep, restricted);
    for (int c = 0; c < 3; c++) { rp.plane[c] = (uint64_t)(uintptr_t)(buf->data() + off[c]); rp.stride[c] = stride[c]; }
  }
  rp.mf = P.is_inter ? (uint64_t)(uintptr_t)(a + P.off_mf) : 0;   // the picture's motion field stays with it: the collocated picture of temporal candidates
  seq_commit(q->ctx, pp);
  q->ctx.dpb.push_back(rp);
  return 0;
}

// motion field of item i after the pipeline (k_motion): per 4x4 unit in RASTER order [list][x, y] motion vectors (int16), [list] ref_idx (int8),
// pred (uint8) - the layout of the oracle's taps
int emu_motion(EmuBatch* b, int i, int16_t* mv, int8_t* ref_idx, uint8_t* pred)
{
  const PicParams& P = b->L.params[i];
  if (!P.is_inter) return -1;
  const uint8_t* a = b->arena.data();
  const MotionUnit* mf = (const MotionUnit*)(a + P.off_mf);
  const int uw = (P.width + 3) >> 2, uh = (P.height + 3) >> 2, side = 1 << (P.log2_ctb - 2);
  for (int uy = 0; uy < uh; uy++)
    for (int ux = 0; ux < uw; ux++) {
      const int cx = ux / side, cy = uy / side, lx = ux % side, ly = uy % side;
      uint32_t z = 0;
      for (int k = 0; k < 4; k++) z |= (((uint32_t)lx >> k) & 1u) << (2 * k) | (((uint32_t)ly >> k) & 1u) << (2 * k + 1);
      const MotionUnit m = mf[((size_t)(cy * P.ctb_w + cx) << P.units_per_ctb_log2) + z];
      const size_t o = (size_t)uy * uw + ux;
      for (int X = 0; X < 2; X++) { mv[4 * o + 2 * X] = m.mv[X][0]; mv[4 * o + 2 * X + 1] = m.mv[X][1]; ref_idx[2 * o + X] = m.ref_idx[X]; }
      pred[o] = (uint8_t)(m.slot_pred[0] >> 6);
    }
  return 0;
}

// output plane c of item i (cropped size, as hipdec_batch_read_plane); dst rows are tightly packed
int emu_plane(EmuBatch* b, int i, int c, void* dst)
{
  const PicParams& P = b->L.params[i];
  const size_t es = b->L.wide ? 2 : 1;
  const int w = c ? P.out_cwidth : P.out_width, h = c ? P.out_cheight : P.out_height;
  for (int y = 0; y < h; y++)
    memcpy((uint8_t*)dst + (size_t)y * w * es, b->arena.data() + P.off_out[c] + (size_t)y * P.out_stride[c], (size_t)w * es);
  return 0;
}

// FNV-1a over the read-only upload region (parameter blocks, tables, bitstreams): no kernel may ever write there
uint64_t emu_upload_hash(EmuBatch* b)
{
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < b->L.upload_size; i++) { h ^= b->arena[i]; h *= 1099511628211ull; }
  return h;
}

int emu_out_size(EmuBatch* b, int i, int* out /* w, h, cw, ch, bytes per sample */)
{
  const PicParams& P = b->L.params[i];
  out[0] = P.out_width; out[1] = P.out_height; out[2] = P.out_cwidth; out[3] = P.out_cheight; out[4] = b->L.wide ? 2 : 1;
  return 0;
}

}  // extern "C"
