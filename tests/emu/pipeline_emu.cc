// pipeline_emu.cc — CPU-TEST-ONLY: runs the kernels AFTER the parser (residual, reconstruction, deblocking, SAO; the very
// sources of libheif_amd/csrc/*.hip, compiled for the host against tests/emu/shim/hip/hip_runtime.h) on a batch that
// emu_run_parse() has parsed, with the same argument blocks as decoder.hip:launch_all.  NOT part of the product.
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
  if (stages & 2) launch_recon(ra, L.wide, nullptr);
  if (stages & 4) launch_deblock(fa, n, L.max_w, L.max_h, L.wide, nullptr);
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
