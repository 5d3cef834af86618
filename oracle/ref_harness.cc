// ref_harness.cc — test infrastructure: drives the REAL reference colour pipeline
// (libheif/color-conversion/colorconversion.cc:490 convert_colorspace and the ops it selects)
// compiled from /root/reference into oracle/_ref/libheif.so.  Built into oracle/_ref/libref_harness.so
// by oracle/Makefile.ref; never linked into the product.
#include <cstring>
#include <memory>
#include "libheif/heif.h"
#include "api_structs.h"
#include "image/pixelimage.h"
#include "color-conversion/colorconversion.h"
#include "nclx.h"

extern "C" {

// planes: tightly packed input planes (u8 for bpp<=8, u16 LE otherwise).
// out: up to 4 planes copied tightly into out_buf[i] (caller-allocated, >= w*h*8 bytes each);
// out_info[i*3+0..2] = width, height, bytes per row actually copied.  Returns number of planes
// (order: interleaved | R,G,B[,A] | Y,Cb,Cr[,A]) or a negative libheif error code.
int ref_convert_colorspace(int w, int h, int bpp, int in_colorspace, int in_chroma,
                           const void* const* planes, int n_planes,
                           int has_nclx, int cp, int tc, int mc, int full_range,
                           int target_colorspace, int target_chroma, int out_bpp,
                           int preferred_upsampling, int only_use_preferred,
                           void** out_buf, int* out_info)
{
  heif_init(nullptr);
  heif_image* img = nullptr;
  heif_error err = heif_image_create(w, h, (heif_colorspace)in_colorspace, (heif_chroma)in_chroma, &img);
  if (err.code) return -(int)err.code;
  static const heif_channel ycc[4] = {heif_channel_Y, heif_channel_Cb, heif_channel_Cr, heif_channel_Alpha};
  for (int c = 0; c < n_planes; c++) {
    int pw = w, ph = h;
    if (c == 1 || c == 2) {
      if (in_chroma == heif_chroma_420) { pw = (w + 1) / 2; ph = (h + 1) / 2; }
      else if (in_chroma == heif_chroma_422) { pw = (w + 1) / 2; }
    }
    err = heif_image_add_plane(img, ycc[c], pw, ph, bpp);
    if (err.code) { heif_image_release(img); return -(int)err.code; }
    size_t stride;
    uint8_t* dst = heif_image_get_plane2(img, ycc[c], &stride);
    size_t row = (size_t)pw * (bpp > 8 ? 2 : 1);
    for (int y = 0; y < ph; y++) memcpy(dst + y * stride, (const uint8_t*)planes[c] + y * row, row);
  }
  if (has_nclx) {
    heif_color_profile_nclx* n = heif_nclx_color_profile_alloc();
    n->color_primaries = (heif_color_primaries)cp;
    n->transfer_characteristics = (heif_transfer_characteristics)tc;
    n->matrix_coefficients = (heif_matrix_coefficients)mc;
    n->full_range_flag = (uint8_t)full_range;
    heif_image_set_nclx_color_profile(img, n);
    heif_nclx_color_profile_free(n);
  }
  heif_color_conversion_options opt;
  heif_color_conversion_options_set_defaults(&opt);
  opt.preferred_chroma_upsampling_algorithm = (heif_chroma_upsampling_algorithm)preferred_upsampling;
  opt.only_use_preferred_chroma_algorithm = (uint8_t)only_use_preferred;
  // same call libheif makes from HeifContext::convert_to_output_colorspace (context.cc:1515)
  nclx_profile target = nclx_profile::undefined();
  auto res = convert_colorspace(img->image, (heif_colorspace)target_colorspace, (heif_chroma)target_chroma,
                                target, out_bpp, opt, nullptr, heif_get_global_security_limits());
  if (!res) { heif_image_release(img); return -(int)res.error().error_code; }
  std::shared_ptr<HeifPixelImage> out = *res;
  int n = 0;
  static const heif_channel order_i[1] = {heif_channel_interleaved};
  static const heif_channel order_rgb[4] = {heif_channel_R, heif_channel_G, heif_channel_B, heif_channel_Alpha};
  const heif_channel* order; int cnt;
  if (out->has_channel(heif_channel_interleaved)) { order = order_i; cnt = 1; }
  else if (out->has_channel(heif_channel_R)) { order = order_rgb; cnt = 4; }
  else { order = ycc; cnt = 4; }
  for (int i = 0; i < cnt; i++) {
    if (!out->has_channel(order[i])) continue;
    size_t stride;
    const uint8_t* src = out->get_channel_memory(order[i], &stride);
    int pw = out->get_width(order[i]), ph = out->get_height(order[i]);
    int bits = out->get_bits_per_pixel(order[i]);
    size_t bytes_pp;
    if (order[i] == heif_channel_interleaved) {
      switch (out->get_chroma_format()) {
        case heif_chroma_interleaved_RGB: bytes_pp = 3; break;
        case heif_chroma_interleaved_RGBA: bytes_pp = 4; break;
        case heif_chroma_interleaved_RRGGBB_BE: case heif_chroma_interleaved_RRGGBB_LE: bytes_pp = 6; break;
        default: bytes_pp = 8; break;
      }
    } else bytes_pp = bits > 8 ? 2 : 1;
    size_t row = (size_t)pw * bytes_pp;
    for (int y = 0; y < ph; y++) memcpy((uint8_t*)out_buf[n] + y * row, src + y * stride, row);
    out_info[n * 3 + 0] = pw; out_info[n * 3 + 1] = ph; out_info[n * 3 + 2] = (int)row;
    n++;
  }
  heif_image_release(img);
  return n;
}

}  // extern "C"

// decoding options for the pin harness (tests/test_reference_decoder_pin.py), filled in against the REAL header layout: a named decoder
// (plugin_registry.cc:264-288), no geometric transformations, and the decoded planes handed through untouched (output nclx = the image's own:
// with the default options libheif converts limited-range YCbCr to its sRGB default, context.cc:1515-1567).  Caller frees with heif_decoding_options_free.
extern "C" heif_decoding_options* refh_decoding_options(const char* decoder_id)
{
  heif_decoding_options* o = heif_decoding_options_alloc();
  if (!o) return nullptr;
  o->decoder_id = decoder_id;              // the caller keeps the string alive
  o->ignore_transformations = 1;
  o->output_image_nclx_profile_passthrough = 1;
  return o;
}
