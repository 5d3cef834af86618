// colorconversion_hip.cc — see colorconversion_hip.h.
//
// The op offers, at SpeedCosts_Hardware (libheif/color-conversion/colorconversion.h:58-65), the output states the stock
// YCbCr 4:2:0 -> interleaved RGB ops offer (Op_YCbCr420_to_RGB24 / _RGB32 yuv2rgb.cc:298-341, :430-478; the chain
// Op_YCbCr_to_RGB<uint8_t> + Op_RGB_to_RGB24_32 yuv2rgb.cc:35-92, rgb2rgb.cc:30-70; Op_YCbCr420_to_RRGGBBaa yuv2rgb.cc:566-620),
// so the pipeline search prefers it wherever it applies, and produces the pixels those ops would: hipdec_color_plan() picks the
// same chain, the HIP kernels restate its arithmetic bit for bit.  libheifhip.so is found at run time (it is the decoder plugin
// libheif has already loaded); without it the op offers nothing and the stock ops run.
#include "colorconversion_hip.h"
#include "image/pixelimage.h"
#include <dlfcn.h>
#include <cstddef>
#include <cstdint>
#include <mutex>

namespace {

// restated from include/heif_hipdec.h (the C ABI of libheifhip.so)
struct hipdec_nclx { int has_nclx, colour_primaries, transfer_characteristics, matrix_coefficients, full_range_flag; };
struct hipdec_color_image {
  int width, height, chroma, bit_depth;
  const void* plane[4];
  size_t stride[4];
  int on_device;
};
using plan_fn = int (*)(int, int, int, const hipdec_nclx*, int, int, int, int*, int*);
using convert_fn = int (*)(const hipdec_color_image*, const hipdec_nclx*, int, int, int, void*, size_t, int);
using device_count_fn = int (*)();
using last_error_fn = const char* (*)();

struct HipApi {
  plan_fn plan = nullptr;
  convert_fn convert = nullptr;
  last_error_fn last_error = nullptr;
  bool usable = false;
};

std::mutex g_api_mutex;
HipApi g_api;
bool g_api_probed = false;

HipApi hip_api()
{
  std::lock_guard<std::mutex> lock(g_api_mutex);
  if (!g_api.usable && !g_api_probed) {
    // a host that links libheifhip.so itself (static plugin registration) exposes the entry points globally
    g_api_probed = true;
    g_api.plan = (plan_fn) dlsym(RTLD_DEFAULT, "hipdec_color_plan");
    g_api.convert = (convert_fn) dlsym(RTLD_DEFAULT, "hipdec_color_convert");
    g_api.last_error = (last_error_fn) dlsym(RTLD_DEFAULT, "hipdec_last_error");
    auto count = (device_count_fn) dlsym(RTLD_DEFAULT, "hipdec_device_count");
    g_api.usable = g_api.plan && g_api.convert && count && count() > 0;
  }
  return g_api;
}

hipdec_nclx to_hipdec(const nclx_profile& p)
{
  return hipdec_nclx{1, (int) p.get_colour_primaries(), (int) p.get_transfer_characteristics(), (int) p.get_matrix_coefficients(),
                     p.get_full_range_flag() ? 1 : 0};
}

int upsampling_of(const heif_color_conversion_options& options)
{
  return options.preferred_chroma_upsampling_algorithm == heif_chroma_upsampling_nearest_neighbor ? 1 : 2;
}

}  // namespace


// The decoder plugin is dlopen()ed with local symbol scope (libheif/plugins_unix.cc:103-118), so it announces its colour entry
// points itself: libheifhip's init_plugin() looks this function up in the hosting process and calls it (csrc/plugin.hip).
// `usable` = a HIP device is present.
extern "C" __attribute__((visibility("default")))
void heif_color_conversion_register_hip_backend(int (*plan)(int, int, int, const void*, int, int, int, int*, int*),
                                                int (*convert)(const void*, const void*, int, int, int, void*, size_t, int),
                                                const char* (*last_error)(void), int usable)
{
  std::lock_guard<std::mutex> lock(g_api_mutex);
  g_api.plan = (plan_fn) plan;
  g_api.convert = (convert_fn) convert;
  g_api.last_error = last_error;
  g_api.usable = plan && convert && usable;
  g_api_probed = true;
}


std::vector<ColorStateWithCost>
Op_YCbCr_to_RGB_hip::state_after_conversion(const ColorState& input_state,
                                            const ColorState& target_state,
                                            const heif_color_conversion_options& options,
                                            const heif_color_conversion_options_ext& options_ext) const
{
  const HipApi api = hip_api();
  if (!api.usable) {
    return {};
  }

  if (input_state.colorspace != heif_colorspace_YCbCr ||
      (input_state.chroma != heif_chroma_420 && input_state.chroma != heif_chroma_422 && input_state.chroma != heif_chroma_444)) {
    return {};
  }

  if (input_state.has_alpha && input_state.get_alpha_bits_per_pixel() != input_state.bits_per_pixel) {
    return {};
  }

  const hipdec_nclx nclx = to_hipdec(input_state.nclx);
  const int upsampling = upsampling_of(options);
  const int only_preferred = options.only_use_preferred_chroma_algorithm ? 1 : 0;

  std::vector<ColorStateWithCost> states;

  auto offer = [&](heif_chroma chroma, bool alpha, int bpp) {
    int ops[8], n = 0;
    // the planner of libheifhip.so decides whether this conversion is one it restates (same decision table the stock ops encode)
    if (api.plan(input_state.bits_per_pixel, (int) input_state.chroma /* heif_chroma_420 / 422 / 444 = 1 / 2 / 3 */, input_state.has_alpha ? 1 : 0, &nclx,
                 (int) chroma, upsampling, only_preferred, ops, &n) != 0) {
      return;
    }
    // (an 8-bit target from > 8-bit planes - the caller's convert_hdr_to_8bit - is offered as well: the planner restates which of the stock
    //  chains the search ends on for the state, Op_to_sdr_planes before or behind the conversion, so the result is the stock pipeline's.  If the
    //  op only took 8-bit planes, the search would put Op_to_sdr_planes in front of it in every state - and change the pixels in most)
    if (bpp > 8 && input_state.bits_per_pixel <= 8) {
      return;
    }
    ColorState output_state;
    output_state.colorspace = heif_colorspace_RGB;
    output_state.chroma = chroma;
    output_state.has_alpha = alpha;
    output_state.bits_per_pixel = bpp;
    states.emplace_back(output_state, SpeedCosts_Hardware);
  };

  if (input_state.bits_per_pixel == 8) {
    if (!input_state.has_alpha) {
      offer(heif_chroma_interleaved_RGB, false, 8);
    }
    offer(heif_chroma_interleaved_RGBA, true, 8);      // alpha filled with 0xFF when the input has none (yuv2rgb.cc:445)
  }
  else if (!input_state.has_alpha) {
    offer(heif_chroma_interleaved_RRGGBB_LE, false, input_state.bits_per_pixel);
    offer(heif_chroma_interleaved_RRGGBB_BE, false, input_state.bits_per_pixel);
    offer(heif_chroma_interleaved_RGB, false, 8);
    offer(heif_chroma_interleaved_RGBA, true, 8);
  }

  return states;
}


Result<std::shared_ptr<HeifPixelImage>>
Op_YCbCr_to_RGB_hip::convert_colorspace(const std::shared_ptr<const HeifPixelImage>& input,
                                        const ColorState& input_state,
                                        const ColorState& target_state,
                                        const heif_color_conversion_options& options,
                                        const heif_color_conversion_options_ext& options_ext,
                                        const heif_security_limits* limits) const
{
  const HipApi api = hip_api();
  if (!api.usable) {
    return Error::InternalError;
  }

  const int bpp = input->get_bits_per_pixel(heif_channel_Y);
  if (input->get_bits_per_pixel(heif_channel_Cb) != bpp ||
      input->get_bits_per_pixel(heif_channel_Cr) != bpp) {
    return Error::InternalError;
  }

  uint32_t width = input->get_width();
  uint32_t height = input->get_height();

  auto outimg = std::make_shared<HeifPixelImage>();
  outimg->create(width, height, heif_colorspace_RGB, target_state.chroma);

  if (auto err = outimg->add_channel(heif_channel_interleaved, width, height, target_state.bits_per_pixel, limits)) {
    return err;
  }

  hipdec_color_image img{};
  img.width = (int) width;
  img.height = (int) height;
  img.chroma = (int) input->get_chroma_format();   // heif_chroma_420 / 422 / 444 = 1 / 2 / 3
  img.bit_depth = bpp;
  img.plane[0] = input->get_channel_memory(heif_channel_Y, &img.stride[0]);
  img.plane[1] = input->get_channel_memory(heif_channel_Cb, &img.stride[1]);
  img.plane[2] = input->get_channel_memory(heif_channel_Cr, &img.stride[2]);
  if (input->has_channel(heif_channel_Alpha)) {
    img.plane[3] = input->get_channel_memory(heif_channel_Alpha, &img.stride[3]);
  }

  // the ops read the image's own colour profile (yuv2rgb.cc:368-375), the planner the ColorState's
  hipdec_nclx nclx{0, 2, 2, 2, 1};
  if (input->has_nclx_color_profile()) {
    nclx = to_hipdec(input->get_color_profile_nclx());
  }

  size_t out_stride = 0;
  uint8_t* out = outimg->get_channel_memory(heif_channel_interleaved, &out_stride);

  int rc = api.convert(&img, &nclx, (int) target_state.chroma, upsampling_of(options),
                       options.only_use_preferred_chroma_algorithm ? 1 : 0, out, out_stride, 0);
  if (rc != 0) {
    return Error{heif_error_Unsupported_feature, heif_suberror_Unsupported_color_conversion,
                 api.last_error ? api.last_error() : "HIP colour conversion failed"};
  }

  return outimg;
}
