// plugin.hip — libheif decoder plugin: the drop-in replacement for libheif/plugins/decoder_libde265.cc.
//
// libheif dlopen()s this shared object from LIBHEIF_PLUGIN_PATH and reads the exported
// `plugin_info` (libheif/plugins_unix.cc:103-118; example export decoder_libde265.cc:528-534).  Every
// slot of heif_decoder_plugin is filled (a NULL new_decoder makes libheif report a dummy plugin,
// libheif/codecs/decoder.cc:389-392).  The calls back into libheif (heif_image_create & co.) are
// resolved with dlsym() from the hosting process, so the library also loads without libheif (tests,
// other hosts); in that case the plugin functions fail loudly.
#include "hipdec_internal.h"
#include "heif_plugin_abi.h"
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

namespace {

// ---- libheif public API used by a decoder plugin (resolved at run time) ----
struct HostApi {
  hp_error (*image_create)(int w, int h, int colorspace, int chroma, hp_image** out) = nullptr;
  hp_error (*image_add_plane_safe)(hp_image*, int channel, int w, int h, int bit_depth, const void* limits) = nullptr;
  uint8_t* (*image_get_plane2)(hp_image*, int channel, size_t* stride) = nullptr;
  void (*image_release)(const hp_image*) = nullptr;
  hp_nclx_head* (*nclx_alloc)(void) = nullptr;
  void (*nclx_free)(hp_nclx_head*) = nullptr;
  hp_error (*nclx_set_primaries)(hp_nclx_head*, uint16_t) = nullptr;
  hp_error (*nclx_set_transfer)(hp_nclx_head*, uint16_t) = nullptr;
  hp_error (*nclx_set_matrix)(hp_nclx_head*, uint16_t) = nullptr;
  hp_error (*image_set_nclx)(hp_image*, const hp_nclx_head*) = nullptr;
  void (*image_add_warning)(hp_image*, hp_error) = nullptr;
  const void* (*get_global_limits)(void) = nullptr;
  bool ok = false;
};
HostApi g_api;
std::once_flag g_api_once;

void resolve_api()
{
  auto sym = [](const char* n) { return dlsym(RTLD_DEFAULT, n); };
  g_api.image_create = (decltype(g_api.image_create))sym("heif_image_create");
  g_api.image_add_plane_safe = (decltype(g_api.image_add_plane_safe))sym("heif_image_add_plane_safe");
  g_api.image_get_plane2 = (decltype(g_api.image_get_plane2))sym("heif_image_get_plane2");
  g_api.image_release = (decltype(g_api.image_release))sym("heif_image_release");
  g_api.nclx_alloc = (decltype(g_api.nclx_alloc))sym("heif_nclx_color_profile_alloc");
  g_api.nclx_free = (decltype(g_api.nclx_free))sym("heif_nclx_color_profile_free");
  g_api.nclx_set_primaries = (decltype(g_api.nclx_set_primaries))sym("heif_nclx_color_profile_set_color_primaries");
  g_api.nclx_set_transfer = (decltype(g_api.nclx_set_transfer))sym("heif_nclx_color_profile_set_transfer_characteristics");
  g_api.nclx_set_matrix = (decltype(g_api.nclx_set_matrix))sym("heif_nclx_color_profile_set_matrix_coefficients");
  g_api.image_set_nclx = (decltype(g_api.image_set_nclx))sym("heif_image_set_nclx_color_profile");
  g_api.image_add_warning = (decltype(g_api.image_add_warning))sym("heif_image_add_decoding_warning");
  g_api.get_global_limits = (decltype(g_api.get_global_limits))sym("heif_get_global_security_limits");
  g_api.ok = g_api.image_create && g_api.image_add_plane_safe && g_api.image_get_plane2 && g_api.image_release && g_api.nclx_alloc &&
             g_api.nclx_free && g_api.image_set_nclx;
}

const char kSuccess[] = "Success";
const char kNoHost[] = "libheif-hipdec: the hosting process does not export the libheif image API";
const hp_error kOk = {HP_ERR_OK, HP_SUB_UNSPECIFIED, kSuccess};

struct PluginDecoder {
  hipdec_decoder* dec = nullptr;
  int strict = 0;
  const void* limits = nullptr;
  bool flushed = false;       // flush_data was called: the pictures still waiting for output come out (C.5.2.2)
  std::string error_message;  // keeps messages alive beyond the call (decoder_libde265.cc:150-156)
};

hp_error make_error(PluginDecoder* d, int rc)
{
  d->error_message = hipdec_last_error();
  hp_error e;
  e.message = d->error_message.c_str();
  switch (rc) {
    case HIPDEC_ERR_END_OF_DATA: e.code = HP_ERR_DECODER_PLUGIN; e.subcode = HP_SUB_END_OF_DATA; break;
    case HIPDEC_ERR_UNSUPPORTED: e.code = HP_ERR_UNSUPPORTED_FEATURE; e.subcode = HP_SUB_UNSUPPORTED_CODEC; break;
    case HIPDEC_ERR_LIMIT: e.code = HP_ERR_MEMORY; e.subcode = HP_SUB_SECURITY_LIMIT; break;
    default: e.code = HP_ERR_DECODER_PLUGIN; e.subcode = HP_SUB_UNSPECIFIED; break;
  }
  return e;
}

const char* plugin_name() { return "MI355X HIP HEVC decoder (libheif-hipdec), gfx950"; }
// Hands the colour boundary's entry points to a libheif that carries the HIP colour op (libheif_amd/integration/colorconversion_hip.cc):
// the plugin is loaded with local symbol scope, so the op cannot look them up itself.  A stock libheif has no such symbol: nothing
// happens and its CPU colour ops run.
void announce_color_backend()
{
  using reg_fn = void (*)(int (*)(int, int, int, const hipdec_nclx*, int, int, int, int*, int*),
                          int (*)(const hipdec_color_image*, const hipdec_nclx*, int, int, int, void*, size_t, int), const char* (*)(void), int);
  if (auto reg = (reg_fn)dlsym(RTLD_DEFAULT, "heif_color_conversion_register_hip_backend")) {
    reg(hipdec_color_plan, hipdec_color_convert, hipdec_last_error, hipdec_device_count() > 0 ? 1 : 0);
    hipdec_set_plane_tracking(1);   // this libheif converts on the GPU: keep decoded planes findable on the device (costs a hash pass per plane)
    hipdec_set_reserved_wave_slots(1);   // ... and keep a wave slot per SIMD for those kernels beside the resident CABAC pools
  }
}
// The same for the image-level hooks of that libheif (libheif_amd/integration/image_ops_hip.cc): 'irot' / 'imir' / 'clap' through
// hipdec_image_transform, 'grid' items through hipdec_grid_*.  The table's layout is restated there.
struct ImageOpsBackend {
  int version;
  int (*image_transform)(const hipdec_color_image*, int, const int*, hipdec_color_image*);
  int (*grid_create)(hipdec_grid**, int, int, int, int, const void* const*, const size_t*, const int*, int, uint64_t);
  void (*grid_free)(hipdec_grid*);
  int (*grid_info)(const hipdec_grid*, hipdec_image_info*, int*);
  int (*grid_decode)(hipdec_grid*);
  int (*grid_wait)(hipdec_grid*);
  int (*grid_read_plane_tracked)(hipdec_grid*, int, void*, size_t);
  const char* (*last_error)(void);
  const char* decoder_id;
  void (*forget_plane)(const void*);   // version 2: the host announces an in-place edit of a plane the decoder handed over
};
const char kPluginId[] = "hipdec";
void announce_image_ops_backend()
{
  using reg_fn = void (*)(const ImageOpsBackend*, int);
  if (auto reg = (reg_fn)dlsym(RTLD_DEFAULT, "heif_image_ops_register_hip_backend")) {
    // a host whose hooks announce their in-place edits (heif_image_ops_hip_capabilities() bit 0) gets table version 2 and the cheap plane identity
    using cap_fn = int (*)(void);
    const auto caps = (cap_fn)dlsym(RTLD_DEFAULT, "heif_image_ops_hip_capabilities");
    const bool announces = caps && (caps() & 1) && !getenv("HIPDEC_PLANE_IDENTITY_FULL");   // (A/B knob: the full hash although the host announces)
    static const ImageOpsBackend table1 = {1, hipdec_image_transform, hipdec_grid_create, hipdec_grid_free, hipdec_grid_info, hipdec_grid_decode,
                                           hipdec_grid_wait, hipdec_grid_read_plane_tracked, hipdec_last_error, kPluginId, nullptr};
    static const ImageOpsBackend table2 = {2, hipdec_image_transform, hipdec_grid_create, hipdec_grid_free, hipdec_grid_info, hipdec_grid_decode,
                                           hipdec_grid_wait, hipdec_grid_read_plane_tracked, hipdec_last_error, kPluginId, hipdec_forget_plane};
    reg(announces ? &table2 : &table1, hipdec_device_count() > 0 ? 1 : 0);
    hipdec_set_plane_tracking(announces ? 2 : 1);
    hipdec_set_reserved_wave_slots(1);
  }
}
void init_plugin()
{
  std::call_once(g_api_once, resolve_api);
  announce_color_backend();
  announce_image_ops_backend();
}
void deinit_plugin() { hipdec_forget_resident_planes(); }   // heif_deinit(): nothing of ours may outlive the host's use of the library
int does_support_format(int format) { return format == HP_COMPRESSION_HEVC ? 200 /* above libde265's 100 */ : 0; }
int does_support_format2(const hp_format_description* f) { return f ? does_support_format(f->format) : 0; }

hp_error new_decoder2(void** out, const hp_decoder_options* opt)
{
  std::call_once(g_api_once, resolve_api);
  static thread_local std::string msg;
  uint64_t max_px = 0;
  const void* limits = opt ? opt->limits : nullptr;
  if (!limits && g_api.get_global_limits) limits = g_api.get_global_limits();
  if (limits) max_px = ((const hp_security_limits_head*)limits)->max_image_size_pixels;
  hipdec_decoder* dec = nullptr;
  int rc = hipdec_decoder_new(&dec, opt ? opt->strict_decoding : 0, max_px);
  if (rc) { msg = hipdec_last_error(); return hp_error{HP_ERR_DECODER_PLUGIN, HP_SUB_UNSPECIFIED, msg.c_str()}; }
  PluginDecoder* d = new PluginDecoder();
  d->dec = dec; d->strict = opt ? opt->strict_decoding : 0; d->limits = limits;
  *out = d;
  return kOk;
}
hp_error new_decoder(void** out)
{
  hp_decoder_options o{HP_COMPRESSION_HEVC, 0, 0, nullptr};
  return new_decoder2(out, &o);
}
void free_decoder(void* p)
{
  PluginDecoder* d = (PluginDecoder*)p;
  if (!d) return;
  hipdec_decoder_free(d->dec);
  delete d;
}
void set_strict_decoding(void* p, int flag)
{
  PluginDecoder* d = (PluginDecoder*)p;
  d->strict = flag;
  hipdec_decoder_set_strict(d->dec, flag);
}
hp_error push_data2(void* p, const void* data, size_t size, uintptr_t user_data)
{
  PluginDecoder* d = (PluginDecoder*)p;
  d->flushed = false;
  int rc = hipdec_decoder_push_data(d->dec, data, size);
  if (!rc) hipdec_decoder_set_user_data(d->dec, user_data);   // handed back with the picture this data decodes to (decoder_libde265.cc:360, :417-419)
  return rc ? make_error(d, rc) : kOk;
}
hp_error push_data(void* p, const void* data, size_t size) { return push_data2(p, data, size, 0); }
hp_error flush_data(void* p) { if (p) ((PluginDecoder*)p)->flushed = true; return kOk; }

hp_error decode_next_image2(void* p, hp_image** out_img, uintptr_t* out_user_data, const void* limits)
{
  PluginDecoder* d = (PluginDecoder*)p;
  *out_img = nullptr;
  if (out_user_data) *out_user_data = 0;
  if (!g_api.ok) return hp_error{HP_ERR_DECODER_PLUGIN, HP_SUB_UNSPECIFIED, kNoHost};
  hipdec_image_info info;
  int have = 0;
  uintptr_t picture_user_data = 0;
  int rc = hipdec_decoder_next_picture(d->dec, d->flushed ? 1 : 0, &info, &have, &picture_user_data);
  if (rc == HIPDEC_ERR_NO_IMAGE) return kOk;  // "nothing (more) to deliver": *out_img stays NULL
  if (rc) return make_error(d, rc);
  if (!have) return kOk;                      // decoded, but an earlier picture in output order is still to come (B pictures): push the next sample
  const bool mono = info.chroma_format_idc == 0;
  hp_image* img = nullptr;
  hp_error err = g_api.image_create(info.width, info.height, mono ? HP_COLORSPACE_MONOCHROME : HP_COLORSPACE_YCBCR, info.chroma_format_idc, &img);
  if (err.code) return err;
  if (info.bit_depth_luma != info.bit_depth_chroma && !mono) {
    g_api.image_release(img);
    return hp_error{HP_ERR_UNSUPPORTED_FEATURE, HP_SUB_UNSPECIFIED, "Channels with different number of bits per pixel are not supported"};
  }
  static const int channel[3] = {HP_CHANNEL_Y, HP_CHANNEL_CB, HP_CHANNEL_CR};
  for (int c = 0; c < (mono ? 1 : 3); c++) {
    const int w = c ? info.chroma_width : info.width, h = c ? info.chroma_height : info.height;
    err = g_api.image_add_plane_safe(img, channel[c], w, h, info.bit_depth_luma, limits ? limits : d->limits);
    if (err.code) {
      d->error_message = err.message ? err.message : "";
      err.message = d->error_message.c_str();
      g_api.image_release(img);
      return err;
    }
    size_t stride = 0;
    uint8_t* dst = g_api.image_get_plane2(img, channel[c], &stride);
    rc = hipdec_decoder_read_plane_tracked(d->dec, c, dst, stride);   // D2H straight into libheif's plane; the device copy stays findable
    if (rc) { g_api.image_release(img); return make_error(d, rc); }
  }
  // VUI colour description -> nclx, as decoder_libde265.cc:426-449
  // Each setter rejects values outside the enumerations it knows; like HEIF_WARN_OR_FAIL (heif_plugin.h:369-380) that is an
  // error under strict decoding (image released, *out_img NULL) and a decoding warning on the image otherwise.
  hp_nclx_head* nclx = g_api.nclx_alloc();
  if (nclx) {
    hp_error (*const setters[3])(hp_nclx_head*, uint16_t) = {g_api.nclx_set_primaries, g_api.nclx_set_transfer, g_api.nclx_set_matrix};
    const int values[3] = {info.colour_primaries, info.transfer_characteristics, info.matrix_coeffs};
    for (int k = 0; k < 3; k++) {
      if (!setters[k]) continue;
      hp_error e = setters[k](nclx, (uint16_t)values[k]);
      if (e.code == HP_ERR_OK) continue;
      if (d->strict) {
        g_api.nclx_free(nclx);
        g_api.image_release(img);
        return e;                      // the setters return static message strings
      }
      if (g_api.image_add_warning) g_api.image_add_warning(img, e);
    }
    nclx->full_range_flag = (uint8_t)info.full_range_flag;
    g_api.image_set_nclx(img, nclx);
    g_api.nclx_free(nclx);
  }
  if (out_user_data) *out_user_data = picture_user_data;
  *out_img = img;
  return kOk;
}
hp_error decode_next_image(void* p, hp_image** out, const void* limits) { return decode_next_image2(p, out, nullptr, limits); }
hp_error decode_image(void* p, hp_image** out) { return decode_next_image2(p, out, nullptr, g_api.get_global_limits ? g_api.get_global_limits() : nullptr); }

// plugin_api_version 6: new_decoder2 reads heif_decoder_plugin_options.limits, which only exists from version 6 on
// (heif_plugin.h:70-82, "only read by plugins reporting plugin_api_version >= 6"); libheif 1.22.0 is the first release with it
// (version table heif_plugin.h:40-48) and rejects plugins newer than it knows (heif_library.cc:75).
const hp_decoder_plugin g_plugin = {
    6,
    plugin_name,
    init_plugin,
    deinit_plugin,
    does_support_format,
    new_decoder,
    free_decoder,
    push_data,
    decode_image,
    set_strict_decoding,
    kPluginId,
    decode_next_image,
    (1u << 24) | (22u << 16),  // LIBHEIF_MAKE_VERSION(1,22,0): first release with plugin API 6
    does_support_format2,
    new_decoder2,
    push_data2,
    flush_data,
    decode_next_image2};

}  // namespace

extern "C" {
// Found by libheif with dlsym(handle, "plugin_info") (libheif/plugins_unix.cc:111).
HIPDEC_API hp_plugin_info plugin_info = {1, HP_PLUGIN_TYPE_DECODER, &g_plugin, nullptr};
// For hosts that register statically: heif_register_decoder_plugin(hipdec_get_decoder_plugin())
HIPDEC_API const void* hipdec_get_decoder_plugin(void) { return &g_plugin; }
}
