// image_ops_hip.cc — see image_ops_hip.h.
//
// Everything here is a "try the backend first" wrapper: a declined, absent or failing backend leaves libheif's own code path in
// charge, so the patched library never decodes fewer files than the stock one, and error reporting for broken input stays the
// stock code's.  The backend's entry points arrive through heif_image_ops_register_hip_backend(), called by libheifhip.so's
// init_plugin() (the plugin is dlopen()ed with local symbol scope, libheif/plugins_unix.cc:103-118, so it announces itself).
#include "image_ops_hip.h"
#include "image/pixelimage.h"
#include "image-items/grid.h"
#include "image-items/image_item.h"
#include "codecs/decoder.h"
#include "context.h"
#include "file.h"
#include "plugin_registry.h"
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <future>
#include <mutex>
#include <vector>

namespace {

// restated from include/heif_hipdec.h (the C ABI of libheifhip.so)
struct hipdec_color_image {
  int width, height, chroma, bit_depth;
  const void* plane[4];
  size_t stride[4];
  int on_device;
};
struct hipdec_image_info {
  int width, height, chroma_format_idc, chroma_width, chroma_height, bit_depth_luma, bit_depth_chroma;
  int colour_primaries, transfer_characteristics, matrix_coeffs, full_range_flag;
  int coded_width, coded_height;
  size_t bitstream_bytes;
  int num_substreams;
};
enum { XF_ROTATE_CCW = 0, XF_MIRROR = 1, XF_CROP = 2 };

}  // namespace

extern "C" {
// the table libheifhip.so fills in (csrc/plugin.hip: announce_image_ops_backend)
struct heif_hip_image_ops_backend {
  int version;   // 1, or 2: forget_plane follows decoder_id
  int (*image_transform)(const void* in, int op, const int* args, void* out);
  int (*grid_create)(void** out, int rows, int cols, int out_width, int out_height, const void* const* tile_data, const size_t* tile_sizes,
                     const int* devices, int n_devices, uint64_t max_image_size_pixels);
  void (*grid_free)(void* g);
  int (*grid_info)(const void* g, void* info, int* n_shards);
  int (*grid_decode)(void* g);
  int (*grid_wait)(void* g);
  int (*grid_read_plane_tracked)(void* g, int c, void* dst_host, size_t dst_stride);
  const char* (*last_error)(void);
  const char* decoder_id;   // id_name of the decoder plugin these entry points belong to
  void (*forget_plane)(const void* host_plane);   // version 2: "this plane is about to be edited in place" (the device copy must not serve the conversion)
};
}

namespace {

std::mutex g_backend_mutex;
heif_hip_image_ops_backend g_backend{};
bool g_backend_usable = false;

bool backend(heif_hip_image_ops_backend* out)
{
  std::lock_guard<std::mutex> lock(g_backend_mutex);
  *out = g_backend;
  return g_backend_usable;
}

// The images the decoder plugin produces: Y (+ Cb, Cr) planes of one bit depth, 8..16 bits.  Anything else stays with the stock code.
bool describe(const HeifPixelImage& img, hipdec_color_image* d)
{
  const heif_colorspace cs = img.get_colorspace();
  const heif_chroma chroma = img.get_chroma_format();
  std::set<heif_channel> channels = img.get_channel_set();
  if (cs == heif_colorspace_monochrome) {
    if (channels != std::set<heif_channel>{heif_channel_Y}) return false;
  }
  else if (cs == heif_colorspace_YCbCr) {
    if (channels != std::set<heif_channel>{heif_channel_Y, heif_channel_Cb, heif_channel_Cr}) return false;
    if (chroma != heif_chroma_420 && chroma != heif_chroma_422 && chroma != heif_chroma_444) return false;
  }
  else {
    return false;
  }
  const int bits = img.get_bits_per_pixel(heif_channel_Y);
  if (bits < 8 || bits > 16) return false;
  for (heif_channel c : channels) {
    if (img.get_bits_per_pixel(c) != bits) return false;
  }
  *d = hipdec_color_image{};
  d->width = (int) img.get_width();
  d->height = (int) img.get_height();
  d->chroma = cs == heif_colorspace_monochrome ? 0 : (int) chroma;   // heif_chroma_420 / 422 / 444 = 1 / 2 / 3
  d->bit_depth = bits;
  d->plane[0] = img.get_channel_memory(heif_channel_Y, &d->stride[0]);
  if (cs == heif_colorspace_YCbCr) {
    d->plane[1] = img.get_channel_memory(heif_channel_Cb, &d->stride[1]);
    d->plane[2] = img.get_channel_memory(heif_channel_Cr, &d->stride[2]);
  }
  return d->plane[0] != nullptr;
}

// a new image of the input's format with planes of the given luma size; channel order as the decoder plugin adds them
std::shared_ptr<HeifPixelImage> image_like(const HeifPixelImage& img, uint32_t w, uint32_t h, const uint32_t plane_w[3], const uint32_t plane_h[3],
                                           const heif_security_limits* limits, hipdec_color_image* d)
{
  auto out = std::make_shared<HeifPixelImage>();
  out->create(w, h, img.get_colorspace(), img.get_chroma_format());
  out->copy_metadata_from(img);
  const heif_channel ch[3] = {heif_channel_Y, heif_channel_Cb, heif_channel_Cr};
  const int n = img.get_colorspace() == heif_colorspace_monochrome ? 1 : 3;
  *d = hipdec_color_image{};
  for (int c = 0; c < n; c++) {
    if (out->add_channel(ch[c], plane_w[c], plane_h[c], img.get_bits_per_pixel(ch[c]), limits)) {
      return nullptr;
    }
    d->plane[c] = out->get_channel_memory(ch[c], &d->stride[c]);
  }
  return out;
}

// the image through one operation of the backend; nullptr where the stock code has to run
std::shared_ptr<HeifPixelImage> transformed(const HeifPixelImage& img, int op, const int* args, uint32_t out_w, uint32_t out_h,
                                            const heif_security_limits* limits)
{
  heif_hip_image_ops_backend api;
  if (!backend(&api) || !api.image_transform) return nullptr;
  hipdec_color_image in;
  if (!describe(img, &in)) return nullptr;

  // plane sizes of the result: the planes of a subsampled image keep their own geometry through the operation (the cases where they
  // would not - odd sizes - are the ones the backend declines)
  uint32_t pw[3], ph[3];
  const heif_channel ch[3] = {heif_channel_Y, heif_channel_Cb, heif_channel_Cr};
  const int n = in.chroma == 0 ? 1 : 3;
  for (int c = 0; c < n; c++) {
    const uint32_t iw = img.get_width(ch[c]), ih = img.get_height(ch[c]);
    if (op == XF_ROTATE_CCW && args[0] != 180) { pw[c] = ih; ph[c] = iw; }
    else if (op == XF_CROP) {
      const uint32_t sx = c && (in.chroma == 1 || in.chroma == 2) ? 2 : 1, sy = c && in.chroma == 1 ? 2 : 1;
      pw[c] = (uint32_t) args[1] / sx - (uint32_t) args[0] / sx + 1;
      ph[c] = (uint32_t) args[3] / sy - (uint32_t) args[2] / sy + 1;
    }
    else { pw[c] = iw; ph[c] = ih; }
  }

  hipdec_color_image out;
  auto out_img = image_like(img, out_w, out_h, pw, ph, limits, &out);
  if (!out_img) return nullptr;
  if (api.image_transform(&in, op, args, &out) != 0) {
    return nullptr;   // HIPDEC_ERR_UNSUPPORTED (odd geometry: the reference converts to 4:4:4 first) or a device error
  }
  out_img->add_warnings(img.get_warnings());
  return out_img;
}

}  // namespace


extern "C" __attribute__((visibility("default")))
void heif_image_ops_register_hip_backend(const heif_hip_image_ops_backend* api, int usable)
{
  std::lock_guard<std::mutex> lock(g_backend_mutex);
  g_backend = heif_hip_image_ops_backend{};
  if (api && api->version == 2) g_backend = *api;
  else if (api && api->version == 1) { memcpy(&g_backend, api, offsetof(heif_hip_image_ops_backend, forget_plane)); g_backend.forget_plane = nullptr; }
  g_backend_usable = api && (api->version == 1 || api->version == 2) && usable;
}

// What these hooks promise the backend.  Bit 0: every in-place edit of a decoded image between the plugin's hand-over and the colour conversion is
// announced through forget_plane() first (the mirror fall-back below is the only one: rotation and cropping make new images, image_item.cc:958-1004),
// so the backend may identify a handed-over plane cheaply instead of hashing every byte of it.
extern "C" __attribute__((visibility("default")))
int heif_image_ops_hip_capabilities(void) { return 1; }


namespace hip_image_ops {

Result<std::shared_ptr<HeifPixelImage>> rotate_ccw(const std::shared_ptr<HeifPixelImage>& img, int angle_degrees,
                                                   const heif_security_limits* limits)
{
  if (angle_degrees == 90 || angle_degrees == 180 || angle_degrees == 270) {
    const bool swap = angle_degrees != 180;
    const int args[1] = {angle_degrees};
    if (auto out = transformed(*img, XF_ROTATE_CCW, args, swap ? img->get_height() : img->get_width(),
                               swap ? img->get_width() : img->get_height(), limits)) {
      return out;
    }
  }
  return img->rotate_ccw(angle_degrees, limits);
}


Result<std::shared_ptr<HeifPixelImage>> mirror(const std::shared_ptr<HeifPixelImage>& img, heif_transform_mirror_direction direction,
                                               const heif_security_limits* limits)
{
  if (direction == heif_transform_mirror_direction_vertical || direction == heif_transform_mirror_direction_horizontal) {
    const int args[1] = {(int) direction};
    if (auto out = transformed(*img, XF_MIRROR, args, img->get_width(), img->get_height(), limits)) {
      return out;
    }
  }
  // the stock code mirrors IN PLACE (same plane pointers, new content): the backend's device copies of these planes are stale from here on
  {
    heif_hip_image_ops_backend api;
    if (backend(&api) && api.forget_plane) {
      for (heif_channel ch : {heif_channel_Y, heif_channel_Cb, heif_channel_Cr, heif_channel_Alpha, heif_channel_R, heif_channel_G, heif_channel_B, heif_channel_interleaved}) {
        if (img->has_channel(ch)) { size_t stride = 0; api.forget_plane(img->get_channel_memory(ch, &stride)); }
      }
    }
  }
  return img->mirror_inplace(direction, limits);
}


Result<std::shared_ptr<HeifPixelImage>> crop(const std::shared_ptr<HeifPixelImage>& img, uint32_t left, uint32_t right, uint32_t top,
                                             uint32_t bottom, const heif_security_limits* limits)
{
  if (left <= right && top <= bottom && right < img->get_width() && bottom < img->get_height()) {
    const int args[4] = {(int) left, (int) right, (int) top, (int) bottom};
    if (auto out = transformed(*img, XF_CROP, args, right - left + 1, bottom - top + 1, limits)) {
      return out;
    }
  }
  return img->crop(left, right, top, bottom, limits);
}


std::shared_ptr<HeifPixelImage> decode_grid(const ImageItem_Grid& grid_item, const heif_decoding_options& options,
                                            const std::set<heif_item_id>& processed_ids)
{
  heif_hip_image_ops_backend api;
  if (!backend(&api) || !api.grid_create || !api.grid_decode || !api.grid_wait || !api.grid_info || !api.grid_read_plane_tracked ||
      !api.grid_free || !api.decoder_id) {
    return nullptr;
  }

  // the decoder libheif would pick for the tiles has to be the one these entry points belong to
  const heif_decoder_plugin* plugin = get_decoder(heif_compression_HEVC, options.decoder_id);
  if (!plugin || !plugin->id_name || strcmp(plugin->id_name, api.decoder_id) != 0) {
    return nullptr;
  }

  const HeifContext* ctx = grid_item.get_context();
  const ImageGrid& grid = grid_item.get_grid_spec();
  const std::vector<heif_item_id>& ids = grid_item.get_grid_tiles();
  const size_t n_tiles = (size_t) grid.get_rows() * grid.get_columns();
  if (n_tiles == 0 || ids.size() != n_tiles) {
    return nullptr;
  }

  // --- every tile: a plain 'hvc1' item of one size, no alpha, no transformation of its own

  std::vector<std::shared_ptr<const ImageItem>> tiles;
  std::vector<std::vector<uint8_t>> streams(n_tiles);
  uint32_t tile_w = 0, tile_h = 0;
  for (size_t t = 0; t < n_tiles; t++) {
    if (processed_ids.contains(ids[t])) return nullptr;
    std::shared_ptr<const ImageItem> tile = ctx->get_image(ids[t], true);
    if (!tile || tile->get_item_error() || tile->get_infe_type() != fourcc("hvc1") || tile->get_alpha_channel()) {
      return nullptr;
    }
    if (t == 0) { tile_w = tile->get_width(); tile_h = tile->get_height(); }
    else if (tile->get_width() != tile_w || tile->get_height() != tile_h) return nullptr;
    if (!options.ignore_transformations) {
      auto props = tile->get_properties();
      if (!props) return nullptr;
      for (const auto& p : *props) {
        if (std::dynamic_pointer_cast<Box_irot>(p) || std::dynamic_pointer_cast<Box_imir>(p) || std::dynamic_pointer_cast<Box_clap>(p) ||
            std::dynamic_pointer_cast<Box_iscl>(p)) {
          return nullptr;
        }
      }
    }
    tiles.push_back(tile);
  }
  if ((uint64_t) tile_w * grid.get_columns() < grid.get_width() || (uint64_t) tile_h * grid.get_rows() < grid.get_height()) {
    return nullptr;   // "Grid tiles do not cover whole image": the stock loop reports it
  }

  // --- the tile streams in the plugin's framing: hvcC parameter sets + item data (Decoder::get_compressed_data, codecs/decoder.cc:275-299)

  for (size_t t = 0; t < n_tiles; t++) {
    auto conf = tiles[t]->read_bitstream_configuration_data();
    if (!conf) return nullptr;
    DataExtent extent;
    extent.set_from_image_item(tiles[t]->get_file(), tiles[t]->get_id());
    auto data = extent.read_data();
    if (!data || (*data)->empty()) return nullptr;
    streams[t] = std::move(*conf);
    streams[t].insert(streams[t].end(), (*data)->begin(), (*data)->end());
  }

  if (options.cancel_decoding && options.cancel_decoding(options.progress_user_data)) {
    return nullptr;   // the stock loop turns this into heif_error_Canceled
  }

  // --- tile 0 through the ordinary path, concurrently with the grid decode below: the composed image takes its format and every piece
  //     of metadata from that image (decode_and_paste_tile_image, grid.cc:531-556: create_clone_image_at_new_size + copy_metadata_from)

  auto first_future = std::async(std::launch::async, [&]() { return tiles[0]->decode_image(options, false, 0, 0, processed_ids); });
  struct Join { std::future<Result<std::shared_ptr<HeifPixelImage>>>& f; ~Join() { if (f.valid()) f.wait(); } } join_first{first_future};

  // --- all tiles on the GPUs of this node, pasted into one canvas on the device

  const heif_security_limits* limits = ctx->get_security_limits();
  std::vector<const void*> ptrs(n_tiles);
  std::vector<size_t> sizes(n_tiles);
  for (size_t t = 0; t < n_tiles; t++) { ptrs[t] = streams[t].data(); sizes[t] = streams[t].size(); }

  void* g = nullptr;
  if (api.grid_create(&g, (int) grid.get_rows(), (int) grid.get_columns(), (int) grid.get_width(), (int) grid.get_height(), ptrs.data(), sizes.data(),
                      nullptr, 0, limits ? limits->max_image_size_pixels : 0) != 0 || !g) {
    return nullptr;
  }
  struct Free { decltype(api.grid_free) f; void* g; ~Free() { f(g); } } free_grid{api.grid_free, g};

  hipdec_image_info info{};
  int shards = 0;
  if (api.grid_info(g, &info, &shards) != 0 || api.grid_decode(g) != 0) {
    return nullptr;
  }

  auto first = first_future.get();
  if (!first || !*first) return nullptr;
  const std::shared_ptr<HeifPixelImage>& tile_img = *first;
  hipdec_color_image fmt;
  if (!describe(*tile_img, &fmt) || tile_img->get_width() != tile_w || tile_img->get_height() != tile_h) {
    return nullptr;
  }

  if (info.coded_width != (int) (tile_w * grid.get_columns()) ||
      info.coded_height != (int) (tile_h * grid.get_rows()) || info.chroma_format_idc != fmt.chroma || info.bit_depth_luma != fmt.bit_depth ||
      (fmt.chroma && info.bit_depth_chroma != fmt.bit_depth)) {
    return nullptr;   // tiles that differ from their 'ispe' / from tile 0: the stock loop reports it
  }

  if (api.grid_wait(g) != 0) {
    return nullptr;
  }

  auto canvas = std::make_shared<HeifPixelImage>();
  if (canvas->create_clone_image_at_new_size(tile_img, grid.get_width(), grid.get_height(), limits)) {
    return nullptr;
  }
  canvas->copy_metadata_from(*tile_img);

  const heif_channel ch[3] = {heif_channel_Y, heif_channel_Cb, heif_channel_Cr};
  for (int c = 0; c < (fmt.chroma ? 3 : 1); c++) {
    size_t stride = 0;
    uint8_t* mem = canvas->get_channel_memory(ch[c], &stride);
    if (!mem || api.grid_read_plane_tracked(g, c, mem, stride) != 0) {
      return nullptr;
    }
  }

  // the progress protocol of the stock loop (grid.cc:302-307, :470-480, :455-457), reported once the canvas is complete
  if (options.start_progress) options.start_progress(heif_progress_step_total, (int) n_tiles, options.progress_user_data);
  if (options.on_progress) {
    for (size_t t = 0; t <= n_tiles; t++) options.on_progress(heif_progress_step_total, (int) t, options.progress_user_data);
  }
  if (options.end_progress) options.end_progress(heif_progress_step_total, options.progress_user_data);

  return canvas;
}

}  // namespace hip_image_ops
