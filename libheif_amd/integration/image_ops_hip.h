// image_ops_hip.h — the image-level hooks of the HIP backend inside libheif (the second half of the integration patch; the
// first half is the colour op, colorconversion_hip.h).
//
// What a libheif maintainer adds next to libheif/image-items/image_item.cc and grid.cc:
//   * image_item.cc:958-1004 — the three transformative properties call hip_image_ops::rotate_ccw / mirror / crop instead of
//     HeifPixelImage::rotate_ccw / mirror_inplace / crop.  Each tries libheifhip.so's hipdec_image_transform() first (the planes
//     the decoder plugin has just handed over are found on the device; the result stays there behind its host copy for the next
//     property and for the colour conversion) and runs the stock member function wherever the backend declines
//     (HIPDEC_ERR_UNSUPPORTED: odd sizes of subsampled images that the reference converts to 4:4:4 first), is absent, or fails;
//   * grid.cc:250-468 — ImageItem_Grid::decode_full_grid_image asks hip_image_ops::decode_grid() before its per-tile loop: when
//     every tile is a plain HEVC item the backend decodes, all tile streams go to hipdec_grid_* in one call (tile t -> GPU
//     t mod n_devices, pasted device-to-device into one canvas), and the composed image is filled from the canvas — no
//     per-tile plugin round trip, no host paste (decode_and_paste_tile_image :482-577 / HeifPixelImage::copy_image_to).  A null
//     result means "not taken": the stock loop runs and reports errors exactly as before.
#pragma once
#include <memory>
#include <set>
#include "error.h"
#include "libheif/heif.h"
#include "libheif/heif_properties.h"

class HeifPixelImage;
class ImageItem_Grid;

namespace hip_image_ops {

Result<std::shared_ptr<HeifPixelImage>> rotate_ccw(const std::shared_ptr<HeifPixelImage>& img, int angle_degrees,
                                                   const heif_security_limits* limits);

Result<std::shared_ptr<HeifPixelImage>> mirror(const std::shared_ptr<HeifPixelImage>& img, heif_transform_mirror_direction direction,
                                               const heif_security_limits* limits);

Result<std::shared_ptr<HeifPixelImage>> crop(const std::shared_ptr<HeifPixelImage>& img, uint32_t left, uint32_t right, uint32_t top,
                                             uint32_t bottom, const heif_security_limits* limits);

// nullptr: the fast path does not apply (or failed) — the caller runs the stock tile loop
std::shared_ptr<HeifPixelImage> decode_grid(const ImageItem_Grid& grid, const heif_decoding_options& options,
                                            const std::set<heif_item_id>& processed_ids);

}  // namespace hip_image_ops
