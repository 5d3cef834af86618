// colorconversion_hip.h — the reference-side binding of the colour boundary: a ColorConversionOperation that hands HEIC colour
// conversions to libheifhip.so (hipdec_color_convert, include/heif_hipdec.h).
//
// This is the file a libheif maintainer adds next to libheif/color-conversion/yuv2rgb.h; the one-line registration goes into
// ColorConversionPipeline::init_ops() (libheif/color-conversion/colorconversion.cc:235-269):
//
//     ops.emplace_back(std::make_shared<Op_YCbCr_to_RGB_hip>());
//
// It is compiled INTO libheif (against its internal headers), not into libheifhip.so; oracle/Makefile.ref builds a libheif with
// it for the tests (oracle/_ref/libheif_hipcolor.so).  Interface: libheif/color-conversion/colorconversion.h:78-100.
#ifndef LIBHEIF_COLORCONVERSION_HIP_H
#define LIBHEIF_COLORCONVERSION_HIP_H

#include "colorconversion.h"

class Op_YCbCr_to_RGB_hip : public ColorConversionOperation
{
public:
  std::vector<ColorStateWithCost>
  state_after_conversion(const ColorState& input_state,
                         const ColorState& target_state,
                         const heif_color_conversion_options& options,
                         const heif_color_conversion_options_ext& options_ext) const override;

  Result<std::shared_ptr<HeifPixelImage>>
  convert_colorspace(const std::shared_ptr<const HeifPixelImage>& input,
                     const ColorState& input_state,
                     const ColorState& target_state,
                     const heif_color_conversion_options& options,
                     const heif_color_conversion_options_ext& options_ext,
                     const heif_security_limits* limits) const override;
};

#endif
