#!/usr/bin/env python3
"""The libheif side of the integration as a patch: exact-text edits of three reference translation units, applied at build time
(oracle/Makefile.ref writes the results under oracle/_ref/gen/; no reference source enters this repository).

    apply_patch.py colorconversion <libheif/color-conversion/colorconversion.cc> <out.cc>
    apply_patch.py image_item      <libheif/image-items/image_item.cc>           <out.cc>
    apply_patch.py grid            <libheif/image-items/grid.cc>                 <out.cc>

Every edit must match exactly once: a reference that has moved on makes the build fail here instead of silently dropping a hook."""
import sys

EDITS = {
    # ColorConversionPipeline::init_ops() also registers Op_YCbCr_to_RGB_hip (colorconversion_hip.cc)
    "colorconversion": [
        ('#include "colorconversion.h"\n',
         '#include "colorconversion.h"\n#include "colorconversion_hip.h"\n'),
        ('  ops.emplace_back(std::make_shared<Op_RGB_to_RGB24_32>());\n',
         '  ops.emplace_back(std::make_shared<Op_YCbCr_to_RGB_hip>());\n  ops.emplace_back(std::make_shared<Op_RGB_to_RGB24_32>());\n'),
    ],
    # ImageItem::decode_image(): 'irot' / 'imir' / 'clap' try the backend first (image_ops_hip.cc)
    "image_item": [
        ('#include "image_item.h"\n',
         '#include "image_item.h"\n#include "image_ops_hip.h"\n'),
        ('img->rotate_ccw(rot->get_rotation_ccw(), m_heif_context->get_security_limits());',
         'hip_image_ops::rotate_ccw(img, rot->get_rotation_ccw(), m_heif_context->get_security_limits());'),
        ('img->mirror_inplace(mirror->get_mirror_direction(),\n                                                get_context()->get_security_limits());',
         'hip_image_ops::mirror(img, mirror->get_mirror_direction(), get_context()->get_security_limits());'),
        ('img->crop(left, right, top, bottom, m_heif_context->get_security_limits());',
         'hip_image_ops::crop(img, left, right, top, bottom, m_heif_context->get_security_limits());'),
    ],
    # ImageItem_Grid::decode_full_grid_image(): all tiles to the backend in one call before the per-tile loop (image_ops_hip.cc)
    "grid": [
        ('#include "grid.h"\n',
         '#include "grid.h"\n#include "image_ops_hip.h"\n'),
        ('  uint32_t y0 = 0;\n  int reference_idx = 0;\n',
         '  if (auto composed = hip_image_ops::decode_grid(*this, options, processed_ids)) {\n    return composed;\n  }\n\n'
         '  uint32_t y0 = 0;\n  int reference_idx = 0;\n'),
    ],
}


def main():
    which, src, dst = sys.argv[1:4]
    text = open(src).read()
    for old, new in EDITS[which]:
        if text.count(old) != 1:
            sys.exit("apply_patch.py %s: expected exactly one occurrence of %r in %s, found %d" % (which, old, src, text.count(old)))
        text = text.replace(old, new)
    open(dst, "w").write(text)


if __name__ == "__main__":
    main()
