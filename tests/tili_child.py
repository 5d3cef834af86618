"""Child process of tests/test_tili_gpu.py: the experimental-features build of the reference libheif (HIPDEC_TEST_LIBHEIF =
libheif_experimental.so: the only configuration that instantiates 'tili' items) with libheifhip.so as decoder plugin; decodes the
requested tiles of a 'tili' file with heif_image_handle_decode_image_tile() and stores their planes.  A process of its own because every
build of the reference exports the same symbols."""
import ctypes as C
import json
import sys
import numpy as np

import libheif_host as lh


def main():
    job = json.load(open(sys.argv[1]))
    L = lh.load_hip_plugin()
    from libheif_amd.decoder import coalesce_stats
    data = open(job["file"], "rb").read()
    ctx, h = lh.open_heic(data)
    L.heif_image_handle_decode_image_tile.restype = lh.HeifError
    L.heif_image_handle_decode_image_tile.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
    out = {"size": np.array([L.heif_image_handle_get_width(h), L.heif_image_handle_get_height(h)])}
    for tx, ty in job["tiles"]:
        before = coalesce_stats()[0]
        img = C.c_void_p()
        lh.check(L.heif_image_handle_decode_image_tile(h, C.byref(img), lh.COLORSPACE_YCBCR, lh.CHROMA_420, None, tx, ty))
        out["decodes_%d_%d" % (tx, ty)] = np.array([coalesce_stats()[0] - before])
        for c, ch in enumerate((lh.CHANNEL_Y, lh.CHANNEL_CB, lh.CHANNEL_CR)):
            bps = 2 if L.heif_image_get_bits_per_pixel_range(img, ch) > 8 else 1
            out["t%d_%d_c%d" % (tx, ty, c)] = lh._plane(L, img, ch, bytes_per_sample=bps)
        L.heif_image_release(img)
    # the whole image at once is refused by the reference itself ('tili' images can only be accessed per tile, tiled.cc)
    img = C.c_void_p()
    e = L.heif_decode_image(h, C.byref(img), lh.COLORSPACE_YCBCR, lh.CHROMA_420, None)
    out["whole_image_error_code"] = np.array([e.code])
    L.heif_image_handle_release(h)
    L.heif_context_free(ctx)
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
