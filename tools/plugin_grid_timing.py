"""Wall time of heif_decode_image() on a synthetic grid HEIC through the REAL reference libheif with libheifhip.so as
the decoder plugin, for several values of heif_context_set_max_decoding_threads (GPU box only; dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools import streamgen
import heic_util as hu
import libheif_host as lh

rows, cols, tw, th = 6, 8, 512, 512          # an iPhone-like 4096x3072 photo: 48 tiles of 512x512, no WPP inside a tile
cfg = dict(wpp=0, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
streams = streamgen.make_streams([(tw, th, 100 + i, 8, cfg) for i in range(rows * cols)])
heic = hu.build_heic([(s, tw, th) for s in streams], grid=(rows, cols, cols * tw, rows * th))
lh.load_hip_plugin()
which = os.environ.get("HIPDEC_TEST_LIBHEIF", "libheif.so")   # libheif_hipcolor.so: the build with the integration patch (grid fast path)
lh.decode(heic, lh.COLORSPACE_YCBCR, lh.CHROMA_420, max_threads=4)     # warm-up
lh.decode(heic, lh.COLORSPACE_RGB, lh.CHROMA_RGB, max_threads=4)
from libheif_amd.decoder import coalesce_stats
for out_name, cs, chroma in (("YCbCr planes", lh.COLORSPACE_YCBCR, lh.CHROMA_420), ("RGB24", lh.COLORSPACE_RGB, lh.CHROMA_RGB)):
    for threads in (1, 4, 16, 48):
        c0 = coalesce_stats()
        t0 = time.perf_counter()
        for _ in range(3):
            lh.decode(heic, cs, chroma, max_threads=threads)
        dt = (time.perf_counter() - t0) / 3
        c1 = coalesce_stats()
        print("%s, %s, max_decoding_threads %2d: %.1f ms per %dx%d grid photo (%.1f Mpixel/s); %d tile decodes in %d launch sets" %
              (which, out_name, threads, dt * 1e3, cols * tw, rows * th, cols * tw * rows * th / dt / 1e6, c1[0] - c0[0], c1[1] - c0[1]), flush=True)
