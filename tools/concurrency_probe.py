"""Does a small kernel get on the chip while the CABAC work pool of a big batch is resident?  (GPU box, dev tool.)
A batch of 512 4K stills is started asynchronously; while it runs, hipdec_color_convert (its own pooled stream: upload of one 4K image,
one conversion kernel, copy back) is called in a loop and timed.  Run with HIPDEC_POOL_WAVES=8192 (every wave slot taken) and with
7168 / 6144 (one / two slots per SIMD left free)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tools import streamgen
import libheif_amd
from libheif_amd.decoder import Batch

cfg = dict(wpp=1, qp=27, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
streams = streamgen.make_streams([(3840, 2160, 7000 + i, 8, cfg) for i in range(16)])
n = int(os.environ.get("PROBE_STILLS", "512"))
big = Batch([streams[i % 16] for i in range(n)])
big.run(); big.status()                      # warm-up (arena, kernels)

L = libheif_amd.load_library()
class ColorImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("chroma", C.c_int), ("bit_depth", C.c_int), ("plane", C.c_void_p * 4), ("stride", C.c_size_t * 4), ("on_device", C.c_int)]
class Nclx(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("has_nclx", "colour_primaries", "transfer_characteristics", "matrix_coefficients", "full_range_flag")]
L.hipdec_color_convert.argtypes = [C.POINTER(ColorImage), C.POINTER(Nclx), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
w, h = 3840, 2160
rng = np.random.default_rng(1)
planes = [rng.integers(0, 255, (h, w), np.uint8), rng.integers(0, 255, (h // 2, w // 2), np.uint8), rng.integers(0, 255, (h // 2, w // 2), np.uint8)]
img = ColorImage(w, h, 1, 8)
for c, p in enumerate(planes):
    img.plane[c] = p.ctypes.data; img.stride[c] = p.strides[0]
nclx = Nclx(1, 1, 13, 6, 1)
out = np.zeros((h, w * 3), np.uint8)

def convert():
    t0 = time.perf_counter()
    rc = L.hipdec_color_convert(C.byref(img), C.byref(nclx), 10, 1, 0, out.ctypes.data, w * 3, 0)
    assert rc == 0
    return (time.perf_counter() - t0) * 1e3

idle = [convert() for _ in range(5)]
t0 = time.perf_counter()
big.run()                                     # asynchronous
busy = []
while time.perf_counter() - t0 < 0.25:        # the parse of 512 stills takes ~0.2 s
    busy.append(convert())
big.status()
total = (time.perf_counter() - t0) * 1e3
print("pool waves %s: conversion alone %.1f ms (min of 5); during the batch: %s ms; batch + probes took %.0f ms" %
      (os.environ.get("HIPDEC_POOL_WAVES", "default"), min(idle), ", ".join("%.1f" % b for b in busy[:12]), total))
