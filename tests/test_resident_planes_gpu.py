"""The colour boundary's device-resident planes (hipdec_decoder_read_plane_tracked -> hipdec_color_convert) are only re-used while
the host plane still holds exactly the bytes the decoder handed over: libheif edits decoded planes IN PLACE before the colour
conversion (mirror_inplace, /root/reference/libheif/image-items/image_item.cc:969 — same pointer, stride and size).  Round 2 keyed the
re-use on a sparse sample of the plane; an edit outside the sampled windows went unnoticed."""
import ctypes as C
import numpy as np
import pytest

import libheif_amd
from libheif_amd import color
from libheif_amd._capi import check
from libheif_amd.decoder import ImageInfo, _bind
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu

NCLX = (1, 13, 6, 1)


class ColorImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("chroma", C.c_int), ("bit_depth", C.c_int),
                ("plane", C.c_void_p * 4), ("stride", C.c_size_t * 4), ("on_device", C.c_int)]


def _lib():
    lib = _bind(libheif_amd.load_library())
    lib.hipdec_set_plane_tracking.restype = None
    lib.hipdec_set_plane_tracking.argtypes = [C.c_int]
    lib.hipdec_color_boundary_stats.restype = None
    lib.hipdec_color_boundary_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
    lib.hipdec_decoder_read_plane_tracked.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.hipdec_color_convert.argtypes = [C.POINTER(ColorImage), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    return lib


def _resident(lib):
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib.hipdec_color_boundary_stats(C.byref(a), C.byref(b), C.byref(c))
    return b.value


def _decode_tracked(lib, stream):
    h = C.c_void_p()
    check(lib.hipdec_decoder_new(C.byref(h), 0, 0))
    check(lib.hipdec_decoder_push_data(h, stream, len(stream)))
    info = ImageInfo()
    check(lib.hipdec_decoder_decode(h, C.byref(info)))
    planes = []
    for c in range(3):
        w, hh = (info.width, info.height) if c == 0 else (info.chroma_width, info.chroma_height)
        a = np.empty((hh, w), np.uint8)
        check(lib.hipdec_decoder_read_plane_tracked(h, c, a.ctypes.data, w))
        planes.append(a)
    return h, planes


def _convert(lib, planes):
    h, w = planes[0].shape
    img = ColorImage(w, h, 1, 8)
    for c in range(3):
        img.plane[c] = planes[c].ctypes.data
        img.stride[c] = planes[c].strides[0]
    img.on_device = 0
    ns = color._nclx_struct(NCLX)
    out = np.zeros((h, w * 3), np.uint8)
    check(lib.hipdec_color_convert(C.byref(img), C.byref(ns), color.CHROMA_RGB, color.UPSAMPLING_NEAREST, 0, out.ctypes.data, w * 3, 0))
    return out


@pytest.mark.parametrize("edit", ["none", "one_interior_pixel", "mirror_flat_chroma"])
def test_an_in_place_edit_between_decode_and_conversion_is_seen(edit):
    lib = _lib()
    lib.hipdec_set_plane_tracking(1)
    src = orc.synth_image(328, 200, 8, 1, seed=41)
    if edit == "mirror_flat_chroma":       # neutral chroma: the chroma planes are their own mirror image, only luma tells
        src[1][:] = 128; src[2][:] = 128
    stream = orc.encode(src, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1, qp=12)
    dec, planes = _decode_tracked(lib, stream)
    ref = orc.decode(stream)["planes"]
    for c in range(3):
        np.testing.assert_array_equal(planes[c], ref[c])
    before = _resident(lib)
    if edit == "one_interior_pixel":       # row 37, byte 101: in none of the 16 rows x 3 windows round 2 sampled
        planes[0][37, 101] ^= 0x80
    elif edit == "mirror_flat_chroma":     # what mirror_inplace does: same buffers, rows reversed
        for p in planes:
            p[:] = p[:, ::-1].copy()
    got = _convert(lib, planes)
    exp = orc.color_420_to_rgb24(planes[0], planes[1], planes[2], NCLX).reshape(got.shape)
    np.testing.assert_array_equal(got, exp)
    hits = _resident(lib) - before
    if edit == "none":
        assert hits == 3                   # all three planes were read from the decoder's device copy
    elif edit == "one_interior_pixel":
        assert hits == 2                   # the edited luma plane was uploaded
    lib.hipdec_decoder_free(dec)
    lib.hipdec_forget_resident_planes()


def test_entries_serve_one_conversion_and_tracking_can_be_switched_off():
    lib = _lib()
    lib.hipdec_set_plane_tracking(1)
    stream = orc.encode(orc.synth_image(128, 72, 8, 1, seed=42), vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    dec, planes = _decode_tracked(lib, stream)
    b0 = _resident(lib)
    first = _convert(lib, planes)
    assert _resident(lib) - b0 == 3
    again = _convert(lib, planes)          # the entries are gone: upload path, same pixels
    assert _resident(lib) - b0 == 3
    np.testing.assert_array_equal(first, again)
    lib.hipdec_decoder_free(dec)
    lib.hipdec_set_plane_tracking(0)
    dec, planes = _decode_tracked(lib, stream)
    lib.hipdec_set_plane_tracking(0)       # (a conversion switches tracking on again: the decode above ran without)
    b1 = _resident(lib)
    np.testing.assert_array_equal(_convert(lib, planes), first)
    assert _resident(lib) == b1
    lib.hipdec_decoder_free(dec)
    lib.hipdec_forget_resident_planes()


def test_registry_is_bounded_by_the_device_bytes_it_pins():
    """a host that never converts colour never consumes entries; every entry keeps a whole launch-set arena alive.  With a byte cap
    (HIPDEC_RESIDENT_MAX_BYTES; default an eighth of the device) the registry drops its oldest entries instead of pinning HBM without bound."""
    import os, subprocess, sys
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_resident_planes_gpu as t
from oracle import pyoracle as orc
lib = t._lib()
lib.hipdec_resident_plane_stats.restype = None
lib.hipdec_resident_plane_stats.argtypes = [C.POINTER(C.c_uint64)] * 2
lib.hipdec_set_plane_tracking(1)
s = orc.encode(orc.synth_image(256, 192, 8, 1, seed=2))
keep, peak = [], 0
for i in range(40):
    h, planes = t._decode_tracked(lib, s)       # never converted: nothing consumes the entries
    keep.append(planes)
    lib.hipdec_decoder_free(h)
    n, b = C.c_uint64(), C.c_uint64()
    lib.hipdec_resident_plane_stats(C.byref(n), C.byref(b))
    peak = max(peak, b.value)
print("PEAK", peak, n.value)
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    cap = 24 << 20                                  # six 4 MiB arenas
    env = dict(os.environ, HIPDEC_RESIDENT_MAX_BYTES=str(cap), HIPDEC_RESIDENT_TTL_MS="600000")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    peak, entries = [int(x) for x in out.stdout.split("PEAK")[1].split()]
    assert 0 < peak <= cap + (8 << 20), (peak, cap)   # (one arena above the cap at most, before the sweep behind the insert)
    assert entries < 40 * 3
