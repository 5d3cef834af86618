"""'irot' / 'imir' / 'clap' on the device against the reference itself: the REAL libheif decodes a HEIC that carries the
property through the HIP decoder plugin and applies the transformation on the host (ImageItem::decode_image, image_item.cc:949-1081 ->
HeifPixelImage::rotate_ccw / mirror_inplace / crop); hipdec_image_transform applies it to the very same decoded planes on the GPU.  Both
results must be identical, plane by plane."""
import ctypes as C
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle import pyoracle as orc
import heic_util
import libheif_host as lh

pytestmark = pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")

XF_ROTATE, XF_MIRROR, XF_CROP = 0, 1, 2


class ColorImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("chroma", C.c_int), ("bit_depth", C.c_int),
                ("plane", C.c_void_p * 4), ("stride", C.c_size_t * 4), ("on_device", C.c_int)]


def _heic(w, h, transforms, bit_depth=8, chroma=1, seed=5):
    planes = orc.synth_image(w, h, bit_depth, chroma, seed=seed)
    stream = orc.encode(planes, bit_depth=bit_depth)
    return heic_util.build_heic([(stream, w, h, chroma)], bit_depth=bit_depth, chroma_format_idc=chroma, transforms=transforms)


def test_transform_properties_are_recognised_by_the_reference():
    """(CPU) the synthetic files carry the properties the way libheif parses them: the handle reports the transformed size"""
    assert lh.primary_size(_heic(200, 136, [("irot", 1)])) == (136, 200)
    assert lh.primary_size(_heic(200, 136, [("irot", 2)])) == (200, 136)
    assert lh.primary_size(_heic(200, 136, [("imir", 1)])) == (200, 136)
    assert lh.primary_size(_heic(200, 136, [("clap", (180, 120, 10, 8))])) == (180, 120)


def _device_transform(planes, w, h, bit_depth, chroma, op, args):
    from libheif_amd import _capi
    L = _capi.load_library()
    L.hipdec_image_transform.argtypes = [C.POINTER(ColorImage), C.c_int, C.POINTER(C.c_int), C.POINTER(ColorImage)]
    src = ColorImage(w, h, chroma, bit_depth)
    keep = [np.ascontiguousarray(p) for p in planes]
    for c, p in enumerate(keep):
        src.plane[c] = p.ctypes.data
        src.stride[c] = p.strides[0]
    if op == XF_ROTATE and args[0] != 180:
        ow, oh = h, w
    elif op == XF_CROP:
        ow, oh = args[1] - args[0] + 1, args[3] - args[2] + 1
    else:
        ow, oh = w, h
    sx, sy = (2, 2) if chroma == 1 else (1, 1)
    outs = []
    dst = ColorImage()
    for c in range(len(keep)):
        if c == 0:
            pw, ph = ow, oh
        elif op == XF_CROP:
            pw = (args[1]) // sx - args[0] // sx + 1
            ph = (args[3]) // sy - args[2] // sy + 1
        else:
            pw, ph = (ow + sx - 1) // sx, (oh + sy - 1) // sy
        o = np.full((ph, pw + 3), 0xEE, keep[c].dtype)
        outs.append((o, pw))
        dst.plane[c] = o.ctypes.data
        dst.stride[c] = o.strides[0]
    a = (C.c_int * 4)(*(list(args) + [0] * (4 - len(args))))
    rc = L.hipdec_image_transform(C.byref(src), op, a, C.byref(dst))
    return rc, dst, [o[:, :pw] for o, pw in outs]


CASES = [
    ("irot", 1, XF_ROTATE, [90]), ("irot", 2, XF_ROTATE, [180]), ("irot", 3, XF_ROTATE, [270]),
    ("imir", 0, XF_MIRROR, [0]), ("imir", 1, XF_MIRROR, [1]),
    # (libheif tightens the plugin's max_image_size_pixels to (clap width + 64) x (clap height + 64), image_item.cc:1268-1275: a clean aperture
    #  much smaller than the coded picture is refused before any decoder runs — the windows here stay within that margin)
    ("clap", (180, 120, 10, 8), XF_CROP, [10, 189, 8, 127]), ("clap", (196, 130, 4, 6), XF_CROP, [4, 199, 6, 135]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth", [8, 10])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s" % (c[0], c[1]))
def test_device_transform_equals_what_libheif_does_to_the_same_planes(case, bit_depth):
    kind, arg, op, args = case
    w, h = 200, 136
    lh.load_hip_plugin()
    data = _heic(w, h, [(kind, arg)], bit_depth=bit_depth)
    want = lh.decode(data)                                   # the reference applies the property on the host
    raw = lh.decode(data, ignore_transformations=True)       # the decoded planes before it
    assert raw["planes"][0].shape == (h, w)
    rc, dst, got = _device_transform(raw["planes"], w, h, bit_depth, 1, op, args)
    assert rc == 0
    assert (dst.width, dst.height) == (want["planes"][0].shape[1], want["planes"][0].shape[0])
    for c in range(3):
        np.testing.assert_array_equal(got[c], want["planes"][c], err_msg="plane %d" % c)


@pytest.mark.gpu
def test_monochrome_and_444_planes_and_the_cases_the_reference_converts_first():
    from libheif_amd import _capi
    _capi.load_library()
    rng = np.random.default_rng(3)
    y = rng.integers(0, 256, (75, 41)).astype(np.uint8)          # odd sizes are fine without subsampled chroma
    rc, dst, got = _device_transform([y], 41, 75, 8, 0, XF_ROTATE, [90])
    assert rc == 0 and (dst.width, dst.height) == (75, 41)
    np.testing.assert_array_equal(got[0], np.rot90(y, 1))
    p444 = [rng.integers(0, 1024, (75, 41)).astype(np.uint16) for _ in range(3)]
    rc, dst, got = _device_transform(p444, 41, 75, 10, 3, XF_MIRROR, [1])
    assert rc == 0
    for c in range(3):
        np.testing.assert_array_equal(got[c], p444[c][:, ::-1])
    # 4:2:0 with an odd width: libheif converts to 4:4:4 before a 90 degree turn (pixelimage.cc:1195-1204) -> loud UNSUPPORTED, not a wrong image
    p420 = [rng.integers(0, 256, (76, 41)).astype(np.uint8), rng.integers(0, 256, (38, 21)).astype(np.uint8), rng.integers(0, 256, (38, 21)).astype(np.uint8)]
    rc, _, _ = _device_transform(p420, 41, 76, 8, 1, XF_ROTATE, [90])
    assert rc == -4
    rc, _, _ = _device_transform(p420, 41, 76, 8, 1, XF_CROP, [1, 20, 0, 9])      # odd left offset
    assert rc == -4
    rc, _, _ = _device_transform(p420, 41, 76, 8, 1, XF_CROP, [0, 20, 0, 9])      # odd window width: the half-covered chroma column stays the host's
    assert rc == -4
    rc, _, _ = _device_transform(p420, 41, 76, 8, 1, XF_CROP, [0, 41, 0, 9])      # right edge outside
    assert rc == -1
