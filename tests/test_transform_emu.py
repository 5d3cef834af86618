"""The plane kernels of the transformative item properties (libheif_amd/csrc/transform.hip: 'irot', 'imir', 'clap') under the SIMT
emulation, against the index arithmetic of the reference's ComponentStorage::rotate_ccw / mirror_inplace and HeifPixelImage::crop
(libheif/image/pixelimage.cc:1305-1355, :1433-1530) restated with numpy.  The GPU twin, against the compiled reference itself through
heif_decode_image, is tests/test_transform_gpu.py."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def emu():
    global _LIB
    if _LIB is None:
        from test_parse_emu import build_emu
        build_emu()
        L = C.CDLL(os.path.join(HERE, "emu", "libparse_emu.so"))
        L.hipdec_plane_rotate_ccw.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.hipdec_plane_mirror.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.hipdec_plane_crop.argtypes = [C.c_void_p, C.c_size_t] + [C.c_int] * 7 + [C.c_void_p, C.c_size_t, C.c_void_p]
        _LIB = L
    return _LIB


def reference_rotate(a, angle):
    """ComponentStorage::rotate_ccw<T>: 270: out[y][x] = in[h-1-x][y]; 180: out[y][x] = in[h-1-y][w-1-x]; 90: out[y][x] = in[x][w-1-y]"""
    h, w = a.shape
    if angle == 180:
        return a[::-1, ::-1].copy()
    out = np.zeros((w, h), a.dtype)
    for y in range(w):
        for x in range(h):
            out[y, x] = a[h - 1 - x, y] if angle == 270 else a[x, w - 1 - y]
    return out


def _plane(w, h, dtype, seed):
    rng = np.random.default_rng(seed)
    pad = 5   # a stride wider than the row
    buf = rng.integers(0, 256 if dtype == np.uint8 else 1024, (h, w + pad)).astype(dtype)
    return buf


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("size", [(64, 64), (1, 1), (130, 67), (3, 200), (257, 5), (96, 128)])
def test_rotate_matches_the_reference_index_arithmetic(dtype, size):
    L = emu()
    w, h = size
    src = _plane(w, h, dtype, 7)
    es = src.itemsize
    assert np.array_equal(reference_rotate(src[:, :w], 90), np.rot90(src[:, :w], 1))   # the restatement is a plain counter-clockwise turn
    for angle in (90, 180, 270):
        ow, oh = (w, h) if angle == 180 else (h, w)
        dst = np.full((oh, ow + 3), 0xEE, dtype)
        rc = L.hipdec_plane_rotate_ccw(src.ctypes.data, src.strides[0], w, h, es, angle, dst.ctypes.data, dst.strides[0], None)
        assert rc == 0
        np.testing.assert_array_equal(dst[:, :ow], reference_rotate(src[:, :w], angle), err_msg="angle %d" % angle)
        assert (dst[:, ow:] == 0xEE).all()   # nothing behind the rows


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("size", [(64, 64), (1, 1), (130, 67), (3, 200), (1027, 9)])
def test_mirror_and_crop_match_the_reference_index_arithmetic(dtype, size):
    L = emu()
    w, h = size
    src = _plane(w, h, dtype, 11)
    es = src.itemsize
    for direction, want in ((0, src[::-1, :w]), (1, src[:, :w][:, ::-1])):   # heif_transform_mirror_direction: 0 vertical (rows), 1 horizontal
        dst = np.full((h, w + 2), 0xEE, dtype)
        assert L.hipdec_plane_mirror(src.ctypes.data, src.strides[0], w, h, es, direction, dst.ctypes.data, dst.strides[0], None) == 0
        np.testing.assert_array_equal(dst[:, :w], want)
        assert (dst[:, w:] == 0xEE).all()
    rng = np.random.default_rng(w * 31 + h)
    for _ in range(6):
        left, top = int(rng.integers(0, w)), int(rng.integers(0, h))
        ow, oh = int(rng.integers(1, w - left + 1)), int(rng.integers(1, h - top + 1))
        dst = np.full((oh, ow + 4), 0xEE, dtype)
        assert L.hipdec_plane_crop(src.ctypes.data, src.strides[0], w, h, es, left, top, ow, oh, dst.ctypes.data, dst.strides[0], None) == 0
        np.testing.assert_array_equal(dst[:, :ow], src[top:top + oh, left:left + ow])
        assert (dst[:, ow:] == 0xEE).all()


def test_bad_arguments_are_refused():
    L = emu()
    a = np.zeros((8, 8), np.uint8)
    d = np.zeros((8, 8), np.uint8)
    assert L.hipdec_plane_rotate_ccw(a.ctypes.data, 8, 8, 8, 1, 45, d.ctypes.data, 8, None) != 0
    assert L.hipdec_plane_rotate_ccw(a.ctypes.data, 8, 8, 8, 3, 90, d.ctypes.data, 8, None) != 0
    assert L.hipdec_plane_mirror(a.ctypes.data, 8, 8, 8, 1, 2, d.ctypes.data, 8, None) != 0
    assert L.hipdec_plane_crop(a.ctypes.data, 8, 8, 8, 1, 4, 4, 5, 4, d.ctypes.data, 8, None) != 0    # reaches past the right edge
    assert L.hipdec_plane_crop(a.ctypes.data, 8, 8, 8, 1, 0, 0, 8, 8, d.ctypes.data, 7, None) != 0    # destination stride too small
