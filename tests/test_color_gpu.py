"""GPU parity of the fused HIP colour stage (through the C ABI) against the colour oracle, and —
where oracle/_ref was built — against the real reference ops."""
import numpy as np
import pytest

from oracle import pyoracle as orc
import ref_harness as ref

pytestmark = pytest.mark.gpu


def _planes(rng, w, h, bpp, chroma=1):
    hi = 1 << bpp
    cw, chh = ((w + 1) // 2, (h + 1) // 2) if chroma == 1 else ((w + 1) // 2, h) if chroma == 2 else (w, h)
    return rng.integers(0, hi, (h, w)), rng.integers(0, hi, (chh, cw)), rng.integers(0, hi, (chh, cw))


@pytest.mark.parametrize("w,h", [(64, 48), (66, 50), (130, 34), (1920, 1080), (3840, 2160), (5, 3), (2, 2)])
@pytest.mark.parametrize("matrix,primaries", [(1, 1), (6, 1), (9, 9), (2, 2), (12, 1)])
def test_a9_int_rgb24(w, h, matrix, primaries):
    from libheif_amd import color
    rng = np.random.default_rng(w + matrix)
    y, cb, cr = _planes(rng, w, h, 8)
    nclx = (primaries, 13, matrix, 1)
    got = color.convert_colorspace([y, cb, cr], 8, 1, nclx, color.CHROMA_RGB, upsampling=color.UPSAMPLING_NEAREST)
    exp = orc.color_420_to_rgb24(y, cb, cr, nclx).reshape(h, -1)
    np.testing.assert_array_equal(got, exp)
    if ref.available() and w * h < 300000:
        np.testing.assert_array_equal(got, ref.convert([y, cb, cr], 8, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGB, upsampling=ref.UPS_NN)[0])


def test_a9_rgb32():
    from libheif_amd import color
    rng = np.random.default_rng(9)
    y, cb, cr = _planes(rng, 258, 66, 8)
    nclx = (1, 13, 6, 1)
    got = color.convert_colorspace([y, cb, cr], 8, 1, nclx, color.CHROMA_RGBA, upsampling=color.UPSAMPLING_NEAREST)
    np.testing.assert_array_equal(got, orc.color_420_to_rgb24(y, cb, cr, nclx, alpha=True).reshape(66, -1))


@pytest.mark.parametrize("matrix,primaries,full", [(6, 1, 0), (1, 1, 0), (2, 2, 0), (9, 9, 0), (0, 1, 0), (0, 1, 1), (8, 1, 1), (12, 9, 0)])
@pytest.mark.parametrize("w,h", [(96, 40), (1281, 855)])
def test_a10_a11_float_rgb24_bit_exact(matrix, primaries, full, w, h):
    """float path: tolerance stated by north_star is for the nclx transform only; we hold 0 LSB
    (device code is built with -ffp-contract=off)."""
    from libheif_amd import color
    rng = np.random.default_rng(matrix * 3 + full + w)
    y, cb, cr = _planes(rng, w, h, 8)
    nclx = (primaries, 13, matrix, full)
    got = color.convert_colorspace([y, cb, cr], 8, 1, nclx, color.CHROMA_RGB, upsampling=color.UPSAMPLING_NEAREST)
    r, g, b = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, 1, nclx)
    exp = orc.color_rgb_planar_to_interleaved8(r, g, b).reshape(h, -1)
    np.testing.assert_array_equal(got, exp)
    if ref.available() and w * h < 300000:
        np.testing.assert_array_equal(got, ref.convert([y, cb, cr], 8, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGB, upsampling=ref.UPS_NN)[0])


@pytest.mark.parametrize("bpp", [10, 12])
@pytest.mark.parametrize("le", [True, False])
@pytest.mark.parametrize("matrix,primaries,full", [(9, 9, 0), (9, 9, 1), (1, 1, 0)])
def test_a12_rrggbb(bpp, le, matrix, primaries, full):
    from libheif_amd import color
    rng = np.random.default_rng(bpp + matrix + full)
    w, h = 322, 70
    y, cb, cr = _planes(rng, w, h, bpp)
    nclx = (primaries, 16, matrix, full)
    got = color.convert_colorspace([y, cb, cr], bpp, 1, nclx, color.CHROMA_RRGGBB_LE if le else color.CHROMA_RRGGBB_BE,
                                   upsampling=color.UPSAMPLING_NEAREST)
    np.testing.assert_array_equal(got, orc.color_420_to_rrggbb(y, cb, cr, bpp, nclx, little_endian=le))


@pytest.mark.parametrize("w,h", [(64, 48), (65, 49), (34, 130), (4, 4), (1280, 854)])
@pytest.mark.parametrize("bpp", [8, 10])
def test_a13_bilinear(w, h, bpp):
    from libheif_amd import color
    rng = np.random.default_rng(w + h + bpp)
    y, cb, cr = _planes(rng, w, h, bpp)
    got = color.convert_colorspace([y, cb, cr], bpp, 1, (1, 13, 6, 1), color.CHROMA_444,
                                   upsampling=color.UPSAMPLING_BILINEAR, only_preferred=True)
    np.testing.assert_array_equal(got[0], y)
    np.testing.assert_array_equal(got[1], orc.color_bilinear_420_to_444(cb, w, h))
    np.testing.assert_array_equal(got[2], orc.color_bilinear_420_to_444(cr, w, h))


def test_bilinear_then_float_chain_matches_reference_planner():
    """`heif-dec -C bilinear` chain: 420_bilinear_to_444 -> Op_YCbCr_to_RGB<u8> -> Op_RGB_to_RGB24_32."""
    from libheif_amd import color
    rng = np.random.default_rng(77)
    w, h = 130, 66
    y, cb, cr = _planes(rng, w, h, 8)
    nclx = (1, 13, 6, 1)
    got = color.convert_colorspace([y, cb, cr], 8, 1, nclx, color.CHROMA_RGB, upsampling=color.UPSAMPLING_BILINEAR, only_preferred=True)
    cb4, cr4 = orc.color_bilinear_420_to_444(cb, w, h), orc.color_bilinear_420_to_444(cr, w, h)
    r, g, b = orc.color_ycbcr_to_rgb_planar(y, cb4, cr4, 8, 3, nclx)
    np.testing.assert_array_equal(got, orc.color_rgb_planar_to_interleaved8(r, g, b).reshape(h, -1))
    if ref.available():
        exp = ref.convert([y, cb, cr], 8, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGB, upsampling=ref.UPS_BILINEAR, only_preferred=True)[0]
        np.testing.assert_array_equal(got, exp)


def test_a14_hdr_to_8bit_chain():
    from libheif_amd import color
    rng = np.random.default_rng(3)
    w, h = 194, 50
    y, cb, cr = _planes(rng, w, h, 10)
    nclx = (9, 16, 9, 1)
    got = color.convert_colorspace([y, cb, cr], 10, 1, nclx, color.CHROMA_RGB, upsampling=color.UPSAMPLING_NEAREST)
    y8, cb8, cr8 = (orc.color_to_sdr(p, 10) for p in (y, cb, cr))
    np.testing.assert_array_equal(got, orc.color_420_to_rgb24(y8, cb8, cr8, nclx).reshape(h, -1))


def test_unsupported_matrix_fails_loudly():
    from libheif_amd import color, HipDecError
    rng = np.random.default_rng(1)
    y, cb, cr = _planes(rng, 32, 16, 8)
    with pytest.raises(HipDecError):
        color.convert_colorspace([y, cb, cr], 8, 1, (1, 13, 11, 1), color.CHROMA_RGB)


def test_f4_to_hdr_matches_the_compiled_reference_and_swap_pq():
    """Op_to_hdr_planes against the reference's own pipeline (YCbCr 8 bit -> YCbCr 10 / 12 bit planes through oracle/_ref), the endianness swap
    against numpy, the PQ EOTF against the published formula (1e-6 relative)"""
    import ctypes as C
    import libheif_amd
    from libheif_amd._capi import DeviceBuffer, check
    from test_color_emu import _pq_reference
    lib = libheif_amd.load_library()
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    lib.hipdec_color_to_hdr.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    lib.hipdec_color_swap_endianness.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    lib.hipdec_color_pq_to_linear.argtypes = [vp, sz, ci, ci, ci, ci, ci, vp, sz, vp]
    rng = np.random.default_rng(5)
    w, h = 322, 70
    planes = [np.ascontiguousarray(rng.integers(0, 256, s).astype(np.uint8)) for s in ((h, w), (h // 2, w // 2), (h // 2, w // 2))]
    for bits in (10, 12):
        got = []
        for p in planes:
            src = DeviceBuffer.from_numpy(p); dst = DeviceBuffer(p.size * 2)
            check(lib.hipdec_color_to_hdr(src.ptr, p.shape[1], p.shape[1], p.shape[0], bits, dst.ptr, p.shape[1] * 2, None))
            check(lib.hipdec_stream_synchronize(None))
            got.append(dst.to_numpy(p.shape, np.uint16))
        if ref.available():
            want = ref.convert(planes, 8, ref.CH_420, (1, 13, 6, 1), ref.CS_YCBCR, ref.CH_420, out_bpp=bits)
            for g, e in zip(got, want):
                np.testing.assert_array_equal(g, e[:, :g.shape[1]])
        np.testing.assert_array_equal(got[0], (planes[0].astype(np.uint32) << (bits - 8)) | (planes[0].astype(np.uint32) >> (16 - bits)))
    px = np.ascontiguousarray(rng.integers(0, 1 << 16, (h, w * 3)).astype(np.uint16))
    src = DeviceBuffer.from_numpy(px); dst = DeviceBuffer(px.nbytes)
    check(lib.hipdec_color_swap_endianness(src.ptr, w * 6, w, h, 3, dst.ptr, w * 6, None))
    check(lib.hipdec_stream_synchronize(None))
    np.testing.assert_array_equal(dst.to_numpy((h, w * 3), np.uint16), px.byteswap())
    code = np.ascontiguousarray(rng.integers(0, 1 << 10, (h, w * 3)).astype(np.uint16))
    src = DeviceBuffer.from_numpy(code); dst = DeviceBuffer(code.size * 4)
    check(lib.hipdec_color_pq_to_linear(src.ptr, w * 6, w, h, 3, 10, 0, dst.ptr, w * 12, None))
    check(lib.hipdec_stream_synchronize(None))
    np.testing.assert_allclose(dst.to_numpy((h, w * 3), np.float32), _pq_reference(code, 10), rtol=1e-6, atol=1e-9)


def test_f4_hlg_inverse_oetf_matches_the_published_formula():
    """hybrid log-gamma code values -> scene linear light (ARIB STD-B67 / BT.2100; SURVEY 8 f4 asks for a PQ / HLG stage, the reference has neither):
    the table form (10 / 12 bit) and the per-sample form (16 bit) against the formula in fp64, 1e-6 relative"""
    import ctypes as C
    import libheif_amd
    from libheif_amd._capi import DeviceBuffer, check
    from test_color_emu import _hlg_reference
    lib = libheif_amd.load_library()
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    lib.hipdec_color_hlg_to_linear.argtypes = [vp, sz, ci, ci, ci, ci, ci, vp, sz, vp]
    rng = np.random.default_rng(8)
    w, h = 322, 70
    for bits in (10, 12, 16):
        code = np.ascontiguousarray(rng.integers(0, 1 << bits, (h, w * 3)).astype(np.uint16))
        code[0, :3] = (0, (1 << bits) - 1, ((1 << bits) - 1) // 2)
        src = DeviceBuffer.from_numpy(code); dst = DeviceBuffer(code.size * 4)
        check(lib.hipdec_color_hlg_to_linear(src.ptr, w * 6, w, h, 3, bits, 0, dst.ptr, w * 12, None))
        check(lib.hipdec_stream_synchronize(None))
        np.testing.assert_allclose(dst.to_numpy((h, w * 3), np.float32), _hlg_reference(code, bits), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("w,h", [(64, 48), (65, 49), (34, 13), (2, 3), (1, 2), (1281, 33)])
@pytest.mark.parametrize("bpp", [8, 10])
def test_f4_bilinear_422_to_444(w, h, bpp):
    """Op_YCbCr422_bilinear_to_YCbCr444 (chroma_sampling.cc:732-954) on the device against the restatement (pinned to the compiled reference op
    in tests/test_color_oracle.py) and, when oracle/_ref is present, against the compiled reference itself"""
    import ctypes as C
    import libheif_amd
    from libheif_amd._capi import DeviceBuffer, check
    lib = libheif_amd.load_library()
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    lib.hipdec_color_bilinear_422_to_444.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    rng = np.random.default_rng(w * 7 + h + bpp)
    dt = np.uint8 if bpp <= 8 else np.uint16
    y = rng.integers(0, 1 << bpp, (h, w)).astype(dt)
    chroma = [np.ascontiguousarray(rng.integers(0, 1 << bpp, (h, (w + 1) // 2)).astype(dt)) for _ in range(2)]
    got = []
    for p in chroma:
        src = DeviceBuffer.from_numpy(p); dst = DeviceBuffer(w * h * p.itemsize)
        check(lib.hipdec_color_bilinear_422_to_444(src.ptr, p.strides[0], w, h, bpp, dst.ptr, w * p.itemsize, None))
        check(lib.hipdec_stream_synchronize(None))
        got.append(dst.to_numpy((h, w), dt))
        np.testing.assert_array_equal(got[-1], orc.color_bilinear_422_to_444(p, w, h))
    if ref.available():
        want = ref.convert([y] + chroma, bpp, ref.CH_422, (1, 13, 6, 1), ref.CS_YCBCR, ref.CH_444, upsampling=ref.UPS_BILINEAR, only_preferred=True)
        np.testing.assert_array_equal(got[0], want[1])
        np.testing.assert_array_equal(got[1], want[2])


@pytest.mark.parametrize("bpp,nclx", [(8, (1, 13, 6, 1)), (8, (1, 13, 1, 0)), (10, (9, 16, 9, 0))])
def test_f4_422_input_through_the_c_planner_matches_the_compiled_reference(bpp, nclx):
    """4:2:2 planes -> interleaved RGB with bilinear upsampling forced: hipdec_color_convert (planner + Op_YCbCr422_bilinear_to_YCbCr444 + the
    float op + interleave, all on the device) against the compiled reference pipeline on the same planes"""
    import ctypes as C
    import libheif_amd
    from test_transform_gpu import ColorImage
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    lib = libheif_amd.load_library()

    class Nclx(C.Structure):
        _fields_ = [("has_nclx", C.c_int), ("colour_primaries", C.c_int), ("transfer_characteristics", C.c_int), ("matrix_coefficients", C.c_int), ("full_range_flag", C.c_int)]
    lib.hipdec_color_convert.argtypes = [C.POINTER(ColorImage), C.POINTER(Nclx), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    w, h = 130, 46
    rng = np.random.default_rng(bpp + nclx[2])
    dt = np.uint8 if bpp <= 8 else np.uint16
    planes = [np.ascontiguousarray(rng.integers(0, 1 << bpp, s).astype(dt)) for s in ((h, w), (h, w // 2), (h, w // 2))]
    img = ColorImage(w, h, 2, bpp)
    for c, p in enumerate(planes):
        img.plane[c] = p.ctypes.data
        img.stride[c] = p.strides[0]
    out = np.zeros((h, w * 3), np.uint8)
    n = Nclx(1, *nclx)
    rc = lib.hipdec_color_convert(C.byref(img), C.byref(n), 10, 2, 1, out.ctypes.data, out.strides[0], 0)
    assert rc == 0
    want = ref.convert(planes, bpp, ref.CH_422, nclx, ref.CS_RGB, ref.CH_RGB, out_bpp=8, upsampling=ref.UPS_BILINEAR, only_preferred=True)[0]
    np.testing.assert_array_equal(out, want)
