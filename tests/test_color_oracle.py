"""The colour-stage oracle (oracle/color_oracle.c) against (1) the reference's own known-answer test
and (2) the real reference ops compiled from /root/reference (oracle/_ref, prebuilt)."""
import numpy as np
import pytest

import ref_harness as ref
from oracle import pyoracle as orc

needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (make -C oracle ref)")


def _planes(rng, w, h, bpp, chroma=1):
    hi = 1 << bpp
    y = rng.integers(0, hi, (h, w))
    cw, chh = ((w + 1) // 2, (h + 1) // 2) if chroma == 1 else ((w + 1) // 2, h) if chroma == 2 else (w, h)
    return y, rng.integers(0, hi, (chh, cw)), rng.integers(0, hi, (chh, cw))


def test_bilinear_kat_from_reference_tests():
    """libheif tests/conversion.cc:685-725 ("Bilinear upsampling"): 2x2 chroma -> exact 4x4."""
    cb = np.array([[10, 40], [100, 240]])
    cr = np.array([[255, 200], [50, 0]])
    exp_cb = np.array([[10, 18, 33, 40], [33, 47, 76, 90], [78, 106, 162, 190], [100, 135, 205, 240]])
    exp_cr = np.array([[255, 241, 214, 200], [204, 190, 163, 150], [101, 88, 63, 50], [50, 38, 13, 0]])
    np.testing.assert_array_equal(orc.color_bilinear_420_to_444(cb, 4, 4), exp_cb)
    np.testing.assert_array_equal(orc.color_bilinear_420_to_444(cr, 4, 4), exp_cr)


@needs_ref
@pytest.mark.parametrize("w,h", [(64, 48), (66, 50), (130, 34)])
@pytest.mark.parametrize("matrix,primaries", [(1, 1), (6, 1), (9, 9), (2, 2), (5, 5), (12, 1)])
def test_a9_int_420_to_rgb24_matches_reference(w, h, matrix, primaries):
    rng = np.random.default_rng(w * 1000 + matrix)
    y, cb, cr = _planes(rng, w, h, 8)
    nclx = (primaries, 13, matrix, 1)
    got = orc.color_420_to_rgb24(y, cb, cr, nclx)
    exp = ref.convert([y, cb, cr], 8, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGB, upsampling=ref.UPS_NN)[0]
    np.testing.assert_array_equal(got.reshape(h, -1), exp)


@needs_ref
def test_a9_rgb32_alpha_fill():
    rng = np.random.default_rng(5)
    y, cb, cr = _planes(rng, 64, 32, 8)
    nclx = (1, 13, 6, 1)
    got = orc.color_420_to_rgb24(y, cb, cr, nclx, alpha=True)
    exp = ref.convert([y, cb, cr], 8, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGBA, upsampling=ref.UPS_NN)[0]
    np.testing.assert_array_equal(got.reshape(32, -1), exp)


@needs_ref
@pytest.mark.parametrize("matrix,primaries,full", [(6, 1, 0), (1, 1, 0), (2, 2, 0), (9, 9, 0), (0, 1, 0), (0, 1, 1), (8, 1, 1), (12, 9, 0)])
def test_a10_a11_float_limited_range_to_rgb24_matches_reference(matrix, primaries, full):
    """default pipeline for limited-range 8-bit input: Op_YCbCr_to_RGB<u8> -> Op_RGB_to_RGB24_32."""
    rng = np.random.default_rng(matrix * 7 + full)
    w, h = 96, 40
    y, cb, cr = _planes(rng, w, h, 8)
    nclx = (primaries, 13, matrix, full)
    r, g, b = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, 1, nclx)
    got = orc.color_rgb_planar_to_interleaved8(r, g, b)
    exp = ref.convert([y, cb, cr], 8, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGB, upsampling=ref.UPS_NN)[0]
    np.testing.assert_array_equal(got.reshape(h, -1), exp)


@needs_ref
@pytest.mark.parametrize("bpp", [10, 12])
@pytest.mark.parametrize("matrix,primaries,full", [(9, 9, 0), (9, 9, 1), (1, 1, 0), (6, 6, 0)])
@pytest.mark.parametrize("le", [True, False])
def test_a12_420_to_rrggbb_matches_reference(bpp, matrix, primaries, full, le):
    rng = np.random.default_rng(bpp + matrix)
    w, h = 80, 36
    y, cb, cr = _planes(rng, w, h, bpp)
    nclx = (primaries, 16, matrix, full)
    got = orc.color_420_to_rrggbb(y, cb, cr, bpp, nclx, little_endian=le)
    exp = ref.convert([y, cb, cr], bpp, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RRGGBB_LE if le else ref.CH_RRGGBB_BE,
                      upsampling=ref.UPS_NN)[0]
    np.testing.assert_array_equal(got, exp)


@needs_ref
@pytest.mark.parametrize("w,h", [(64, 48), (65, 49), (34, 130), (18, 18)])
@pytest.mark.parametrize("bpp", [8, 10])
def test_a13_bilinear_matches_reference(w, h, bpp):
    rng = np.random.default_rng(w + h + bpp)
    y, cb, cr = _planes(rng, w, h, bpp)
    exp = ref.convert([y, cb, cr], bpp, ref.CH_420, (1, 13, 6, 1), ref.CS_YCBCR, ref.CH_444,
                      upsampling=ref.UPS_BILINEAR, only_preferred=True)
    np.testing.assert_array_equal(exp[0], y)
    # the reference leaves the right/bottom border of odd-sized planes at HeifPixelImage's zero fill
    np.testing.assert_array_equal(orc.color_bilinear_420_to_444(cb, w, h), exp[1])
    np.testing.assert_array_equal(orc.color_bilinear_420_to_444(cr, w, h), exp[2])


@needs_ref
@pytest.mark.parametrize("w,h", [(64, 48), (65, 49), (34, 13), (2, 3), (1, 2), (3, 1)])
@pytest.mark.parametrize("bpp", [8, 10])
def test_f4_bilinear_422_matches_reference(w, h, bpp):
    """Op_YCbCr422_bilinear_to_YCbCr444 (chroma_sampling.cc:732-954): the compiled reference op against the restatement"""
    rng = np.random.default_rng(w * 7 + h + bpp)
    hi = 1 << bpp
    dt = np.uint8 if bpp <= 8 else np.uint16
    y = rng.integers(0, hi, (h, w)).astype(dt)
    cb = rng.integers(0, hi, (h, (w + 1) // 2)).astype(dt)
    cr = rng.integers(0, hi, (h, (w + 1) // 2)).astype(dt)
    exp = ref.convert([y, cb, cr], bpp, ref.CH_422, (1, 13, 6, 1), ref.CS_YCBCR, ref.CH_444,
                      upsampling=ref.UPS_BILINEAR, only_preferred=True)
    np.testing.assert_array_equal(exp[0], y)
    np.testing.assert_array_equal(orc.color_bilinear_422_to_444(cb, w, h), exp[1])
    np.testing.assert_array_equal(orc.color_bilinear_422_to_444(cr, w, h), exp[2])


@needs_ref
def test_a14_to_sdr_then_default_pipeline():
    """10-bit + convert_hdr_to_8bit: Op_to_sdr_planes first, then the 8-bit row."""
    rng = np.random.default_rng(3)
    w, h = 64, 32
    y, cb, cr = _planes(rng, w, h, 10)
    nclx = (9, 16, 9, 1)
    y8, cb8, cr8 = (orc.color_to_sdr(p, 10) for p in (y, cb, cr))
    got = orc.color_420_to_rgb24(y8, cb8, cr8, nclx)
    exp = ref.convert([y, cb, cr], 10, ref.CH_420, nclx, ref.CS_RGB, ref.CH_RGB, out_bpp=8, upsampling=ref.UPS_NN)[0]
    np.testing.assert_array_equal(got.reshape(h, -1), exp)


@needs_ref
def test_unsupported_matrices_fail_like_reference():
    """libheif tests/conversion.cc:548-587: matrix 11 / 14 must not build a pipeline."""
    rng = np.random.default_rng(1)
    y, cb, cr = _planes(rng, 32, 16, 8)
    for m in (11, 14):
        with pytest.raises(RuntimeError):
            ref.convert([y, cb, cr], 8, ref.CH_420, (1, 13, m, 1), ref.CS_RGB, ref.CH_RGB)
