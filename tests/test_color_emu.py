"""CPU check of the colour kernels' logic: libheif_amd/csrc/color.hip compiled for the host against the SIMT emulator of
tests/emu/shim and called through its own C entry points (hipdec_color_*), against the colour oracle — which is itself
pinned to the reference's compiled ops (tests/test_color_oracle.py).  Test infrastructure; the product runs the same
source on the GPU (tests/test_color_gpu.py)."""
import ctypes as C
import numpy as np
import pytest

from oracle import pyoracle as orc
import test_parse_emu as tpe


class Nclx(C.Structure):
    _fields_ = [("has_nclx", C.c_int), ("colour_primaries", C.c_int), ("transfer_characteristics", C.c_int),
                ("matrix_coefficients", C.c_int), ("full_range_flag", C.c_int)]


def _lib():
    L = tpe.emu()
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    np_ = C.POINTER(Nclx)
    L.hipdec_color_420_to_rgb24.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, np_, vp, sz, ci, vp]
    L.hipdec_color_ycbcr_to_rgb24_float.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, np_, vp, sz, ci, vp]
    L.hipdec_color_420_to_rrggbb.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, np_, vp, sz, ci, vp]
    L.hipdec_color_ycbcr_to_rrggbb_float.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, ci, np_, vp, sz, ci, vp]
    L.hipdec_color_bilinear_422_to_444.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    L.hipdec_color_hdr_to_rgb24.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, ci, np_, vp, sz, ci, ci, vp]
    L.hipdec_color_mono_to_rgb24.argtypes = [vp, sz, vp, sz, ci, ci, vp, sz, ci, vp]
    L.hipdec_color_bilinear_420_to_444.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    L.hipdec_color_to_sdr.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    L.emu_color_last_error.restype = C.c_char_p
    return L


def _planes(rng, w, h, bpp):
    dt = np.uint8 if bpp == 8 else np.uint16
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return [np.ascontiguousarray(rng.integers(0, 1 << bpp, s).astype(dt)) for s in ((h, w), (ch, cw), (ch, cw))]


def _args(planes):
    out = []
    for p in planes:
        out += [p.ctypes.data, p.strides[0]]
    return out


def _ok(L, rc):
    assert rc == 0, L.emu_color_last_error().decode()


@pytest.mark.parametrize("w,h", [(64, 48), (66, 50), (130, 34), (5, 3), (2, 2)])
@pytest.mark.parametrize("matrix,primaries", [(1, 1), (6, 1), (9, 9), (2, 2), (12, 1)])
def test_emulated_int_rgb24(w, h, matrix, primaries):
    L = _lib()
    y, cb, cr = _planes(np.random.default_rng(w + matrix), w, h, 8)
    nclx = (primaries, 13, matrix, 1)
    for alpha in (0, 1):
        bpp = 4 if alpha else 3
        out = np.zeros((h, w * bpp), np.uint8)
        _ok(L, L.hipdec_color_420_to_rgb24(*_args([y, cb, cr]), w, h, C.byref(Nclx(1, *nclx)), out.ctypes.data, out.strides[0], alpha, None))
        np.testing.assert_array_equal(out, orc.color_420_to_rgb24(y, cb, cr, nclx, alpha=bool(alpha)).reshape(h, -1))


@pytest.mark.parametrize("matrix,primaries,full", [(6, 1, 0), (1, 1, 0), (2, 2, 0), (9, 9, 0), (0, 1, 0), (0, 1, 1), (8, 1, 1), (12, 9, 0)])
def test_emulated_float_rgb24_bit_exact(matrix, primaries, full):
    L = _lib()
    w, h = 98, 42
    y, cb, cr = _planes(np.random.default_rng(matrix * 3 + full), w, h, 8)
    nclx = (primaries, 13, matrix, full)
    out = np.zeros((h, w * 3), np.uint8)
    _ok(L, L.hipdec_color_ycbcr_to_rgb24_float(*_args([y, cb, cr]), w, h, 1, C.byref(Nclx(1, *nclx)), out.ctypes.data, out.strides[0], 0, None))
    r, g, b = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, 1, nclx)
    np.testing.assert_array_equal(out, orc.color_rgb_planar_to_interleaved8(r, g, b).reshape(h, -1))


@pytest.mark.parametrize("bpp", [10, 12])
@pytest.mark.parametrize("le", [True, False])
def test_emulated_rrggbb(bpp, le):
    L = _lib()
    w, h = 70, 38
    y, cb, cr = _planes(np.random.default_rng(bpp + le), w, h, bpp)
    nclx = (9, 16, 9, 0)
    out = np.zeros((h, w * 6), np.uint8)
    _ok(L, L.hipdec_color_420_to_rrggbb(*_args([y, cb, cr]), w, h, bpp, C.byref(Nclx(1, *nclx)), out.ctypes.data, out.strides[0], int(le), None))
    np.testing.assert_array_equal(out, np.asarray(orc.color_420_to_rrggbb(y, cb, cr, bpp, nclx, little_endian=le)).reshape(h, -1))


@pytest.mark.parametrize("w,h", [(64, 48), (65, 49), (34, 130), (4, 4)])
@pytest.mark.parametrize("bpp", [8, 10])
def test_emulated_bilinear_and_sdr(w, h, bpp):
    L = _lib()
    y, cb, cr = _planes(np.random.default_rng(w + h + bpp), w, h, bpp)
    for plane in (cb, cr):
        out = np.zeros((h, w), plane.dtype)
        _ok(L, L.hipdec_color_bilinear_420_to_444(plane.ctypes.data, plane.strides[0], w, h, bpp, out.ctypes.data, out.strides[0], None))
        np.testing.assert_array_equal(out, orc.color_bilinear_420_to_444(plane, w, h))
    if bpp > 8:
        out = np.zeros((h, w), np.uint8)
        _ok(L, L.hipdec_color_to_sdr(y.ctypes.data, y.strides[0], w, h, bpp, out.ctypes.data, out.strides[0], None))
        np.testing.assert_array_equal(out, orc.color_to_sdr(y, bpp))
    # Op_YCbCr422_bilinear_to_YCbCr444: a chroma plane of (w + 1) / 2 x h samples
    L.hipdec_color_bilinear_422_to_444.argtypes = L.hipdec_color_bilinear_420_to_444.argtypes
    p422 = np.random.default_rng(w * 3 + h).integers(0, 1 << bpp, (h, (w + 1) // 2)).astype(cb.dtype)
    out = np.full((h, w + 2), 0xEE, cb.dtype)
    _ok(L, L.hipdec_color_bilinear_422_to_444(p422.ctypes.data, p422.strides[0], w, h, bpp, out.ctypes.data, out.strides[0], None))
    np.testing.assert_array_equal(out[:, :w], orc.color_bilinear_422_to_444(p422, w, h))
    assert (out[:, w:] == 0xEE).all()


def _pq_reference(code, bits):
    """SMPTE ST 2084 / BT.2100 table 4 in float64"""
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    e = code.astype(np.float64) / ((1 << bits) - 1)
    p = e ** (1 / m2)
    return (np.maximum(p - c1, 0) / (c2 - c3 * p)) ** (1 / m1)


def _hlg_reference(code, bits):
    """ARIB STD-B67 / BT.2100 HLG inverse OETF in fp64 (scene linear light, 1.0 = nominal peak)"""
    e = code.astype(np.float64) / ((1 << bits) - 1)
    a = 0.17883277
    b, c = 1 - 4 * a, 0.5 - a * np.log(4 * a)
    return np.where(e <= 0.5, e * e / 3.0, (np.exp((e - c) / a) + b) / 12.0)


def _bind_f4(L):
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    L.hipdec_color_to_hdr.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    L.hipdec_color_swap_endianness.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    L.hipdec_color_pq_to_linear.argtypes = [vp, sz, ci, ci, ci, ci, ci, vp, sz, vp]
    L.hipdec_color_hlg_to_linear.argtypes = [vp, sz, ci, ci, ci, ci, ci, vp, sz, vp]
    return L


@pytest.mark.parametrize("w,h", [(64, 48), (67, 5), (1, 1), (130, 33)])
def test_emulated_to_hdr_swap_and_pq(w, h):
    """the remaining colour ops of SURVEY §8(f4): Op_to_hdr_planes (hdr_sdr.cc:60-103), Op_RRGGBBaa_swap_endianness (rgb2rgb.cc:738-761) — both
    byte-exact restatements — and the PQ EOTF the reference does not have (published formula, stated tolerance 1e-6)"""
    L = _bind_f4(_lib())
    rng = np.random.default_rng(w * 7 + h)
    p8 = np.ascontiguousarray(rng.integers(0, 256, (h, w)).astype(np.uint8))
    for bits in (10, 12, 16):
        out = np.zeros((h, w), np.uint16)
        _ok(L, L.hipdec_color_to_hdr(p8.ctypes.data, p8.strides[0], w, h, bits, out.ctypes.data, out.strides[0], None))
        np.testing.assert_array_equal(out, (p8.astype(np.uint32) << (bits - 8)) | (p8.astype(np.uint32) >> (16 - bits)))
    for comps in (3, 4):
        px = np.ascontiguousarray(rng.integers(0, 1 << 16, (h, w * comps)).astype(np.uint16))
        out = np.zeros_like(px)
        _ok(L, L.hipdec_color_swap_endianness(px.ctypes.data, px.strides[0], w, h, comps, out.ctypes.data, out.strides[0], None))
        np.testing.assert_array_equal(out, px.byteswap())
    for bits, be in ((10, 0), (12, 1)):
        code = np.ascontiguousarray(rng.integers(0, 1 << bits, (h, w * 3)).astype(np.uint16))
        code[0, :3] = (0, (1 << bits) - 1, 1)
        src = code.byteswap() if be else code
        out = np.zeros((h, w * 3), np.float32)
        _ok(L, L.hipdec_color_pq_to_linear(src.ctypes.data, src.strides[0], w, h, 3, bits, be, out.ctypes.data, out.strides[0], None))
        np.testing.assert_allclose(out, _pq_reference(code, bits), rtol=1e-6, atol=1e-9)
        assert out[0, 0] == 0.0 and abs(out[0, 1] - 1.0) < 1e-6
    for bits, be in ((10, 0), (12, 1), (16, 0)):      # hybrid log-gamma: both branches of the curve, the table form (<= 12 bit) and the per-sample form
        code = np.ascontiguousarray(rng.integers(0, 1 << bits, (h, w * 3)).astype(np.uint16))
        code[0, :3] = (0, (1 << bits) - 1, ((1 << bits) - 1) // 2)
        src = code.byteswap() if be else code
        out = np.zeros((h, w * 3), np.float32)
        _ok(L, L.hipdec_color_hlg_to_linear(src.ctypes.data, src.strides[0], w, h, 3, bits, be, out.ctypes.data, out.strides[0], None))
        np.testing.assert_allclose(out, _hlg_reference(code, bits), rtol=1e-6, atol=1e-9)
        assert out[0, 0] == 0.0 and abs(out[0, 1] - 1.0) < 1e-6 and abs(out[0, 2] - 1.0 / 12.0) < 2e-3


# ---- > 8-bit planes of any chroma format -> RRGGBB: Op_YCbCr_to_RGB<uint16_t> + Op_RGB_HDR_to_RRGGBBaa_BE [+ swap], one pass ------------------
def _planes_cf(rng, w, h, bpp, chroma):
    cw = w if chroma == 3 else (w + 1) // 2
    ch = (h + 1) // 2 if chroma == 1 else h
    return [np.ascontiguousarray(rng.integers(0, 1 << bpp, s).astype(np.uint16)) for s in ((h, w), (ch, cw), (ch, cw))]


@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("nclx", [(9, 16, 9, 0), (1, 13, 6, 1), (1, 13, 0, 1), (1, 13, 8, 1), None], ids=["bt2020-limited", "bt601-full", "gbr", "ycgco", "none"])
@pytest.mark.parametrize("bpp,le", [(10, True), (12, False)])
def test_emulated_generic_rrggbb_equals_the_compiled_reference_pipeline(chroma, nclx, bpp, le):
    """the host-compiled kernel against libheif's own convert_colorspace() (oracle/_ref) on the same planes: this pins the planner rule too
    (which ops the reference chains for these states, default options) — and the colour oracle's restatement"""
    import ref_harness as rh
    if not rh.available():
        pytest.skip("oracle/_ref not built")
    L = _lib()
    w, h = 70, 38
    y, cb, cr = _planes_cf(np.random.default_rng(bpp + chroma), w, h, bpp, chroma)
    out = np.zeros((h, w * 6), np.uint8)
    ns = Nclx(1, *nclx) if nclx else Nclx(0, 2, 2, 2, 1)
    _ok(L, L.hipdec_color_ycbcr_to_rrggbb_float(*_args([y, cb, cr]), w, h, bpp, chroma, C.byref(ns), out.ctypes.data, out.strides[0], int(le), None))
    ref = rh.convert([y, cb, cr], bpp, chroma, nclx, rh.CS_RGB, rh.CH_RRGGBB_LE if le else rh.CH_RRGGBB_BE)[0]
    np.testing.assert_array_equal(out, ref[:, :w * 6])
    r, g, b = orc.color_ycbcr_to_rgb_planar(y, cb, cr, bpp, chroma, nclx)
    inter = np.stack([r, g, b], axis=-1).astype(np.uint16)
    np.testing.assert_array_equal(out, (inter if le else inter.byteswap()).reshape(h, -1).view(np.uint8))


def test_emulated_422_bilinear_then_generic_rrggbb_equals_the_reference_with_only_preferred_upsampling():
    """only_use_preferred_chroma_algorithm with bilinear: the reference upsamples the 16-bit chroma planes first (Op_YCbCr422_bilinear_to_YCbCr444)"""
    import ref_harness as rh
    if not rh.available():
        pytest.skip("oracle/_ref not built")
    L = _lib()
    w, h, bpp, nclx = 70, 38, 10, (9, 16, 9, 0)
    y, cb, cr = _planes_cf(np.random.default_rng(5), w, h, bpp, 2)
    up = [np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)]
    for src, dst in zip((cb, cr), up):
        _ok(L, L.hipdec_color_bilinear_422_to_444(src.ctypes.data, src.strides[0], w, h, bpp, dst.ctypes.data, dst.strides[0], None))
    out = np.zeros((h, w * 6), np.uint8)
    _ok(L, L.hipdec_color_ycbcr_to_rrggbb_float(*_args([y, up[0], up[1]]), w, h, bpp, 3, C.byref(Nclx(1, *nclx)), out.ctypes.data, out.strides[0], 1, None))
    ref = rh.convert([y, cb, cr], bpp, 2, nclx, rh.CS_RGB, rh.CH_RRGGBB_LE, upsampling=rh.UPS_BILINEAR, only_preferred=True)[0]
    np.testing.assert_array_equal(out, ref[:, :w * 6])


# ---- > 8-bit planes -> 8-bit interleaved RGB: which chain the reference's search ends on, and the fused kernels that run it --------------------
def _run_plan_emu(L, planes, bpp, chroma, nclx, steps, w, h):
    """the plan of libheif_amd/color.py:plan executed with the host-compiled kernels (what hipdec_color_convert does on the device)"""
    ns = Nclx(1, *nclx) if nclx else Nclx(0, 2, 2, 2, 1)
    from libheif_amd import color
    y, cb, cr = planes
    steps = list(steps)
    first = steps[0]
    if steps[0] == "Op_to_sdr_planes":
        if steps[1:] in (["Op_YCbCr420_to_RGB24"],):                      # fused: to_sdr + the integer op
            out = np.zeros((h, w * 3), np.uint8)
            _ok(L, L.hipdec_color_hdr_to_rgb24(*_args([y, cb, cr]), w, h, bpp, chroma, C.byref(ns), out.ctypes.data, out.strides[0], 0, 1, None))
            return out
        sdr = []
        for p in (y, cb, cr):
            o = np.zeros(p.shape, np.uint8)
            _ok(L, L.hipdec_color_to_sdr(p.ctypes.data, p.strides[0], p.shape[1], p.shape[0], bpp, o.ctypes.data, o.strides[0], None))
            sdr.append(o)
        y, cb, cr = sdr
        bpp = 8
        steps = steps[1:]
    if steps[0].endswith("bilinear_to_YCbCr444"):
        up = L.hipdec_color_bilinear_420_to_444 if "420" in steps[0] else L.hipdec_color_bilinear_422_to_444
        new = []
        for p in (cb, cr):
            o = np.zeros((h, w), p.dtype)
            _ok(L, up(p.ctypes.data, p.strides[0], w, h, bpp, o.ctypes.data, o.strides[0], None))
            new.append(o)
        cb, cr = new
        chroma = 3
        steps = steps[1:]
    out = np.zeros((h, w * 3), np.uint8)
    if steps[0] is not first:      # not the chain's first op: the pipeline attached the replaced profile to the intermediate image
        ns = Nclx(1, *color.replaced_nclx(nclx))
    if steps[0] == "Op_YCbCr420_to_RGB24":
        _ok(L, L.hipdec_color_420_to_rgb24(*_args([y, cb, cr]), w, h, C.byref(ns), out.ctypes.data, out.strides[0], 0, None))
    elif steps[0] == "Op_YCbCr_to_RGB<u8>":
        _ok(L, L.hipdec_color_ycbcr_to_rgb24_float(*_args([y, cb, cr]), w, h, chroma, C.byref(ns), out.ctypes.data, out.strides[0], 0, None))
    elif steps[:2] == ["Op_YCbCr_to_RGB<u16>", "Op_to_sdr_planes"]:
        _ok(L, L.hipdec_color_hdr_to_rgb24(*_args([y, cb, cr]), w, h, bpp, chroma, C.byref(ns), out.ctypes.data, out.strides[0], 0, 0, None))
    else:
        raise AssertionError(steps)
    return out


@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("bpp", [10, 12])
def test_emulated_hdr_to_rgb24_follows_the_reference_pipeline_in_every_state(chroma, bpp):
    """> 8-bit planes -> RGB24 over colour profiles x upsampling options: the chain the Python mirror plans (the C planner is checked against the
    mirror in tests/test_color_boundary.py), run with the host-compiled kernels, equals libheif's own convert_colorspace() bit for bit.
    (Round 2's rule - Op_to_sdr_planes always first - was only right for full-range 4:2:0.)"""
    import ref_harness as rh
    from libheif_amd import color
    if not rh.available():
        pytest.skip("oracle/_ref not built")
    L = _lib()
    w, h = 70, 38
    planes = _planes_cf(np.random.default_rng(bpp * 7 + chroma), w, h, bpp, chroma)
    n = 0
    for nclx in ((9, 16, 9, 0), (9, 16, 9, 1), (1, 13, 6, 1), (1, 13, 1, 0), (1, 13, 0, 1), (1, 13, 12, 0), None):
        for ups, only in ((rh.UPS_BILINEAR, False), (rh.UPS_NN, False), (rh.UPS_BILINEAR, True), (rh.UPS_NN, True)):
            try:
                ref = rh.convert(planes, bpp, chroma, nclx, rh.CS_RGB, rh.CH_RGB, upsampling=ups, only_preferred=only)[0][:, :w * 3]
            except RuntimeError:
                continue          # a state the reference itself has no pipeline for
            steps = color.plan(bpp, chroma, nclx, 10, ups, only)
            np.testing.assert_array_equal(_run_plan_emu(L, planes, bpp, chroma, nclx, steps, w, h), ref, err_msg=str((nclx, ups, only, steps)))
            n += 1
    assert n >= 24


@pytest.mark.parametrize("w,h", [(70, 38), (64, 48), (5, 3)])
@pytest.mark.parametrize("alpha_plane", [False, True], ids=["no-alpha", "alpha"])
def test_emulated_mono_to_rgb_equals_the_compiled_reference_pipeline(w, h, alpha_plane):
    """Op_mono_to_RGB24_32: what libheif's own convert_colorspace() makes of an 8-bit monochrome image (with and without an alpha plane)"""
    import ref_harness as rh
    if not rh.available():
        pytest.skip("oracle/_ref not built")
    L = _lib()
    rng = np.random.default_rng(w)
    y = np.ascontiguousarray(rng.integers(0, 256, (h, w)).astype(np.uint8))
    a = np.ascontiguousarray(rng.integers(0, 256, (h, w)).astype(np.uint8))
    for tgt, bpp in ((rh.CH_RGBA, 4),) + (() if alpha_plane else ((rh.CH_RGB, 3),)):
        out = np.zeros((h, w * bpp), np.uint8)
        _ok(L, L.hipdec_color_mono_to_rgb24(y.ctypes.data, y.strides[0], a.ctypes.data if alpha_plane else None, a.strides[0] if alpha_plane else 0, w, h,
                                            out.ctypes.data, out.strides[0], int(bpp == 4), None))
        if alpha_plane:
            continue        # (the harness adds planes as Y, Cb, Cr, alpha: a monochrome image with alpha is checked by value below)
        ref = rh.convert([y], 8, 0, (1, 13, 6, 1), rh.CS_RGB, tgt, in_colorspace=rh.CS_MONOCHROME)[0][:, :w * bpp]
        np.testing.assert_array_equal(out, ref)
    if alpha_plane:
        px = out.reshape(h, w, 4)
        for c in range(3):
            np.testing.assert_array_equal(px[:, :, c], y)
        np.testing.assert_array_equal(px[:, :, 3], a)
