"""GPU parity for 4:2:2 and 4:4:4 (chroma_format_idc 2 / 3) stills through the C ABI and through an unmodified libheif: bit-exact planes
against the CPU oracle over the coding-tool matrix, batches mixing chroma formats, the per-unit maps (4:4:4: one intra_chroma_pred_mode per
NxN partition; 4:2:2: Table 8-3 modes), the colour stage on the decoded planes, full-HD stills, and heif_decode_image() handing out
heif_chroma_422 / heif_chroma_444 planes.  See tests/test_chroma_formats_emu.py for what differs from 4:2:0 in the syntax and the decoding process."""
import numpy as np
import pytest

from oracle import pyoracle as orc
import heic_util as hu
import libheif_host as lh
from test_chroma_formats_emu import CONFIGS, CONFIGS_422, FORMATS

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")


def _decode_gpu(stream):
    from libheif_amd.decoder import HipDecoder
    d = HipDecoder()
    d.push_data(stream)
    img = d.decode_next_image()
    assert d.decode_next_image() is None
    d.free()
    return img


def _ids(v):
    return ("4%d%d" % ((2, 2) if v == 2 else (4, 4))) if isinstance(v, int) else (",".join("%s=%s" % kv for kv in v.items()) or "default")


@pytest.mark.parametrize("cf,cfg", [(cf, c) for cf in (2, 3) for c in (CONFIGS if cf == 3 else CONFIGS_422)], ids=_ids)
@pytest.mark.parametrize("size", [(200, 136), (74, 41)])
def test_decode_matches_oracle(cf, cfg, size):
    bd = cfg.get("bit_depth", 8)
    stream = orc.encode(orc.synth_image(size[0], size[1], bd, cf, seed=3 + size[0]), **cfg)
    ref = orc.decode(stream)
    img = _decode_gpu(stream)
    assert img.info["chroma_format_idc"] == cf
    assert (img.info["width"], img.info["height"]) == (ref["width"], ref["height"])
    assert (img.info["chroma_width"], img.info["chroma_height"]) == (ref["width"] if cf == 3 else ref["width"] // 2, ref["height"])
    assert img.nclx == ref["nclx"]
    assert len(img.planes) == 3
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="component %d" % c)


def test_batch_mixing_chroma_formats_matches_oracle():
    """4:4:4, 4:2:2, 4:2:0 and 4:0:0 pictures in ONE batch (one set of launches): three, two or one reconstruction wave chains per row chain"""
    from libheif_amd.decoder import Batch
    streams = []
    for i, (w, h, cf) in enumerate([(128, 64, 3), (64, 128, 1), (200, 136, 2), (72, 40, 0), (136, 72, 1), (75, 41, 3), (264, 200, 3), (264, 200, 1), (70, 41, 2), (264, 200, 2)]):
        streams.append(orc.encode(orc.synth_image(w, h, 8, cf, seed=40 + i), qp=24 + 2 * i, stress=i & 1, tile_cols=1 + (i % 2), wpp=(i >> 1) & 1))
    b = Batch(streams)
    b.run(); b.status()
    for i, s in enumerate(streams):
        ref = orc.decode(s)
        got = b.planes(i)
        assert len(got) == len(ref["planes"])
        for c in range(len(got)):
            np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))


@FORMATS
def test_intermediate_maps_and_taps_match_oracle(cf):
    from libheif_amd.decoder import Batch
    stream = orc.encode(orc.synth_image(200, 136, 8, cf, seed=11), stress=1, transform_skip=1, lossless_pct=10)
    ref = orc.decode(stream, taps=True)
    b = Batch([stream])
    b.run(); b.status()
    m = b.maps(0)
    np.testing.assert_array_equal(m["log2_cb"], ref["map_log2_cb"])
    np.testing.assert_array_equal(m["log2_tb"], ref["map_log2_tb"])
    np.testing.assert_array_equal(m["intra_luma"], ref["map_intra_luma"])
    np.testing.assert_array_equal(m["intra_chroma"], ref["map_intra_chroma"])     # 4:4:4: per partition for NxN coding units; 4:2:2: after Table 8-3
    np.testing.assert_array_equal(m["qp_y"], ref["map_qp_y"])
    fmask = 0x79 if cf == 2 else 0x7f    # 4:2:2: a unit's cbf_cb / cbf_cr bits are those of ONE of its two chroma blocks (the other's sit in unit z ^ 1)
    np.testing.assert_array_equal(m["flags"] & fmask, ref["map_flags"] & fmask)
    for c in range(3):
        np.testing.assert_array_equal(b.tap(0, c), ref["post_deblock"][c])


@FORMATS
@pytest.mark.parametrize("vui", [(1, 13, 6, 1), (1, 13, 1, 0), None], ids=["bt601-full", "bt709-limited", "unspecified"])
def test_planes_to_rgb24_match_the_oracle_chain(vui, cf):
    """Op_YCbCr_to_RGB<uint8_t> + Op_RGB_to_RGB24_32 — the chain libheif's planner has for 4:2:2 / 4:4:4 planes — on the decoded planes in HBM"""
    from libheif_amd.decoder import Batch
    kw = dict(vui_primaries=vui[0], vui_transfer=vui[1], vui_matrix=vui[2], vui_full_range=vui[3]) if vui else {}
    stream = orc.encode(orc.synth_image(200, 136, 8, cf, seed=21), **kw)
    ref = orc.decode(stream)
    b = Batch([stream]); b.run(); b.status()
    rgb = b.to_rgb(0, 10)
    y, cb, cr = ref["planes"]
    r, g, bb = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, cf, ref["nclx"])
    np.testing.assert_array_equal(rgb, orc.color_rgb_planar_to_interleaved8(r, g, bb).reshape(136, -1))


@pytest.mark.parametrize("cfs", [(3, 3, 3), (2, 2, 2), (1, 3, 2, 1)], ids=["444", "422", "mixed"])
def test_run_rgb_on_batches_with_other_chroma_formats(cfs):
    """hipdec_batch_run_rgb: such batches take the unfused path (decode, then ONE batched colour launch); every item's RGB equals the per-item
    colour stage, which the test above pins to the oracle chain"""
    from libheif_amd.decoder import Batch
    vui = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    streams = [orc.encode(orc.synth_image(200 + 64 * k, 136, 8, cf, seed=60 + k), **vui) for k, cf in enumerate(cfs)]
    f = Batch(streams); f.alloc_rgb(10); f.run_rgb(); f.status()
    f.run_rgb(); f.status()
    assert f.kernel_timing_us()["colour"] > 0.0                 # not the fused SAO + RGB kernel
    for i, st in enumerate(streams):
        one = Batch([st]); one.run(); one.status()
        np.testing.assert_array_equal(f.rgb(i), one.to_rgb(0, 10))
        ref = orc.decode(st)
        for c in range(3):
            np.testing.assert_array_equal(f.planes(i)[c], ref["planes"][c])
        one.free()
    f.free()


@FORMATS
@pytest.mark.parametrize("out_chroma", [14, 12], ids=["RRGGBB_LE", "RRGGBB_BE"])
def test_main10_planes_to_rrggbb_match_the_oracle_chain(cf, out_chroma):
    """Op_YCbCr_to_RGB<uint16_t> + Op_RGB_HDR_to_RRGGBBaa_BE [+ swap] on the decoded 10-bit planes in HBM (tests/test_color_emu.py pins the chain
    to the compiled reference pipeline)"""
    from libheif_amd.decoder import Batch
    stream = orc.encode(orc.synth_image(200, 136, 10, cf, seed=2), bit_depth=10, vui_primaries=9, vui_transfer=16, vui_matrix=9, vui_full_range=0)
    ref = orc.decode(stream)
    b = Batch([stream]); b.run(); b.status()
    got = b.to_rgb(0, out_chroma)
    r, g, bb = orc.color_ycbcr_to_rgb_planar(*ref["planes"], 10, cf, ref["nclx"])
    inter = np.stack([r, g, bb], axis=-1).astype(np.uint16)
    np.testing.assert_array_equal(got, (inter if out_chroma == 14 else inter.byteswap()).reshape(136, -1).view(np.uint8))
    b.free()


@pytest.mark.parametrize("cf", [1, 2, 3], ids=["420", "422", "444"])
@pytest.mark.parametrize("full", [0, 1], ids=["limited", "full"])
def test_main10_planes_to_rgb24_follow_the_reference_chain(cf, full):
    """> 8-bit planes to 8-bit RGB, one fused pass: Op_to_sdr_planes + the integer op for full-range 4:2:0, the generic op at 10 bits with
    Op_to_sdr_planes behind it for everything else (tests/test_color_emu.py pins the rule to the compiled reference pipeline)"""
    from libheif_amd.decoder import Batch
    stream = orc.encode(orc.synth_image(200, 136, 10, cf, seed=6), bit_depth=10, vui_primaries=9, vui_transfer=16, vui_matrix=9, vui_full_range=full)
    ref = orc.decode(stream)
    b = Batch([stream]); b.run(); b.status()
    got = b.to_rgb(0, 10)
    y, cb, cr = ref["planes"]
    if cf == 1 and full:
        exp = orc.color_420_to_rgb24(*[(p >> 2).astype(np.uint8) for p in (y, cb, cr)], ref["nclx"]).reshape(136, -1)
    else:
        r, g, bb = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 10, cf, ref["nclx"])
        exp = np.stack([r >> 2, g >> 2, bb >> 2], axis=-1).astype(np.uint8).reshape(136, -1)
    np.testing.assert_array_equal(got, exp)
    b.free()


@FORMATS
def test_color_boundary_takes_the_decoded_device_planes(cf):
    """hipdec_color_convert (the colour boundary libheif's HIP op forwards to) on device planes of the new formats: Main10 -> RGB24 goes through
    Op_to_sdr_planes first, then the generic op; equals the same chain on the oracle's planes through the Python mirror"""
    import ctypes as C
    import libheif_amd
    from libheif_amd import color
    from libheif_amd._capi import DeviceBuffer, check, Nclx
    nclx = (9, 16, 9, 0)
    stream = orc.encode(orc.synth_image(200, 136, 10, cf, seed=4), bit_depth=10, vui_primaries=9, vui_transfer=16, vui_matrix=9, vui_full_range=0)
    ref = orc.decode(stream)
    class Img(C.Structure):
        _fields_ = [("width", C.c_int), ("height", C.c_int), ("chroma", C.c_int), ("bit_depth", C.c_int), ("plane", C.c_void_p * 4), ("stride", C.c_size_t * 4),
                    ("on_device", C.c_int)]
    lib = libheif_amd.load_library()
    lib.hipdec_color_convert.argtypes = [C.POINTER(Img), C.POINTER(Nclx), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    planes = [np.ascontiguousarray(p) for p in _decode_gpu(stream).planes]
    for c in range(3):
        np.testing.assert_array_equal(planes[c], ref["planes"][c])
    for out_chroma, bpp in ((10, 3), (14, 6), (12, 6)):
        want = color.convert_colorspace(ref["planes"], 10, cf, nclx, out_chroma)          # the Python mirror: op by op
        img = Img(200, 136, cf, 10, (C.c_void_p * 4)(*[p.ctypes.data for p in planes], None), (C.c_size_t * 4)(*[p.strides[0] for p in planes], 0), 0)
        got = np.zeros((136, 200 * bpp), np.uint8)
        check(lib.hipdec_color_convert(C.byref(img), C.byref(Nclx(1, *nclx)), out_chroma, 2, 0, got.ctypes.data, got.strides[0], 0))   # the C boundary: plan + chain
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("cf,bd", [(3, 8), (2, 10)], ids=["444-8bit", "422-10bit"])
def test_full_hd_still_matches_oracle(cf, bd):
    """(4:2:2 10-bit is what cameras that write HEIF with more than 4:2:0 produce)"""
    stream = orc.encode(orc.synth_image(1920, 1080, bd, cf, seed=77), qp=30, bit_depth=bd)
    ref = orc.decode(stream)
    img = _decode_gpu(stream)
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="component %d" % c)


SRGB_VUI = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)


@needs_ref
@FORMATS
@pytest.mark.parametrize("bd", [8, 10])
def test_heif_decode_image_hands_out_the_planes(bd, cf):
    """an unmodified libheif + the plugin: a 4:2:2 / 4:4:4 HEIC item arrives as heif_chroma_422 / heif_chroma_444 planes, bit-exact"""
    lh.load_hip_plugin()
    w, h = 264, 200
    s = orc.encode(orc.synth_image(w, h, bd, cf, seed=9), bit_depth=bd, stress=1, **SRGB_VUI)
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, w, h)], bit_depth=bd, chroma_format_idc=cf), lh.COLORSPACE_YCBCR, cf)
    assert out["bit_depth"] == bd and len(out["planes"]) == 3
    for c in range(3):
        np.testing.assert_array_equal(out["planes"][c], ref["planes"][c], err_msg="component %d" % c)


@needs_ref
@FORMATS
def test_heif_decode_image_to_rgb_matches_reference_colour_ops_on_oracle_planes(cf):
    import ref_harness as rh
    lh.load_hip_plugin()
    w, h = 200, 136
    s = orc.encode(orc.synth_image(w, h, 8, cf, seed=5), **SRGB_VUI)
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, w, h)], chroma_format_idc=cf), lh.COLORSPACE_RGB, lh.CHROMA_RGB)
    exp = rh.convert(ref["planes"], 8, cf, ref["nclx"], rh.CS_RGB, rh.CH_RGB)[0]
    np.testing.assert_array_equal(out["rgb"], exp[:, :w * 3])


@FORMATS
@pytest.mark.parametrize("bd", [8, 10])
def test_device_grid_canvas_takes_tiles_of_the_format(cf, bd):
    """hipdec_grid_*: 2 x 3 tiles of the format decoded as one set of launches and pasted into a canvas with the tiles' subsampling (clipped to the
    output size), then the colour stage over the whole canvas — against the oracle's tiles pasted the same way"""
    from libheif_amd.grid import GridDecoderC, GridLayout
    from libheif_amd import color
    rows, cols, tw, th = 2, 3, 128, 64
    ow, oh = cols * tw - 6, rows * th - 10
    nclx = (9, 16, 9, 0) if bd > 8 else (1, 13, 6, 1)
    vui = dict(vui_primaries=nclx[0], vui_transfer=nclx[1], vui_matrix=nclx[2], vui_full_range=nclx[3])
    streams = {t: orc.encode(orc.synth_image(tw, th, bd, cf, seed=50 + t), bit_depth=bd, wpp=t % 2, **vui) for t in range(rows * cols)}
    g = GridDecoderC(streams, GridLayout(rows, cols, tw, th, ow, oh, bit_depth=bd), [0])
    g.decode(); g.wait()
    sw = 1 if cf == 3 else 2
    canvas = [np.zeros((rows * th, cols * tw), np.uint16), np.zeros((rows * th, cols * tw // sw), np.uint16), np.zeros((rows * th, cols * tw // sw), np.uint16)]
    for t, s in streams.items():
        ref = orc.decode(s)
        r, c = divmod(t, cols)
        canvas[0][r * th:(r + 1) * th, c * tw:(c + 1) * tw] = ref["planes"][0]
        for k in (1, 2):
            canvas[k][r * th:(r + 1) * th, c * tw // sw:(c + 1) * tw // sw] = ref["planes"][k]
    got = g.planes()
    want = [canvas[0][:oh, :ow], canvas[1][:oh, :(ow + sw - 1) // sw], canvas[2][:oh, :(ow + sw - 1) // sw]]
    for k in range(3):
        np.testing.assert_array_equal(got[k], want[k], err_msg="plane %d" % k)
    out_chroma = 10 if bd == 8 else 14
    rgb = g.to_rgb(out_chroma)
    np.testing.assert_array_equal(rgb, color.convert_colorspace(want, bd, cf, nclx, out_chroma, upsampling=1))
    g.free()


def test_monochrome_stills_go_to_rgb_through_the_mono_op():
    """Op_mono_to_RGB24_32 on the decoded 4:0:0 plane (R = G = B = Y; RGBA: alpha 0xFF), alone and inside a batch's ONE colour launch beside
    colour items (tests/test_color_emu.py pins the op to the compiled reference pipeline)"""
    from libheif_amd.decoder import Batch
    mono = orc.encode(orc.synth_image(200, 136, 8, 0, seed=8))
    col = orc.encode(orc.synth_image(200, 136, 8, 1, seed=9), vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    y = orc.decode(mono)["planes"][0]
    b = Batch([mono]); b.run(); b.status()
    np.testing.assert_array_equal(b.to_rgb(0, 10), np.repeat(y, 3, axis=1))
    rgba = b.to_rgb(0, 11).reshape(136, 200, 4)
    for c in range(3):
        np.testing.assert_array_equal(rgba[:, :, c], y)
    assert (rgba[:, :, 3] == 255).all()
    b.free()
    f = Batch([mono, col, mono]); f.alloc_rgb(10); f.run_rgb(); f.status()
    one = Batch([col]); one.run(); one.status()
    np.testing.assert_array_equal(f.rgb(0), np.repeat(y, 3, axis=1))
    np.testing.assert_array_equal(f.rgb(2), np.repeat(y, 3, axis=1))
    np.testing.assert_array_equal(f.rgb(1), one.to_rgb(0, 10))
    f.free(); one.free()
