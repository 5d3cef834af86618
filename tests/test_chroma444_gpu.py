"""GPU parity for 4:4:4 (chroma_format_idc 3) stills through the C ABI and through an unmodified libheif: bit-exact planes against the CPU
oracle over the coding-tool matrix, batches mixing chroma formats, the per-unit maps (one intra_chroma_pred_mode per NxN partition), the
colour stage on 4:4:4 planes, a full-HD still, and heif_decode_image() handing out heif_chroma_444 planes.  See tests/test_chroma444_emu.py
for what differs from 4:2:0 in the syntax and the decoding process."""
import numpy as np
import pytest

from oracle import pyoracle as orc
import heic_util as hu
import libheif_host as lh
from test_chroma444_emu import CONFIGS

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


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: ",".join("%s=%s" % kv for kv in c.items()) or "default")
@pytest.mark.parametrize("size", [(200, 136), (75, 41)])
def test_decode_444_matches_oracle(cfg, size):
    bd = cfg.get("bit_depth", 8)
    stream = orc.encode(orc.synth_image(size[0], size[1], bd, 3, seed=3 + size[0]), **cfg)
    ref = orc.decode(stream)
    img = _decode_gpu(stream)
    assert img.info["chroma_format_idc"] == 3
    assert (img.info["width"], img.info["height"], img.info["chroma_width"], img.info["chroma_height"]) == (ref["width"], ref["height"], ref["width"], ref["height"])
    assert img.nclx == ref["nclx"]
    assert len(img.planes) == 3
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="component %d" % c)


def test_batch_mixing_chroma_formats_matches_oracle():
    """4:4:4, 4:2:0 and 4:0:0 pictures in ONE batch (one set of launches): three, two or one reconstruction wave chains per row chain"""
    from libheif_amd.decoder import Batch
    streams = []
    for i, (w, h, cf) in enumerate([(128, 64, 3), (64, 128, 1), (200, 136, 3), (72, 40, 0), (136, 72, 1), (75, 41, 3), (264, 200, 3), (264, 200, 1)]):
        streams.append(orc.encode(orc.synth_image(w, h, 8, cf, seed=40 + i), qp=24 + 2 * i, stress=i & 1, tile_cols=1 + (i % 2), wpp=(i >> 1) & 1))
    b = Batch(streams)
    b.run(); b.status()
    for i, s in enumerate(streams):
        ref = orc.decode(s)
        got = b.planes(i)
        assert len(got) == len(ref["planes"])
        for c in range(len(got)):
            np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))


def test_444_intermediate_maps_and_taps_match_oracle():
    from libheif_amd.decoder import Batch
    stream = orc.encode(orc.synth_image(200, 136, 8, 3, seed=11), stress=1, transform_skip=1, lossless_pct=10)
    ref = orc.decode(stream, taps=True)
    b = Batch([stream])
    b.run(); b.status()
    m = b.maps(0)
    np.testing.assert_array_equal(m["log2_cb"], ref["map_log2_cb"])
    np.testing.assert_array_equal(m["log2_tb"], ref["map_log2_tb"])
    np.testing.assert_array_equal(m["intra_luma"], ref["map_intra_luma"])
    np.testing.assert_array_equal(m["intra_chroma"], ref["map_intra_chroma"])     # per partition for NxN coding units
    np.testing.assert_array_equal(m["qp_y"], ref["map_qp_y"])
    np.testing.assert_array_equal(m["flags"] & 0x7f, ref["map_flags"] & 0x7f)
    for c in range(3):
        np.testing.assert_array_equal(b.tap(0, c), ref["post_deblock"][c])


@pytest.mark.parametrize("vui", [(1, 13, 6, 1), (1, 13, 1, 0), None], ids=["bt601-full", "bt709-limited", "unspecified"])
def test_444_planes_to_rgb24_match_the_oracle_chain(vui):
    """Op_YCbCr_to_RGB<uint8_t> + Op_RGB_to_RGB24_32 — the chain libheif's planner has for 4:4:4 planes — on the decoded planes in HBM"""
    from libheif_amd.decoder import Batch
    kw = dict(vui_primaries=vui[0], vui_transfer=vui[1], vui_matrix=vui[2], vui_full_range=vui[3]) if vui else {}
    stream = orc.encode(orc.synth_image(200, 136, 8, 3, seed=21), **kw)
    ref = orc.decode(stream)
    b = Batch([stream]); b.run(); b.status()
    rgb = b.to_rgb(0, 10)
    y, cb, cr = ref["planes"]
    r, g, bb = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, 3, ref["nclx"])
    np.testing.assert_array_equal(rgb, orc.color_rgb_planar_to_interleaved8(r, g, bb).reshape(136, -1))


def test_444_main10_to_rrggbb_is_refused_loudly():
    from libheif_amd.decoder import Batch
    from libheif_amd._capi import HipDecError
    stream = orc.encode(orc.synth_image(72, 40, 10, 3, seed=2), bit_depth=10)
    b = Batch([stream]); b.run(); b.status()
    with pytest.raises(HipDecError) as e:
        b.to_rgb(0, 12)
    assert "4:2:0" in str(e.value)
    for c, p in enumerate(b.planes(0)):           # the planes themselves are there
        np.testing.assert_array_equal(p, orc.decode(stream)["planes"][c])


def test_full_hd_444_still_matches_oracle():
    stream = orc.encode(orc.synth_image(1920, 1080, 8, 3, seed=77), qp=30)
    ref = orc.decode(stream)
    img = _decode_gpu(stream)
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="component %d" % c)


SRGB_VUI = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)


@needs_ref
@pytest.mark.parametrize("bd", [8, 10])
def test_heif_decode_image_hands_out_444_planes(bd):
    """an unmodified libheif + the plugin: a 4:4:4 HEIC item arrives as heif_chroma_444 planes, bit-exact"""
    lh.load_hip_plugin()
    w, h = 264, 200
    s = orc.encode(orc.synth_image(w, h, bd, 3, seed=9), bit_depth=bd, stress=1, **SRGB_VUI)
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, w, h)], bit_depth=bd, chroma_format_idc=3), lh.COLORSPACE_YCBCR, 3)
    assert out["bit_depth"] == bd and len(out["planes"]) == 3
    for c in range(3):
        np.testing.assert_array_equal(out["planes"][c], ref["planes"][c], err_msg="component %d" % c)


@needs_ref
def test_heif_decode_image_444_to_rgb_matches_reference_colour_ops_on_oracle_planes():
    import ref_harness as rh
    lh.load_hip_plugin()
    w, h = 200, 136
    s = orc.encode(orc.synth_image(w, h, 8, 3, seed=5), **SRGB_VUI)
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, w, h)], chroma_format_idc=3), lh.COLORSPACE_RGB, lh.CHROMA_RGB)
    exp = rh.convert(ref["planes"], 8, rh.CH_444, ref["nclx"], rh.CS_RGB, rh.CH_RGB)[0]
    np.testing.assert_array_equal(out["rgb"], exp[:, :w * 3])
