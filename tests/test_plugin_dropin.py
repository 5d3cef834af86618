"""Drop-in boundary: libheifhip.so as a heif_decoder_plugin under the REAL reference libheif
(oracle/_ref/libheif.so).  CPU part: the plugin loads through heif_load_plugin, registers as an HEVC
decoder, the synthetic HEIC files parse, and decoding without a GPU fails loudly.  GPU part
(-m gpu): heif_decode_image() end to end — single items, grids (libheif's own tile fan-out and
canvas paste), RGB output through libheif's own colour pipeline — bit-exact against the oracle."""
import numpy as np
import pytest

from oracle import pyoracle as orc
import heic_util as hu
import libheif_host as lh

needs_ref = pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")


# libheif's default decoding options convert every decoded image to the sRGB nclx
# (context.cc:1533-1558: output_image_nclx_profile == NULL and no passthrough), so the YCbCr parity cases signal
# exactly that profile in the VUI: then heif_decode_image() hands the plugin's planes through untouched.
SRGB_VUI = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)


def _still(w, h, seed=1, **cfg):
    cfg = dict(SRGB_VUI, **cfg)
    return orc.encode(orc.synth_image(w, h, cfg.get("bit_depth", 8), 1, seed=seed), **cfg)


@needs_ref
def test_synthetic_heic_files_parse_in_the_reference():
    s = _still(200, 136)
    assert lh.primary_size(hu.build_heic([(s, 200, 136)])) == (200, 136)
    tiles = [(_still(128, 128, seed=i), 128, 128) for i in range(6)]
    assert lh.primary_size(hu.build_heic(tiles, grid=(2, 3, 380, 250))) == (380, 250)


@needs_ref
def test_plugin_loads_and_registers_as_hevc_decoder():
    L = lh.load_hip_plugin()
    assert L.heif_have_decoder_for_format(lh.COMPRESSION_HEVC) == 1


@needs_ref
def test_decode_without_gpu_fails_loudly():
    import libheif_amd
    if libheif_amd.load_library().hipdec_device_count() > 0:
        pytest.skip("a GPU is present")
    lh.load_hip_plugin()
    with pytest.raises(lh.LibheifError) as e:
        lh.decode(hu.build_heic([(_still(64, 64), 64, 64)]))
    assert "no HIP device" in str(e.value)


@needs_ref
def test_plugin_abi_matches_reference_headers(reference_dir):
    """struct layouts restated in include/heif_plugin_abi.h against the real headers (compiled check)."""
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r'''
#include <cstddef>
#include <libheif/heif.h>
#include <libheif/heif_plugin.h>
#include "heif_plugin_abi.h"
#define SAME(a, b, f) static_assert(offsetof(a, f) == offsetof(b, f), #f)
static_assert(sizeof(hp_error) == sizeof(heif_error), "heif_error");
static_assert(sizeof(hp_decoder_plugin) == sizeof(heif_decoder_plugin), "heif_decoder_plugin");
static_assert(sizeof(hp_plugin_info) == sizeof(heif_plugin_info), "heif_plugin_info");
static_assert(sizeof(hp_decoder_options) == sizeof(heif_decoder_plugin_options), "options");
SAME(hp_decoder_plugin, heif_decoder_plugin, id_name);
SAME(hp_decoder_plugin, heif_decoder_plugin, decode_next_image);
SAME(hp_decoder_plugin, heif_decoder_plugin, minimum_required_libheif_version);
SAME(hp_decoder_plugin, heif_decoder_plugin, new_decoder2);
SAME(hp_decoder_plugin, heif_decoder_plugin, decode_next_image2);
SAME(hp_security_limits_head, heif_security_limits, max_image_size_pixels);
SAME(hp_nclx_head, heif_color_profile_nclx, full_range_flag);
static_assert(HP_ERR_DECODER_PLUGIN == heif_error_Decoder_plugin_error && HP_SUB_END_OF_DATA == heif_suberror_End_of_data, "codes");
static_assert(HP_COMPRESSION_HEVC == heif_compression_HEVC && HP_CHROMA_420 == heif_chroma_420 && HP_COLORSPACE_MONOCHROME == heif_colorspace_monochrome, "enums");
static_assert(HP_CHANNEL_CR == heif_channel_Cr && HP_PLUGIN_TYPE_DECODER == heif_plugin_type_decoder, "enums");
int main() { return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "abi.cc")
        open(f, "w").write(src)
        subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                               "-I" + os.path.join(root, "oracle", "_ref", "gen"), "-I" + os.path.join(reference_dir, "libheif", "api"), f])


# ---------------------------------------------------------------------------------------------
@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(wpp=0, stress=1), dict(tile_cols=2, tile_rows=2, wpp=0), dict(bit_depth=10)],
                         ids=["default", "stress", "tiles", "main10"])
def test_heif_decode_image_single_item_matches_oracle(cfg):
    lh.load_hip_plugin()
    w, h = 264, 200
    s = _still(w, h, seed=7, **cfg)
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, w, h)], bit_depth=cfg.get("bit_depth", 8)), lh.COLORSPACE_YCBCR, lh.CHROMA_420)
    assert out["bit_depth"] == cfg.get("bit_depth", 8)
    for c in range(3):
        np.testing.assert_array_equal(out["planes"][c], ref["planes"][c], err_msg="component %d" % c)


@needs_ref
@pytest.mark.gpu
def test_heif_decode_image_cropped_item_matches_oracle():
    lh.load_hip_plugin()
    s = _still(452, 458, seed=3)          # not a multiple of the minimum CB size: conformance-window crop inside the plugin
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, ref["width"], ref["height"])]), lh.COLORSPACE_YCBCR, lh.CHROMA_420)
    for c in range(3):
        np.testing.assert_array_equal(out["planes"][c], ref["planes"][c])


@needs_ref
@pytest.mark.gpu
def test_heif_decode_image_grid_matches_oracle_tiles():
    """libheif decodes the tiles on its own worker threads (grid.cc:405-453), one plugin decoder
    per tile, and pastes them: the canvas must equal the oracle tiles pasted the same way."""
    lh.load_hip_plugin()
    rows, cols, tw, th = 2, 3, 128, 128
    streams = [_still(tw, th, seed=20 + i, wpp=i % 2) for i in range(rows * cols)]
    ow, oh = cols * tw - 6, rows * th - 10
    out = lh.decode(hu.build_heic([(s, tw, th) for s in streams], grid=(rows, cols, ow, oh)), lh.COLORSPACE_YCBCR, lh.CHROMA_420, max_threads=4)
    canvas = [np.zeros((rows * th, cols * tw), np.uint8), np.zeros((rows * th // 2, cols * tw // 2), np.uint8), np.zeros((rows * th // 2, cols * tw // 2), np.uint8)]
    for i, s in enumerate(streams):
        ref = orc.decode(s)
        r, c = divmod(i, cols)
        for k in range(3):
            sub = 1 if k == 0 else 2
            canvas[k][r * th // sub:(r + 1) * th // sub, c * tw // sub:(c + 1) * tw // sub] = ref["planes"][k]
    np.testing.assert_array_equal(out["planes"][0], canvas[0][:oh, :ow])
    np.testing.assert_array_equal(out["planes"][1], canvas[1][:oh // 2, :ow // 2])
    np.testing.assert_array_equal(out["planes"][2], canvas[2][:oh // 2, :ow // 2])


@needs_ref
@pytest.mark.gpu
def test_heif_decode_image_to_rgb_matches_reference_colour_ops_on_oracle_planes():
    import ref_harness as rh
    lh.load_hip_plugin()
    w, h = 200, 136
    s = _still(w, h, seed=5, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    ref = orc.decode(s)
    out = lh.decode(hu.build_heic([(s, w, h)]), lh.COLORSPACE_RGB, lh.CHROMA_RGB)
    exp = rh.convert(ref["planes"], 8, rh.CH_420, ref["nclx"], rh.CS_RGB, rh.CH_RGB)[0]
    np.testing.assert_array_equal(out["rgb"], exp[:, :w * 3])


# ---- f2: alpha auxiliary image (libheif/image-items/image_item.cc:949-1081) -------------------------------------------------
def _with_alpha_fixture():
    """the reference's tests/data/with-alpha-512x512.heic, rebuilt from its two committed HEVC items (tests/golden/ref_with_alpha_*:
    item 1 colour 4:2:0, item 2 the monochrome alpha auxiliary image) — the GPU box has no /root/reference"""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    colour = open(os.path.join(gold, "ref_with_alpha_512x512_1.hevc"), "rb").read()
    alpha = open(os.path.join(gold, "ref_with_alpha_512x512_2.hevc"), "rb").read()
    return colour, alpha, hu.build_heic([(colour, 512, 512, 1), (alpha, 512, 512, 0)], alpha_of={2: 1})


@needs_ref
def test_alpha_auxiliary_heic_is_recognised_by_the_reference():
    """tests/component_descriptions.cc:362-366 asserts alpha as the 4th component of this file"""
    import ctypes as C
    _, _, heic = _with_alpha_fixture()
    L = lh.lib()
    ctx, h = lh.open_heic(heic)
    try:
        L.heif_image_handle_has_alpha_channel.argtypes = [C.c_void_p]
        assert L.heif_image_handle_has_alpha_channel(h) == 1
        assert (L.heif_image_handle_get_width(h), L.heif_image_handle_get_height(h)) == (512, 512)
    finally:
        L.heif_image_handle_release(h)
        L.heif_context_free(ctx)


@needs_ref
@pytest.mark.gpu
def test_with_alpha_fixture_decodes_to_rgba_with_the_real_alpha_plane():
    """two plugin decodes (colour item + monochrome alpha item, x265-coded), libheif attaches the alpha plane, RGBA out: colour
    equals the reference colour ops over the oracle's planes, the alpha channel equals the oracle's decode of the alpha item"""
    import ref_harness as rh
    lh.load_hip_plugin()
    colour, alpha, heic = _with_alpha_fixture()
    ref = orc.decode(colour)
    aref = orc.decode(alpha)
    assert aref["chroma_format_idc"] == 0
    out = lh.decode(heic, lh.COLORSPACE_RGB, lh.CHROMA_RGBA)["rgb"].reshape(512, 512, 4)
    exp = rh.convert(ref["planes"], 8, rh.CH_420, ref["nclx"], rh.CS_RGB, rh.CH_RGBA)[0][:, :512 * 4].reshape(512, 512, 4)
    np.testing.assert_array_equal(out[:, :, :3], exp[:, :, :3])
    np.testing.assert_array_equal(out[:, :, 3], aref["planes"][0])
    # and as planes: YCbCr + alpha
    yuv = lh.decode(heic, lh.COLORSPACE_YCBCR, lh.CHROMA_420)
    for c in range(3):
        np.testing.assert_array_equal(yuv["planes"][c], ref["planes"][c])


# ---- f1: one tile of a grid (libheif/api/libheif/heif_tiling.cc:99-133, image-items/grid.cc:580-603) -------------------------
@needs_ref
@pytest.mark.gpu
def test_heif_image_handle_decode_image_tile_decodes_only_that_tile():
    import ctypes as C
    from libheif_amd.decoder import coalesce_stats
    L = lh.load_hip_plugin()
    rows, cols, tw, th = 2, 3, 128, 128
    streams = [_still(tw, th, seed=60 + i) for i in range(rows * cols)]
    heic = hu.build_heic([(s, tw, th) for s in streams], grid=(rows, cols, cols * tw, rows * th))
    ctx, h = lh.open_heic(heic)
    L.heif_image_handle_decode_image_tile.restype = lh.HeifError
    L.heif_image_handle_decode_image_tile.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
    try:
        for (tx, ty) in ((0, 0), (2, 1), (1, 0)):
            before = coalesce_stats()[0]
            img = C.c_void_p()
            lh.check(L.heif_image_handle_decode_image_tile(h, C.byref(img), lh.COLORSPACE_YCBCR, lh.CHROMA_420, None, tx, ty))
            assert coalesce_stats()[0] - before == 1          # exactly one plugin decode: the requested tile
            ref = orc.decode(streams[ty * cols + tx])
            for c, ch in enumerate((lh.CHANNEL_Y, lh.CHANNEL_CB, lh.CHANNEL_CR)):
                np.testing.assert_array_equal(lh._plane(L, img, ch), ref["planes"][c])
            L.heif_image_release(img)
    finally:
        L.heif_image_handle_release(h)
        L.heif_context_free(ctx)
