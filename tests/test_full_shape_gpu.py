"""GPU parity at the REAL shapes of BASELINE.json's configs 3, 4 and 5 (config 1/2 — the 8-bit 4K still — is
tests/test_decode_gpu.py::test_full_size_stills_match_oracle):

  config 3: 8192x6144 grid = 8 x 6 tiles of 1024x1024, (a) through the batch API + GridDecoder's paste and (b) through the
            real libheif + the plugin (libheif's own tile fan-out, libheif/image-items/grid.cc:405-453, and paste :482-577);
  config 4: 3840x2160 Main10, VUI BT.2020 / PQ / limited range -> RRGGBB (Op_YCbCr420_to_RRGGBBaa, yuv2rgb.cc:622-734):
            first run of k_recon16 / k_sao<uint16_t> on a big picture;
  config 5: many DISTINCT 1920x1080 stills in ONE batch, enough substreams (>= 2048) that the CABAC work pool is used.

Inputs are the seeded synthetic streams of SURVEY.md §8(d) (S3, S4, S5); the expected planes come from the CPU oracle run
live (in worker processes: it is a scalar decoder), compared bit-exactly."""
import hashlib
import multiprocessing as mp
import os
import numpy as np
import pytest

from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def _oracle_hashes(stream):
    ref = orc.decode(stream)
    return [hashlib.sha256(np.ascontiguousarray(p).astype("<u2").tobytes()).hexdigest() for p in ref["planes"]], ref["n_substreams"]


def _oracle_planes(stream):
    return [np.ascontiguousarray(p) for p in orc.decode(stream)["planes"]]


def _pool_map(fn, items):
    workers = max(1, min(len(items), (os.cpu_count() or 2) - 1, 32))
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(fn, items)


def _hashes(planes):
    return [hashlib.sha256(np.ascontiguousarray(p).astype("<u2").tobytes()).hexdigest() for p in planes]


def test_config4_main10_4k_to_rrggbb():
    from libheif_amd.decoder import Batch
    from tools import streamgen
    stream = streamgen.make_stream(3840, 2160, 3, 10, vui_matrix=9, vui_primaries=9, vui_transfer=16)    # S4
    ref = orc.decode(stream)
    assert ref["bit_depth_luma"] == 10 and tuple(ref["nclx"]) == (9, 16, 9, 0)
    b = Batch([stream])
    b.run(); b.status()
    got = b.planes(0)
    for c in range(3):
        assert got[c].dtype == np.uint16
        np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="component %d" % c)
    for out_chroma, le in ((14, True), (12, False)):
        rgb = b.to_rgb(0, out_chroma)
        exp = orc.color_420_to_rrggbb(ref["planes"][0], ref["planes"][1], ref["planes"][2], 10, ref["nclx"], little_endian=le)
        np.testing.assert_array_equal(rgb, exp)
    b.free()


def test_config5_many_distinct_1080p_stills_in_one_pool_batch():
    from libheif_amd.decoder import Batch
    from tools import streamgen
    n = 128                                                        # 128 x 17 WPP rows = 2176 substreams >= 2048: pool mode
    streams = streamgen.make_streams([(1920, 1080, 1000 + i, 8, dict()) for i in range(n)])      # S5
    assert len({hashlib.sha256(s).hexdigest() for s in streams}) == n
    exp = _pool_map(_oracle_hashes, streams)
    b = Batch(streams)
    assert sum(b.info(i)["num_substreams"] for i in range(n)) >= 2048
    b.run(); b.status()
    b.alloc_rgb(10)
    b.to_rgb_all()
    for i in range(n):
        assert b.info(i)["num_substreams"] == exp[i][1]
        assert _hashes(b.planes(i)) == exp[i][0], "still %d" % i
    # the fused colour stage of the whole batch (one launch) on three of the stills
    for i in (0, 63, n - 1):
        ref = orc.decode(streams[i])
        nclx = tuple(ref["nclx"])
        if nclx[3] and (6 if nclx[2] == 2 else nclx[2]) not in (0, 8):    # planner rule: the integer op needs full range
            want = orc.color_420_to_rgb24(ref["planes"][0], ref["planes"][1], ref["planes"][2], nclx).reshape(1080, -1)
        else:                                                              # otherwise the float op + interleave
            r, g, bl = orc.color_ycbcr_to_rgb_planar(ref["planes"][0], ref["planes"][1], ref["planes"][2], 8, 1, nclx)
            want = orc.color_rgb_planar_to_interleaved8(r, g, bl).reshape(1080, -1)
        np.testing.assert_array_equal(b.rgb(i), want)
    b.free()


def _grid_streams():
    from tools import streamgen
    return streamgen.make_streams([(1024, 1024, 2 + t, 8, dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1))
                                   for t in range(48)])            # S3: seed = 2 + tile index, WPP (16 substreams per tile)


_CANVAS = {}


def _grid_canvas(streams, rows, cols, tw, th):
    key = hashlib.sha256(b"".join(streams)).hexdigest()
    if key not in _CANVAS:
        _CANVAS[key] = _build_canvas(streams, rows, cols, tw, th)
    return _CANVAS[key]


def _build_canvas(streams, rows, cols, tw, th):
    tiles = _pool_map(_oracle_planes, streams)
    canvas = [np.zeros((rows * th, cols * tw), np.uint8), np.zeros((rows * th // 2, cols * tw // 2), np.uint8),
              np.zeros((rows * th // 2, cols * tw // 2), np.uint8)]
    for t, planes in enumerate(tiles):
        r, c = divmod(t, cols)
        for k in range(3):
            s = 1 if k == 0 else 2
            canvas[k][r * th // s:(r + 1) * th // s, c * tw // s:(c + 1) * tw // s] = planes[k]
    return canvas


def test_config3_8k_grid_through_grid_decoder():
    torch = pytest.importorskip("torch")
    from libheif_amd.grid import GridDecoder, GridLayout
    rows, cols, tw, th = 6, 8, 1024, 1024
    streams = _grid_streams()
    canvas = _grid_canvas(streams, rows, cols, tw, th)
    layout = GridLayout(rows, cols, tw, th, cols * tw, rows * th)
    g = GridDecoder({t: s for t, s in enumerate(streams)}, layout)
    planes = g.decode()
    for k in range(3):
        np.testing.assert_array_equal(planes[k].cpu().numpy(), canvas[k], err_msg="component %d" % k)
    rgb = g.to_rgb((1, 13, 6, 1)).cpu().numpy()
    want = orc.color_420_to_rgb24(canvas[0], canvas[1], canvas[2], (1, 13, 6, 1)).reshape(rows * th, -1)
    np.testing.assert_array_equal(rgb, want)


@pytest.mark.parametrize("devices", [[0], [0] * 8], ids=["one_shard", "eight_shards"])
def test_config3_8k_grid_through_the_c_grid_api(devices):
    """the product path of config 3: hipdec_grid_* (C++, one process); eight shards = the t mod 8 partition of an 8-GPU node, here
    all on the one device of the GPU box"""
    from libheif_amd.grid import GridDecoderC, GridLayout
    rows, cols, tw, th = 6, 8, 1024, 1024
    streams = _grid_streams()
    canvas = _grid_canvas(streams, rows, cols, tw, th)
    g = GridDecoderC({t: s for t, s in enumerate(streams)}, GridLayout(rows, cols, tw, th, cols * tw, rows * th), devices)
    g.decode(); g.wait()
    planes = g.planes()
    for k in range(3):
        np.testing.assert_array_equal(planes[k], canvas[k], err_msg="component %d" % k)
    rgb = g.to_rgb(10)
    want = orc.color_420_to_rgb24(canvas[0], canvas[1], canvas[2], (1, 13, 6, 1)).reshape(rows * th, -1)
    np.testing.assert_array_equal(rgb, want)
    g.free()


def test_config3_8k_grid_through_libheif_and_the_plugin():
    import heic_util as hu
    import libheif_host as lh
    if not lh.available():
        pytest.fail("oracle/_ref/libheif.so is missing on the GPU box: build() must run before the snapshot is taken")
    lh.load_hip_plugin()
    rows, cols, tw, th = 6, 8, 1024, 1024
    streams = _grid_streams()
    canvas = _grid_canvas(streams, rows, cols, tw, th)
    heic = hu.build_heic([(s, tw, th) for s in streams], grid=(rows, cols, cols * tw, rows * th))
    out = lh.decode(heic, lh.COLORSPACE_YCBCR, lh.CHROMA_420, max_threads=48)
    for k in range(3):
        np.testing.assert_array_equal(out["planes"][k], canvas[k], err_msg="component %d" % k)


def test_config2_single_4k_still_fused_rgb():
    """BASELINE config 2 as written: ONE 3840x2160 8-bit 4:2:0 still on one GPU with the fused YCbCr -> RGB stage (decode + colour as one call:
    k_sao_rgb emits RGB24 from the SAO store path).  Planes against the oracle; RGB against the COMPILED reference op on the oracle's planes
    (Op_YCbCr420_to_RGB24, yuv2rgb.cc:345-426, through oracle/_ref's harness) and against the colour oracle; then the same file through the real
    libheif: heif_decode_image(..., RGB, interleaved_RGB) with the plugin."""
    from libheif_amd.decoder import Batch
    from tools import streamgen
    import libheif_host as lh
    w, h = 3840, 2160
    stream = streamgen.make_stream(w, h, 1, 8, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)     # S2, signalling the sRGB nclx
    ref = orc.decode(stream)
    assert tuple(ref["nclx"]) == (1, 13, 6, 1)
    b = Batch([stream])
    b.alloc_rgb(10)
    b.run_rgb()
    b.status()
    got = b.planes(0)
    for c in range(3):
        np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="component %d" % c)
    rgb = b.rgb(0)[:, :w * 3]
    y, cb, cr = ref["planes"]
    exp = orc.color_420_to_rgb24(y, cb, cr, ref["nclx"]).reshape(h, -1)
    np.testing.assert_array_equal(rgb, exp)
    b.free()
    if lh.available():
        import ref_harness as rh
        import heic_util as hu
        exp_ref = rh.convert(ref["planes"], 8, rh.CH_420, ref["nclx"], rh.CS_RGB, rh.CH_RGB)[0]
        np.testing.assert_array_equal(exp_ref[:, :w * 3], exp)                    # the colour oracle IS the reference op on these planes
        lh.load_hip_plugin()
        out = lh.decode(hu.build_heic([(stream, w, h)]), lh.COLORSPACE_RGB, lh.CHROMA_RGB)
        np.testing.assert_array_equal(out["rgb"][:, :w * 3], exp)
