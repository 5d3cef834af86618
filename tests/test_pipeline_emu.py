"""CPU check of the WHOLE device pipeline's logic: the kernel sources of libheif_amd/csrc (parse_core.h,
residual_kernel.hip, recon_kernel.hip, filter_kernels.hip) compiled for the host against the SIMT emulator of
tests/emu/shim and run end to end (CABAC parse -> dequant + inverse transforms -> intra reconstruction wavefront ->
deblocking -> SAO -> crop), decoded planes bit-exact against the oracle over the coding-tool matrix.  The emulation is
test infrastructure; the product runs the same sources on the GPU (tests/test_decode_gpu.py)."""
import ctypes as C
import numpy as np
import pytest

from oracle import pyoracle as orc
import test_parse_emu as tpe


def _lib():
    L = tpe.emu()
    L.emu_run_pipeline.argtypes = [C.c_void_p, C.c_int]
    L.emu_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.emu_out_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return L


def decode_emu(streams):
    """[planes per item] decoded by the emulated kernels; raises on a device status != 0"""
    L = _lib()
    n = len(streams)
    arr = (C.c_char_p * n)(*streams)
    sizes = (C.c_size_t * n)(*[len(s) for s in streams])
    err = C.create_string_buffer(512)
    h = L.emu_create(n, arr, sizes, err, 512)
    assert h, err.value.decode()
    try:
        st = L.emu_run_parse(h)
        assert st == 0, "parse status 0x%x" % (st & 0xffffffff)
        st = L.emu_run_pipeline(h, 15)
        assert st == 0, "pipeline status 0x%x" % (st & 0xffffffff)
        out = []
        for i in range(n):
            sz = (C.c_int * 5)()
            L.emu_out_size(h, i, sz)
            w, hh, cw, ch, es = list(sz)
            dt = np.uint16 if es == 2 else np.uint8
            planes = [np.zeros((hh, w), dt)]
            L.emu_plane(h, i, 0, planes[0].ctypes.data)
            for c in (1, 2):
                if cw and ch:
                    p = np.zeros((ch, cw), dt)
                    L.emu_plane(h, i, c, p.ctypes.data)
                    planes.append(p)
            out.append(planes)
        return out
    finally:
        L.emu_free(h)


def _check(stream, planes):
    ref = orc.decode(stream)
    assert len(planes) == len(ref["planes"])
    for c in range(len(ref["planes"])):
        np.testing.assert_array_equal(planes[c], ref["planes"][c], err_msg="component %d" % c)


CONFIGS = [
    dict(),
    dict(stress=1),
    dict(wpp=0, stress=1),
    dict(num_slices=3, loop_filter_across_slices=0),
    dict(transform_skip=1, stress=1),
    dict(lossless_pct=30),
    dict(bit_depth=10),
    dict(log2_ctb=5, log2_min_cb=4, log2_max_tb=5, max_transform_hierarchy_depth_intra=2, stress=1),
    dict(log2_ctb=4, log2_min_cb=3, log2_max_tb=4, stress=1),
    dict(sao=0, deblock_disable=1),
    dict(cb_qp_offset=3, cr_qp_offset=-4, beta_offset_div2=2, tc_offset_div2=-2, qp=34),
    dict(qp=12, stress=1, zero_residual_pct=30),
    dict(sign_data_hiding=0, cu_qp_delta=0, strong_intra_smoothing=0),
    dict(qp=40),
    dict(pcm_pct=25),                                                     # pcm_sample blocks, loop filters across them
    dict(pcm_pct=30, pcm_loop_filter_disabled=1, stress=1),               # ... and left untouched by deblocking / SAO
    dict(pcm_pct=25, bit_depth=10, lossless_pct=20),                      # PcmBitDepth below BitDepth (shifted samples), beside lossless CUs
    dict(pcm_pct=40, log2_ctb=5, log2_max_tb=4, wpp=0),                   # 32x32 PCM units above the maximum transform size
    dict(dependent_segments=3, wpp=0, stress=1),                          # dependent slice segments: no slice boundary for prediction / filters inside the slice
    dict(dependent_segments=2, num_slices=2, wpp=0, loop_filter_across_slices=0, log2_ctb=4, log2_max_tb=4),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: ",".join("%s=%s" % kv for kv in c.items()) or "default")
def test_emulated_pipeline_matches_oracle(cfg):
    planes = orc.synth_image(200, 136, cfg.get("bit_depth", 8), 1, seed=21)
    stream = orc.encode(planes, **cfg)
    _check(stream, decode_emu([stream])[0])


def test_emulated_pipeline_monochrome_and_cropped_sizes():
    for w, h, cf in [(75, 41, 0), (70, 42, 1), (8, 8, 1), (136, 24, 1)]:
        stream = orc.encode(orc.synth_image(w, h, 8, cf, seed=5))
        _check(stream, decode_emu([stream])[0])


def test_emulated_pipeline_batch_of_mixed_sizes():
    """several items in one batch: one set of (emulated) launches, every item exactly its own picture"""
    streams = []
    for i, (w, h) in enumerate([(128, 64), (64, 128), (200, 136), (72, 40)]):
        streams.append(orc.encode(orc.synth_image(w, h, 8, 1, seed=40 + i), qp=24 + 3 * i, stress=i & 1))
    for s, planes in zip(streams, decode_emu(streams)):
        _check(s, planes)


def test_emulated_pipeline_strong_smoothing_32x32():
    """smooth content at a coarse QP: 32x32 transform blocks with strong intra smoothing"""
    yy, xx = np.mgrid[0:192, 0:256]
    y = (40 + 0.5 * xx + 0.25 * yy).astype(np.uint8)
    planes = [y, np.full((96, 128), 120, np.uint8), np.full((96, 128), 135, np.uint8)]
    stream = orc.encode(planes, qp=38)
    _check(stream, decode_emu([stream])[0])


def test_emulated_pipeline_random_tool_mixes():
    """a fixed-seed sweep over CTB / CB / TB sizes, slices, WPP, transform skip, lossless CUs, SAO, bit depth and picture
    sizes that are not CTB multiples (the development fuzz loop of the kernels, shortened)"""
    import random
    rng = random.Random(20260922)
    done = 0
    while done < 14:
        lc = rng.choice([4, 5, 6])
        cfg = dict(log2_ctb=lc, log2_min_cb=rng.choice([3, min(4, lc)]), qp=rng.choice([10, 18, 26, 32, 38, 44]), stress=rng.choice([0, 1]),
                   wpp=rng.choice([0, 1]), num_slices=rng.choice([1, 1, 2, 4]), transform_skip=rng.choice([0, 1]),
                   strong_intra_smoothing=rng.choice([0, 1]), sao=rng.choice([0, 1]), lossless_pct=rng.choice([0, 0, 15]))
        cfg["log2_max_tb"] = max(cfg["log2_min_cb"], rng.choice([t for t in (3, 4, 5) if t <= lc]))
        bd = rng.choice([8, 8, 10])
        if bd == 10:
            cfg["bit_depth"] = 10
        w, h = rng.choice([8, 16, 40, 72, 136, 200, 264]), rng.choice([8, 24, 40, 72, 136])
        try:
            stream = orc.encode(orc.synth_image(w, h, bd, 1, seed=done + 7), **cfg)
        except orc.OracleError:
            continue      # a parameter mix the test encoder refuses
        _check(stream, decode_emu([stream])[0])
        done += 1


def test_emulated_pipeline_decodes_reference_fixtures(reference_dir):
    """the reference's real x265-coded fixtures through the emulated device pipeline: planes identical to the oracle's"""
    import os
    from heic_util import HeicFile
    for rel, items in (("tests/data/rainbow-451x461.heic", None), ("examples/example.heic", "thumbnails")):
        h = HeicFile(os.path.join(reference_dir, rel))
        ids = h.hevc_items()
        if items == "thumbnails":
            ids = [k[1] for k in h.refs if k[0] == "thmb"]      # 320x212 each (the 1280x854 masters run on the GPU suite)
        for iid in ids:
            stream = h.plugin_stream(iid)
            _check(stream, decode_emu([stream])[0])


@pytest.mark.parametrize("yield_ctbs", [1, 0])
def test_emulated_pipeline_after_pool_scheduled_parse(yield_ctbs, monkeypatch):
    """throughput mode end to end: the parser runs as the work pool (rows as tasks, a row hands its wave back after every
    CTB - the product default - or runs until blocked), then the rest of the pipeline; several pictures in one batch"""
    monkeypatch.setenv("HIPDEC_PARSE_POOL", "1")
    monkeypatch.setenv("HIPDEC_POOL_YIELD", str(yield_ctbs))
    streams = [orc.encode(orc.synth_image(w, h, 8, 1, seed=70 + i), stress=i & 1, qp=24 + 4 * i)
               for i, (w, h) in enumerate([(264, 200), (200, 136), (328, 72), (72, 264)])]
    for s, planes in zip(streams, decode_emu(streams)):
        _check(s, planes)


@pytest.mark.parametrize("scaling_list", [1, 2, 3], ids=["default_lists", "sps_lists", "pps_lists"])
@pytest.mark.parametrize("cfg", [dict(), dict(stress=1, transform_skip=1), dict(bit_depth=10), dict(log2_ctb=4, log2_min_cb=3, log2_max_tb=4, stress=1),
                                 dict(qp=44), dict(qp=10, stress=1), dict(log2_ctb=5, log2_max_tb=5, qp=38), dict(lossless_pct=20, num_slices=2)],
                         ids=["default", "tskip", "main10", "ctb16", "qp44", "qp10", "tb32", "lossless_slices"])
def test_emulated_pipeline_with_scaling_lists(cfg, scaling_list):
    """scaling_list_enabled_flag (8.6.4.2): the default lists of Table 7-6, explicit lists in the SPS, explicit lists in
    the PPS (copies of the default / of earlier matrices, DPCM-coded lists with DC coefficients)"""
    c = dict(cfg, scaling_list=scaling_list)
    stream = orc.encode(orc.synth_image(200, 136, c.get("bit_depth", 8), 1, seed=33 + scaling_list), **c)
    _check(stream, decode_emu([stream])[0])


def test_emulated_pipeline_scaling_lists_32x32():
    yy, xx = np.mgrid[0:192, 0:256]
    planes = [(40 + 0.5 * xx + 0.25 * yy).astype(np.uint8), np.full((96, 128), 120, np.uint8), np.full((96, 128), 135, np.uint8)]
    for sl in (1, 2, 3):
        stream = orc.encode(planes, qp=34, scaling_list=sl)
        ref = orc.decode(stream, taps=True)
        assert (ref["map_log2_tb"] == 5).any(), "the stream was meant to carry 32x32 transform blocks"
        _check(stream, decode_emu([stream])[0])


@pytest.mark.parametrize("cfg", [dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1), dict(),
                                 dict(vui_primaries=1, vui_transfer=1, vui_matrix=1, vui_full_range=0, stress=1),
                                 dict(num_slices=3, loop_filter_across_slices=0, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1),
                                 dict(lossless_pct=30, tile_cols=2, tile_rows=2, wpp=0, loop_filter_across_tiles=0)],
                         ids=["srgb_int_op", "unspecified", "bt709_limited_float", "slices", "lossless_tiles"])
@pytest.mark.parametrize("size", [(200, 136), (70, 42), (452, 264)])
def test_sao_with_fused_rgb_emission_matches_planes_and_colour_oracle(cfg, size):
    """k_sao_rgb (SAO + crop of the three components of a tile, RGB24 emitted from registers + LDS): planes as the plain SAO kernel's, RGB as
    the colour oracle over those planes with the planner's choice of op (integer for full range, float chain otherwise)"""
    L = _lib()
    L.emu_run_pipeline_rgb.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    w, h = size
    streams = [orc.encode(orc.synth_image(w, h, 8, 1, seed=5 + k), **cfg) for k in range(2)]
    arr = (C.c_char_p * 2)(*streams)
    sizes = (C.c_size_t * 2)(*[len(s) for s in streams])
    err = C.create_string_buffer(512)
    hnd = L.emu_create(2, arr, sizes, err, 512)
    assert hnd, err.value.decode()
    try:
        assert L.emu_run_parse(hnd) == 0
        rgb = np.zeros((2, h, w * 3), np.uint8)
        assert L.emu_run_pipeline_rgb(hnd, 1 | 2 | 4 | 16, rgb.ctypes.data) == 0
        for i, s in enumerate(streams):
            ref = orc.decode(s)
            for c in range(3):
                p = np.zeros(ref["planes"][c].shape, np.uint8)
                L.emu_plane(hnd, i, c, p.ctypes.data)
                np.testing.assert_array_equal(p, ref["planes"][c], err_msg="item %d component %d" % (i, c))
            nclx = tuple(ref["nclx"])
            if nclx[3] and (6 if nclx[2] == 2 else nclx[2]) not in (0, 8):
                want = orc.color_420_to_rgb24(ref["planes"][0], ref["planes"][1], ref["planes"][2], nclx).reshape(h, -1)
            else:
                r, g, b = orc.color_ycbcr_to_rgb_planar(ref["planes"][0], ref["planes"][1], ref["planes"][2], 8, 1, nclx)
                want = orc.color_rgb_planar_to_interleaved8(r, g, b).reshape(h, -1)
            np.testing.assert_array_equal(rgb[i], want, err_msg="item %d RGB" % i)
    finally:
        L.emu_free(hnd)
