"""4:2:2 and 4:4:4 (chroma_format_idc 2 / 3) on the CPU: the oracle's own consistency (encode -> decode round trips of the test encoder,
lossless coding exact), the host front end's acceptance rules, and the WHOLE emulated device pipeline (the kernel sources of
libheif_amd/csrc compiled for the host: CABAC parse -> residual -> reconstruction wavefront -> deblocking -> SAO -> crop) bit-exact against
the oracle.

What changes with ChromaArrayType 3 (ITU-T H.265 v2+): chroma transform blocks have the luma blocks' size and position (down to 4x4,
up to 32x32), every transform-tree node carries cbf_cb / cbf_cr (a fifth context at trafoDepth 4), an NxN coding unit has one
intra_chroma_pred_mode per partition, chroma reference samples are smoothed like luma ones (but take no boundary filters), 8x8 chroma
blocks use the mode-dependent scans, QpC = Min(qPi, 51) instead of table 8-10, and the chroma planes are deblocked on the luma edge grid.

With ChromaArrayType 2 the chroma of a transform unit is half as wide and as tall as the luma block: TWO square blocks one above the other,
each with its own cbf_cb / cbf_cr and transform_skip_flag (coded Cb upper, Cb lower, Cr upper, Cr lower; the lower block predicts from the
upper one), the chroma prediction direction goes through Table 8-3, QpC = Min(qPi, 51), the 8x8 chroma deblocking grid is 16 luma samples
wide and 8 tall, PCM / SAO / conformance-window geometry is subsampled horizontally only."""
import random

import numpy as np
import pytest

from oracle import pyoracle as orc
import test_pipeline_emu as tpe


FORMATS = pytest.mark.parametrize("cf", [2, 3], ids=["422", "444"])


def _roundtrip(w, h, bd=8, seed=1, cf=3, **cfg):
    planes = orc.synth_image(w, h, bd, cf, seed=seed)
    stream = orc.encode(planes, bit_depth=bd, **cfg)
    return planes, stream, orc.decode(stream)


@FORMATS
def test_oracle_geometry_and_headers(cf):
    w, h = (75, 41) if cf == 3 else (74, 41)                  # odd sizes are fine where the chroma is not subsampled in that direction
    planes, stream, ref = _roundtrip(w, h, seed=3, cf=cf)
    assert ref["chroma_format_idc"] == cf
    assert [p.shape for p in ref["planes"]] == [(h, w)] + [(h, w if cf == 3 else w // 2)] * 2
    for a, b in zip(planes, ref["planes"]):
        assert np.abs(a.astype(int) - b.astype(int)).mean() < 6   # a lossy but sane reconstruction of every plane


@FORMATS
@pytest.mark.parametrize("bd", [8, 10])
def test_oracle_lossless_is_exact(bd, cf):
    planes, stream, ref = _roundtrip(136, 72, bd=bd, seed=5, cf=cf, lossless_pct=100, stress=1, max_transform_hierarchy_depth_intra=2)
    for a, b in zip(planes, ref["planes"]):
        np.testing.assert_array_equal(a, b)


def test_front_end_accepts_422_and_444():
    from test_frontend_cpu import probe
    rc, info, msg = probe(orc.encode(orc.synth_image(72, 40, 8, 3, seed=2)))
    assert rc == 0, msg
    assert (info.chroma_format_idc, info.chroma_width, info.chroma_height) == (3, 72, 40)
    rc, info, msg = probe(orc.encode(orc.synth_image(70, 41, 8, 2, seed=2)))       # cropped: conformance window in chroma units, x only
    assert rc == 0, msg
    assert (info.chroma_format_idc, info.width, info.height, info.chroma_width, info.chroma_height) == (2, 70, 41, 35, 41)


CONFIGS = [
    dict(),
    dict(stress=1),
    dict(wpp=0, stress=1, transform_skip=1),
    dict(bit_depth=10, stress=1),
    dict(log2_ctb=6, log2_min_cb=3, log2_min_tb=2, log2_max_tb=5, max_transform_hierarchy_depth_intra=4, stress=1),   # cbf_cb / cbf_cr at trafoDepth 4
    dict(log2_ctb=4, log2_min_cb=3, log2_max_tb=4, stress=1),
    dict(log2_ctb=5, log2_min_cb=4, log2_min_tb=3, log2_max_tb=5, max_transform_hierarchy_depth_intra=2),
    dict(cb_qp_offset=9, cr_qp_offset=-10, qp=44, tc_offset_div2=2),       # qPi beyond 43: Min(qPi, 51), not table 8-10
    dict(qp=10, stress=1, zero_residual_pct=30),
    dict(lossless_pct=30, pcm_pct=20),
    dict(pcm_pct=30, pcm_loop_filter_disabled=1, bit_depth=10),
    dict(num_slices=3, loop_filter_across_slices=0, stress=1),
    dict(tile_cols=2, tile_rows=2, wpp=0, loop_filter_across_tiles=0),
    dict(dependent_segments=3, wpp=0),
    dict(sao=0, deblock_disable=1, sign_data_hiding=0, cu_qp_delta=0, strong_intra_smoothing=0),
    dict(qp=38),                                                            # large smooth blocks: 32x32 chroma transforms
]


SCALING = [
    dict(scaling_list=1, stress=1),                                         # default lists (4:4:4: 32x32 chroma matrices = the 16x16 lists upsampled)
    dict(scaling_list=3, bit_depth=10, transform_skip=1),                   # lists in the PPS
    dict(scaling_list=2, qp=38),                                            # lists in the SPS, large blocks
]
CONFIGS_422 = [c for c in CONFIGS if c.get("max_transform_hierarchy_depth_intra") != 4] + SCALING + [
    dict(log2_ctb=6, log2_min_cb=3, log2_min_tb=2, log2_max_tb=5, max_transform_hierarchy_depth_intra=3, stress=1),
]
CONFIGS = CONFIGS + SCALING


def _configs(cf):
    return CONFIGS if cf == 3 else CONFIGS_422


@pytest.mark.parametrize("cf,cfg", [(cf, c) for cf in (2, 3) for c in _configs(cf)],
                         ids=lambda v: ("4%d%d" % ((2, 2) if v == 2 else (4, 4))) if isinstance(v, int) else (",".join("%s=%s" % kv for kv in v.items()) or "default"))
def test_emulated_pipeline_matches_oracle(cf, cfg):
    bd = cfg.get("bit_depth", 8)
    stream = orc.encode(orc.synth_image(200, 136, bd, cf, seed=31), **cfg)
    planes = tpe.decode_emu([stream])[0]
    assert len(planes) == 3 and planes[1].shape == ((136, 200) if cf == 3 else (136, 100))
    tpe._check(stream, planes)


@pytest.mark.parametrize("cfg", [dict(), dict(stress=1, transform_skip=1, wpp=0), dict(pcm_pct=30, lossless_pct=20), dict(bit_depth=10, log2_ctb=5, log2_max_tb=5, stress=1)],
                         ids=["default", "stress", "pcm", "main10"])
@FORMATS
def test_parser_emulation_maps_and_coefficients(cf, cfg):
    """the parse kernel alone: per-unit maps, coefficient levels at their positions, SAO parameters against the oracle's taps"""
    import test_parse_emu as tp
    stream = orc.encode(orc.synth_image(200, 136, cfg.get("bit_depth", 8), cf, seed=9), **cfg)
    status, out = tp.run_emu([stream])
    assert status == 0
    tp.check_against_oracle(stream, out[0])


def test_emulated_pipeline_batch_mixing_chroma_formats():
    """4:4:4, 4:2:2, 4:2:0 and 4:0:0 pictures in ONE batch: the reconstruction wave table holds three, two or one wave chains per row chain"""
    streams = []
    for i, (w, h, cf) in enumerate([(128, 64, 3), (64, 128, 1), (200, 136, 2), (72, 40, 0), (136, 72, 1), (75, 41, 3), (70, 41, 2)]):
        streams.append(orc.encode(orc.synth_image(w, h, 8, cf, seed=40 + i), qp=24 + 3 * i, stress=i & 1, tile_cols=1 + (i % 2)))
    for s, planes in zip(streams, tpe.decode_emu(streams)):
        tpe._check(s, planes)


@FORMATS
def test_emulated_pipeline_random_tool_mixes(cf):
    rng = random.Random(440 + cf)
    done = 0
    while done < 12:
        lc = rng.choice([4, 5, 6])
        cfg = dict(log2_ctb=lc, log2_min_cb=rng.choice([3, min(4, lc)]), qp=rng.choice([10, 18, 26, 32, 38, 44]), stress=rng.choice([0, 1]),
                   wpp=rng.choice([0, 1]), num_slices=rng.choice([1, 1, 2, 4]), transform_skip=rng.choice([0, 1]),
                   strong_intra_smoothing=rng.choice([0, 1]), sao=rng.choice([0, 1]), lossless_pct=rng.choice([0, 0, 15]), pcm_pct=rng.choice([0, 0, 20]),
                   max_transform_hierarchy_depth_intra=rng.choice([0, 1, 2, 3]), cb_qp_offset=rng.choice([0, -7, 9]), cr_qp_offset=rng.choice([0, 5, -10]),
                   tile_cols=rng.choice([1, 1, 2]), dependent_segments=rng.choice([0, 0, 2]), scaling_list=rng.choice([0, 0, 2]))
        cfg["log2_max_tb"] = rng.choice([t for t in (3, 4, 5) if t <= lc])
        cfg["log2_min_tb"] = rng.choice([t for t in (2, 3) if t < cfg["log2_min_cb"] and t <= cfg["log2_max_tb"]])
        cfg["max_transform_hierarchy_depth_intra"] = min(cfg["max_transform_hierarchy_depth_intra"], lc - cfg["log2_min_tb"])
        bd = rng.choice([8, 8, 10, 12])
        w, h = rng.choice([8, 16, 40, 72, 136, 200]), rng.choice([8, 24, 40, 41, 72, 136])
        try:
            stream = orc.encode(orc.synth_image(w, h, bd, cf, seed=done + 7), bit_depth=bd, **cfg)
        except orc.OracleError:
            continue      # a parameter mix the test encoder refuses
        tpe._check(stream, tpe.decode_emu([stream])[0])
        done += 1
