"""Host front end (NAL framing, VPS/SPS/PPS, slice segment headers, substream table) through the C ABI's
hipdec_probe(): pure host code, so it runs without a GPU.  Mirrors what the reference checks before / around the
codec call: framing errors (decoder_libde265.cc:322-368), SPS geometry (hevc_boxes.cc:594-750, crop arithmetic
:688-716), security limits (decoder_libde265.cc:183-199)."""
import ctypes as C
import os
import numpy as np
import pytest

import libheif_amd
from libheif_amd._capi import ImageInfo
from oracle import pyoracle as orc


def probe(data, max_px=0):
    lib = libheif_amd.load_library()
    lib.hipdec_probe.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.POINTER(ImageInfo)]
    info = ImageInfo()
    rc = lib.hipdec_probe(data, len(data), max_px, C.byref(info))
    return rc, info, lib.hipdec_last_error().decode()


def _still(w, h, **cfg):
    return orc.encode(orc.synth_image(w, h, cfg.get("bit_depth", 8), cfg.pop("cf", 1), seed=9), **cfg)


def test_probe_matches_oracle_geometry_and_colour():
    for (w, h, cfg) in [(200, 136, dict()), (70, 42, dict()), (452, 458, dict(wpp=0)), (136, 72, dict(bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)),
                        (264, 136, dict(tile_cols=3, tile_rows=2, wpp=1)), (75, 41, dict(cf=0))]:
        s = _still(w, h, **dict(cfg))
        ref = orc.decode(s)
        rc, info, msg = probe(s)
        assert rc == 0, msg
        assert (info.width, info.height) == (ref["width"], ref["height"]) == (w, h)
        assert (info.colour_primaries, info.transfer_characteristics, info.matrix_coeffs, info.full_range_flag) == ref["nclx"]
        assert info.bit_depth_luma == ref["bit_depth_luma"] and info.chroma_format_idc == ref["chroma_format_idc"]
        assert info.num_substreams == ref["n_substreams"]
        assert info.coded_width % 8 == 0 and info.coded_width >= w and info.bitstream_bytes == len(s)


def test_framing_errors_are_end_of_data():
    s = _still(64, 64)
    assert probe(s[:len(s) - 3])[0] == -2          # NAL size exceeds the data
    assert probe(s + b"\x00\x00")[0] == -2         # truncated length field
    assert probe(b"")[0] == -7                     # nothing pushed
    # parameter sets only: no coded picture
    nals, p = [], 0
    while p < len(s):
        n = int.from_bytes(s[p:p + 4], "big"); nals.append(s[p:p + 4 + n]); p += 4 + n
    only_ps = b"".join(x for x in nals if (x[4] >> 1) & 63 >= 32)
    assert probe(only_ps)[0] == -7
    # a zero-length NAL and an unknown NAL type (AUD = 35) in between are ignored
    assert probe(nals[0] + b"\x00\x00\x00\x00" + b"\x00\x00\x00\x03\x46\x01\x50" + b"".join(nals[1:]))[0] == 0


def test_unsupported_tools_and_limits_are_loud():
    planes = orc.synth_image(64, 64, 8, 1, seed=2)
    assert probe(orc.encode(planes, pcm_pct=30))[0] == 0        # PCM coding units are part of the supported tool set
    for sl in (1, 2, 3):   # scaling lists (default / SPS / PPS) are part of the supported tool set
        assert probe(orc.encode(planes, scaling_list=sl))[0] == 0
    s = orc.encode(planes)
    assert probe(s, max_px=64 * 64 - 1)[0] == -5 and probe(s, max_px=64 * 64)[0] == 0
    # a second coded picture in the same item is outside the still-image path: rejected, not mis-decoded
    nals, p = [], 0
    while p < len(s):
        n = int.from_bytes(s[p:p + 4], "big"); nals.append(s[p:p + 4 + n]); p += 4 + n
    slices = [x for x in nals if (x[4] >> 1) & 63 < 32]
    assert probe(s + b"".join(slices))[0] == -4
    # slice data missing for part of the picture: incomplete picture
    two = orc.encode(orc.synth_image(136, 136, 8, 1, seed=3), num_slices=3, wpp=0)
    nals, p = [], 0
    while p < len(two):
        n = int.from_bytes(two[p:p + 4], "big"); nals.append(two[p:p + 4 + n]); p += 4 + n
    # (the header pass cannot know where a slice ends: the previous slice is taken to run to the end of the picture
    #  and the device-side parser then reports the premature end_of_slice_segment_flag — checked with the emulation)
    from test_parse_emu import run_emu
    status, _ = run_emu([b"".join(nals[:-1])])
    assert status != 0


def test_probe_reference_fixture_dimensions(reference_dir):
    """the dimensions the reference's own tests assert (tests/component_descriptions.cc:286-323: 451x461 displayed,
    ispe 452x462 coded window) and example.heic's 1280x854 items"""
    from heic_util import HeicFile
    f = HeicFile(os.path.join(reference_dir, "tests/data/rainbow-451x461.heic"))
    iid = f.hevc_items()[0]
    rc, info, msg = probe(f.plugin_stream(iid))
    assert rc == 0, msg
    assert (info.width, info.height) == tuple(f.ispe(iid))
    f = HeicFile(os.path.join(reference_dir, "examples/example.heic"))
    for iid in f.hevc_items():
        rc, info, msg = probe(f.plugin_stream(iid))
        assert rc == 0, msg
        assert (info.width, info.height) == tuple(f.ispe(iid))


def test_dependent_slice_segments_are_parsed_and_broken_ones_are_refused():
    """host side of dependent_slice_segment_flag (hevc_headers.hip): the segments of a slice probe as one picture; a missing first or middle
    segment is an incomplete / inconsistent picture, never a crash"""
    planes = orc.synth_image(200, 136, 8, 1, seed=4)
    s = orc.encode(planes, dependent_segments=3, wpp=0)
    assert probe(s)[0] == 0
    nals, p = [], 0
    while p < len(s):
        n = int.from_bytes(s[p:p + 4], "big"); nals.append(s[p:p + 4 + n]); p += 4 + n
    ps = [x for x in nals if (x[4] >> 1) & 63 >= 32]
    sl = [x for x in nals if (x[4] >> 1) & 63 < 32]
    assert len(sl) == 3
    assert probe(b"".join(ps + sl[1:]))[0] in (-2, -3, -4, -6, -7)            # no first slice segment
    # the middle segment gone: the headers alone look like a longer first segment (the parse kernel then finds end_of_slice_segment_flag
    # too early and reports it: tests/test_parse_emu.py::test_missing_dependent_segment_is_a_device_error)
    assert probe(b"".join(ps + [sl[0], sl[2]]))[0] in (0, -2, -3, -4, -6)
    assert probe(b"".join(ps + [sl[0], sl[1]]))[0] in (0, -2, -3, -4, -6)     # (likewise: the second segment looks longer; the device reports the early end)
    assert probe(orc.encode(planes, dependent_segments=3, wpp=1))[0] == 0     # under WPP the segments start at CTB row starts
