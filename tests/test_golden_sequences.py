"""Committed golden SEQUENCE tracks (tests/golden/seq_*.hevcs + golden_sequences.json, written by tests/golden/make_golden_sequences.py): P and B
pictures with the tools of 8.5.3, long-term reference pictures, constrained intra prediction, scaling lists in inter pictures.  Everybody is held to the
same per-picture plane hashes: the CPU oracle (a regression pin for hevc_oracle_inter.c), the device code under the CPU emulation, and - on the GPU box,
where neither /root/reference nor the need for the generator exists - the HIP decoder through the decoder object, the way libheif drives it (one sample
per push, pictures polled in output order).  tests/test_reference_decoder_pin.py holds an independent HEVC decoder to them as soon as one is loadable."""
import hashlib
import json
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLD = os.path.join(HERE, "golden")
INDEX = json.load(open(os.path.join(GOLD, "golden_sequences.json")))


def load(name):
    """-> (access units in decoding order, the index entry)"""
    blob = open(os.path.join(GOLD, "seq_" + name + ".hevcs"), "rb").read()
    g = INDEX[name]
    assert hashlib.sha256(blob).hexdigest() == g["stream_sha256"], "fixture seq_%s.hevcs does not belong to golden_sequences.json" % name
    aus, p = [], 0
    while p < len(blob):
        n = int.from_bytes(blob[p:p + 4], "big")
        aus.append(blob[p + 4:p + 4 + n])
        p += 4 + n
    assert len(aus) == g["samples"]
    return aus, g


def hashes(planes):
    return [hashlib.sha256(np.asarray(p).astype("<u2").tobytes()).hexdigest() for p in planes]


def check_picture(name, g, poc, planes):
    want = g["pictures_sha256"][str(poc)]
    got = hashes(planes)
    assert got == want[:len(got)] and len(got) == len(want), "%s: the picture with PicOrderCnt %d differs from the golden planes" % (name, poc)


@pytest.mark.parametrize("name", sorted(INDEX))
def test_oracle_reproduces_golden_sequence(name):
    from oracle import pyoracle as orc
    aus, g = load(name)
    pics = orc.decode_sequence(aus)
    assert [p["poc"] for p in pics] == g["coding_order_pocs"]
    for p in pics:
        assert (p["width"], p["height"], p["bit_depth_luma"]) == (g["width"], g["height"], g["bit_depth"])
        check_picture(name, g, p["poc"], p["planes"])


@pytest.mark.parametrize("name", sorted(INDEX))
@pytest.mark.parametrize("chain", [0, 16], ids=["per_picture", "chain"])
def test_emulated_device_reproduces_golden_sequence(name, chain):
    """the kernel sources compiled for the host (tests/emu): one launch set per sample, and the look-ahead's chain form (one CABAC launch, one motion launch)"""
    from test_inter_emu import decode_sequence_emu
    aus, g = load(name)
    got = decode_sequence_emu(aus, chain=chain)
    assert len(got) == g["samples"]
    for poc, pic in zip(g["coding_order_pocs"], got):      # the emulation returns the pictures in decoding order
        check_picture(name, g, poc, pic["planes"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(INDEX))
def test_hip_decoder_reproduces_golden_sequence(name):
    """through the decoder object as libheif drives it: push a sample, take every picture that is ready, flush at the end - the pictures come out in
    OUTPUT order (PicOrderCnt 0, 1, 2, ...), each with the user_data of the sample that coded it"""
    from test_sequence_gpu import _play_track
    aus, g = load(name)
    got = _play_track(aus, None)
    assert len(got) == g["samples"]
    for out_idx, (img, user_data) in enumerate(got):
        assert user_data == 900 + g["coding_order_pocs"].index(out_idx), (name, out_idx, user_data)
        check_picture(name, g, out_idx, img.planes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(INDEX))
def test_hip_plugin_reproduces_golden_sequence_through_libheif(name):
    """the track as an image-sequence file (ISO/IEC 14496-12 movie boxes, tests/heic_util.py:build_sequence) through the REAL libheif: Track_Visual pushes
    the samples into the plugin and polls it (sequences/track_visual.cc:175-330); every image heif_track_decode_next_image() delivers, in output order,
    has the golden hashes"""
    import heic_util
    import libheif_host as lh
    if not lh.available():
        pytest.skip("oracle/_ref/libheif.so not built")
    lh.load_hip_plugin()
    aus, g = load(name)
    data = heic_util.build_sequence(aus, g["width"], g["height"], bit_depth=g["bit_depth"], chroma_format_idc=g["chroma_format_idc"])
    got = lh.decode_track(data)
    assert len(got) == g["samples"]
    for poc, img in enumerate(got):
        assert img["bit_depth"] == g["bit_depth"]
        check_picture(name + " through libheif", g, poc, img["planes"])
