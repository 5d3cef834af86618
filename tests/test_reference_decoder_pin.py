"""The PIN for HEVC pixels: every committed golden stream decoded through the REAL reference
libheif by an HEVC decoder plugin that is NOT ours, compared with the plane hashes of tests/golden/golden.json — the hashes the oracle
and the HIP decoder are both held to.

The contract is libheif/plugins/decoder_libde265.cc:402 (de265_decode).  libde265 is an un-vendored dependency of the reference and is
absent from this image, so today `test_independent_hevc_decoder_reproduces_golden` SKIPS with "parity unpinned".  It activates by itself
the moment an independent decoder is loadable:
  * oracle/Makefile.ref `pin` builds libheif's own libde265 plugin into oracle/_ref/plugins/ when pkg-config sees libde265 (build() tries it);
  * any *.so in $HIPDEC_PIN_PLUGIN_PATH / $LIBHEIF_PLUGIN_PATH (ffmpeg plugin, ...) that registers an HEVC decoder is taken as well.
That the harness itself works — finds a second decoder, selects it by decoder_id (plugin_registry.cc:264-288), wraps each stream as a HEIC,
decodes through heif_decode_image() and compares — is proven on every run by the self-test plugin (the ORACLE behind the plugin ABI, id
"oraclepin": oracle/pin_selftest_plugin.c).  That self-test is not a pin and is never counted as one."""
import ctypes as C
import glob
import hashlib
import json
import os
import numpy as np
import pytest

import heic_util
import libheif_host as host
import ref_harness

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
INDEX = json.load(open(os.path.join(GOLD, "golden.json")))
REF = os.path.join(HERE, "..", "oracle", "_ref")
NOT_INDEPENDENT = ("hipdec", "oraclepin")   # ours, and the oracle-behind-the-ABI self-test

pytestmark = pytest.mark.skipif(not host.available(), reason="oracle/_ref/libheif.so not built (needs /root/reference)")


def _load_plugins(dirs):
    L = host.lib()
    for d in dirs:
        for so in sorted(glob.glob(os.path.join(d, "*.so"))):
            info = C.c_void_p()
            L.heif_load_plugin(so.encode(), C.byref(info))      # a file that is no libheif plugin just fails to load


def _hevc_decoder_ids():
    L = host.lib()
    L.heif_get_decoder_descriptors.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int]
    L.heif_decoder_descriptor_get_id_name.restype = C.c_char_p
    L.heif_decoder_descriptor_get_id_name.argtypes = [C.c_void_p]
    L.heif_decoder_descriptor_get_name.restype = C.c_char_p
    L.heif_decoder_descriptor_get_name.argtypes = [C.c_void_p]
    n = L.heif_get_decoder_descriptors(host.COMPRESSION_HEVC, None, 0)
    arr = (C.c_void_p * max(1, n))()
    n = L.heif_get_decoder_descriptors(host.COMPRESSION_HEVC, arr, n)
    return [((L.heif_decoder_descriptor_get_id_name(arr[i]) or b"").decode(), (L.heif_decoder_descriptor_get_name(arr[i]) or b"").decode()) for i in range(n)]


def _independent_decoder():
    dirs = [os.path.join(REF, "plugins")]
    for var in ("HIPDEC_PIN_PLUGIN_PATH", "LIBHEIF_PLUGIN_PATH"):
        dirs += [d for d in os.environ.get(var, "").split(":") if d]
    _load_plugins(dirs)
    ids = [(i, n) for i, n in _hevc_decoder_ids() if i not in NOT_INDEPENDENT]
    return ids[0] if ids else None


def _decode_with(decoder_id, heic):
    """heif_decode_image() with heif_decoding_options.decoder_id pinned: YCbCr planes exactly as the plugin handed them over"""
    L = host.lib()
    ctx, h = host.open_heic(heic)
    img = C.c_void_p()
    H = ref_harness.lib()          # the options are filled in by C code compiled against the real header (oracle/ref_harness.cc)
    H.refh_decoding_options.restype = C.c_void_p
    H.refh_decoding_options.argtypes = [C.c_char_p]
    keep = C.c_char_p(decoder_id.encode())
    opts = C.c_void_p(H.refh_decoding_options(keep))
    try:
        host.check(L.heif_decode_image(h, C.byref(img), host.COLORSPACE_UNDEFINED, host.CHROMA_UNDEFINED, opts))
        bpp = L.heif_image_get_bits_per_pixel_range(img, host.CHANNEL_Y)
        bs = 2 if bpp > 8 else 1
        return [p for p in (host._plane(L, img, c, bs) for c in (host.CHANNEL_Y, host.CHANNEL_CB, host.CHANNEL_CR)) if p is not None]
    finally:
        L.heif_decoding_options_free(opts)
        if img:
            L.heif_image_release(img)
        L.heif_image_handle_release(h)
        L.heif_context_free(ctx)


def _golden():
    out = []
    for table in ("streams", "reference_streams"):
        for name in sorted(INDEX.get(table, {})):
            out.append((table, name))
    return out


def _check(decoder_id, table, name):
    g = INDEX[table][name]
    data = open(os.path.join(GOLD, name + ".hevc"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == g["stream_sha256"]
    cf = g.get("chroma_format_idc")
    if cf is None:
        from oracle import pyoracle as orc
        cf = orc.decode(data)["chroma_format_idc"]
    heic = heic_util.build_heic([(data, g["width"], g["height"], cf)], bit_depth=g["bit_depth"], chroma_format_idc=cf)
    planes = _decode_with(decoder_id, heic)
    got = [hashlib.sha256(np.asarray(p).astype("<u2").tobytes()).hexdigest() for p in planes]
    assert got == g["planes_sha256"][:len(got)], "%s decodes %s/%s differently from the golden planes" % (decoder_id, table, name)


# ---- sequence tracks (inter prediction): tests/golden/seq_*.hevcs as image-sequence files through heif_track_decode_next_image() ------------------------
SEQ_INDEX = json.load(open(os.path.join(GOLD, "golden_sequences.json")))


def _check_sequence(decoder_id, name):
    """the committed track wrapped as an ISO/IEC 14496-12 image sequence (tests/heic_util.py:build_sequence), decoded by libheif's Track_Visual
    (sequences/track_visual.cc:175-330) with the decoder pinned by id: every image, in the order libheif delivers them (= output order), has the
    golden hashes of the picture with that PicOrderCnt"""
    import test_golden_sequences as gs
    aus, g = gs.load(name)
    data = heic_util.build_sequence(aus, g["width"], g["height"], bit_depth=g["bit_depth"], chroma_format_idc=g["chroma_format_idc"])
    H = ref_harness.lib()
    H.refh_decoding_options.restype = C.c_void_p
    H.refh_decoding_options.argtypes = [C.c_char_p]
    keep = C.c_char_p(decoder_id.encode())
    opts = C.c_void_p(H.refh_decoding_options(keep))
    try:
        got = host.decode_track(data, options=opts)
    finally:
        host.lib().heif_decoding_options_free(opts)
    assert len(got) == g["samples"], "%s: %d of %d images came out of the track" % (name, len(got), g["samples"])
    for poc, img in enumerate(got):
        assert img["bit_depth"] == g["bit_depth"]
        gs.check_picture("%s through %s" % (name, decoder_id), g, poc, img["planes"])


def test_harness_selftest_decodes_golden_sequences_through_a_second_decoder():
    """the oracle behind the plugin ABI again (NOT a pin): image-sequence file -> Track_Visual -> the decoder selected by id -> output order -> hashes;
    proves the sequence harness below before an independent decoder shows up"""
    sd = os.path.join(REF, "plugins_selftest")
    if not glob.glob(os.path.join(sd, "*.so")):
        pytest.skip("self-test plugin not built")
    _load_plugins([sd])
    assert "oraclepin" in [i for i, _ in _hevc_decoder_ids()]
    for name in sorted(SEQ_INDEX):
        _check_sequence("oraclepin", name)


def test_independent_hevc_decoder_reproduces_golden_sequences():
    """THE pin for inter prediction (8.5.3: merge / AMVP / temporal candidates / weighted prediction, long-term references, ...): no stream of the
    reference's own fixtures holds a P or B picture, so these generated tracks are what an independent decoder is asked about"""
    dec = _independent_decoder()
    if dec is None:
        pytest.skip("HEVC inter-prediction parity UNPINNED: no independent HEVC decoder plugin is loadable by the reference libheif (see "
                    "test_independent_hevc_decoder_reproduces_golden)")
    for name in sorted(SEQ_INDEX):
        _check_sequence(dec[0], name)


def test_harness_selftest_activates_on_a_second_decoder():
    """the oracle behind the plugin ABI: found among the HEVC decoders, selected by id, every golden stream compared — so the harness below is
    known to work before a real independent decoder ever shows up (this is NOT a pin)"""
    sd = os.path.join(REF, "plugins_selftest")
    if not glob.glob(os.path.join(sd, "*.so")):
        pytest.skip("self-test plugin not built")
    _load_plugins([sd])
    ids = [i for i, _ in _hevc_decoder_ids()]
    assert "oraclepin" in ids
    done = 0
    for table, name in _golden():
        if INDEX[table][name]["width"] * INDEX[table][name]["height"] > 700 * 500:
            continue       # (keeps the CPU tier short; the real pin below takes every stream)
        _check("oraclepin", table, name)
        done += 1
    assert done >= 20


def test_independent_hevc_decoder_reproduces_golden():
    dec = _independent_decoder()
    if dec is None:
        pytest.skip("HEVC pixel parity UNPINNED: no independent HEVC decoder plugin is loadable by the reference libheif (libde265 is absent from this "
                    "image; `make -C oracle -f Makefile.ref pin` builds libheif's libde265 plugin as soon as pkg-config sees libde265)")
    for table, name in _golden():
        _check(dec[0], table, name)
