"""Image sequences through the decoder boundary: libheif pushes one sample per push_data2 with a user_data, polls
decode_next_image2 for its frame, pushes the next (sequences/track_visual.cc:200-280, codecs/decoder.cc:355-563); only a chunk's first sample
carries the parameter sets.  Intra-only tracks and tracks with P pictures (IPPP: skip / merge / AMVP, several reference pictures, AMP), checked
at the C decoder object and at the plugin's own function table (the slots libheif calls), every frame against the oracle."""
import ctypes as C
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle import pyoracle as orc
import libheif_host as lh

pytestmark = pytest.mark.gpu


def _set_lookahead(k):
    """samples the decoder gathers behind a track's first picture before it decodes them as ONE launch set (hipdec_set_sequence_lookahead)"""
    import libheif_amd
    lib = libheif_amd.load_library()
    lib.hipdec_set_sequence_lookahead.argtypes = [C.c_int]
    lib.hipdec_set_sequence_lookahead.restype = None
    lib.hipdec_set_sequence_lookahead(k)


@pytest.fixture(params=[32, 0, 3], ids=["lookahead32", "lookahead0", "lookahead3"])
def lookahead(request):
    _set_lookahead(request.param)
    yield request.param
    _set_lookahead(32)


def _nals(stream):
    out, p = [], 0
    while p < len(stream):
        n = int.from_bytes(stream[p:p + 4], "big")
        out.append(stream[p:p + 4 + n])
        p += 4 + n
    return out


def _samples(n, w=200, h=136, **cfg):
    """n pictures of one configuration: the first sample with parameter sets, the others slice data only; plus the oracle's planes"""
    streams = [orc.encode(orc.synth_image(w, h, cfg.get("bit_depth", 8), 1, seed=40 + i), **cfg) for i in range(n)]
    refs = [orc.decode(s) for s in streams]
    samples = [streams[0]] + [b"".join(x for x in _nals(s) if (x[4] >> 1) & 63 < 32) for s in streams[1:]]
    assert all(all((x[4] >> 1) & 63 < 32 for x in _nals(s)) for s in samples[1:])
    return samples, refs


@pytest.mark.parametrize("cfg", [dict(), dict(wpp=0, stress=1, num_slices=2), dict(bit_depth=10)], ids=["default", "slices", "main10"])
def test_samples_after_the_first_reuse_the_parameter_sets(cfg):
    from libheif_amd.decoder import HipDecoder
    samples, refs = _samples(4, **cfg)
    d = HipDecoder()
    for s, ref in zip(samples, refs):
        d.push_data(s)
        img = d.decode_next_image()
        assert img is not None and d.decode_next_image() is None      # one frame per sample, nothing pending behind it
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c])
    # a sample that brings new parameter sets (another picture size) replaces them
    other = orc.encode(orc.synth_image(64, 72, cfg.get("bit_depth", 8), 1, seed=9), **cfg)
    d.push_data(other)
    img = d.decode_next_image()
    ref = orc.decode(other)
    assert (img.info["width"], img.info["height"]) == (64, 72)
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c])
    d.free()


class HeifError(C.Structure):
    _fields_ = [("code", C.c_int), ("subcode", C.c_int), ("message", C.c_char_p)]


@pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")
def test_plugin_function_table_round_trips_user_data_per_sample(lookahead):
    """the slots of heif_decoder_plugin as libheif calls them for a track: new_decoder2, then per sample push_data2(user_data) and
    decode_next_image2(&user_data) (api/libheif/heif_plugin.h:85-169)"""
    import libheif_amd
    L = lh.load_hip_plugin()           # libheif in the process (the plugin resolves heif_image_create & co from it) + init_plugin
    P = C.CDLL(libheif_amd.library_path())
    vp = C.c_void_p

    class PluginInfo(C.Structure):
        _fields_ = [("version", C.c_int), ("type", C.c_int), ("plugin", vp)]

    class Plugin(C.Structure):
        _fields_ = [("plugin_api_version", C.c_int), ("get_plugin_name", vp), ("init_plugin", vp), ("deinit_plugin", vp), ("does_support_format", vp),
                    ("new_decoder", vp), ("free_decoder", C.CFUNCTYPE(None, vp)), ("push_data", vp), ("decode_image", vp), ("set_strict_decoding", vp),
                    ("id_name", C.c_char_p), ("decode_next_image", vp), ("minimum_required_libheif_version", C.c_uint32), ("does_support_format2", vp),
                    ("new_decoder2", C.CFUNCTYPE(HeifError, C.POINTER(vp), vp)),
                    ("push_data2", C.CFUNCTYPE(HeifError, vp, C.c_char_p, C.c_size_t, C.c_size_t)),
                    ("flush_data", C.CFUNCTYPE(HeifError, vp)),
                    ("decode_next_image2", C.CFUNCTYPE(HeifError, vp, C.POINTER(vp), C.POINTER(C.c_size_t), vp))]

    info = PluginInfo.in_dll(P, "plugin_info")
    plug = C.cast(info.plugin, C.POINTER(Plugin)).contents
    assert plug.plugin_api_version >= 5

    class Options(C.Structure):   # heif_decoder_plugin_options (heif_plugin.h:70-82)
        _fields_ = [("format", C.c_int), ("strict_decoding", C.c_int), ("num_threads", C.c_int), ("limits", vp)]

    opts = Options(lh.COMPRESSION_HEVC, 0, 0, None)
    dec = vp()
    e = plug.new_decoder2(C.byref(dec), C.cast(C.byref(opts), vp))
    assert e.code == 0, e.message
    samples, refs = _samples(3)
    p_aus, p_refs = _p_sequence(4, inter_num_refs=2, amp=1)      # then a track with P pictures through the same decoder instance
    samples, refs = samples + p_aus, refs + p_refs
    got = []

    def poll_all():
        while True:
            img, ud = vp(), C.c_size_t(12345)
            e = plug.decode_next_image2(dec, C.byref(img), C.byref(ud), None)
            assert e.code == 0, e.message
            if not img.value:                  # Ok with *out == NULL (decoder.cc:538-552): libheif pushes the next sample (track_visual.cc:200-260)
                return
            got.append(([lh._plane(L, img, ch) for ch in (lh.CHANNEL_Y, lh.CHANNEL_CB, lh.CHANNEL_CR)], ud.value))
            L.heif_image_release(img)

    try:
        for k, s in enumerate(samples):
            e = plug.push_data2(dec, s, len(s), 1000 + k)
            assert e.code == 0, e.message
            poll_all()
            if k == 0 or lookahead <= 1:
                assert len(got) == k + 1       # the first picture of a decoder never waits; without look-ahead none does (no B pictures here)
        assert plug.flush_data(dec).code == 0
        poll_all()
        assert len(got) == len(samples)
        for k, ((planes, ud), ref) in enumerate(zip(got, refs)):
            assert ud == 1000 + k              # coding order == output order in these tracks; every picture carries its own sample's user_data
            for c in range(3):
                np.testing.assert_array_equal(planes[c], ref["planes"][c], err_msg="sample %d plane %d" % (k, c))
    finally:
        plug.free_decoder(dec)

    # a track with B pictures: samples in coding order; decode_next_image2 returns Ok + NULL while the next picture in output order is still to
    # come (libheif then pushes the next sample, track_visual.cc:200-260), flush_data releases what is left; user_data travels with the picture
    dec = vp()
    assert plug.new_decoder2(C.byref(dec), C.cast(C.byref(opts), vp)).code == 0
    b_aus, b_refs = _p_sequence(7, b_frames=2, b_ref=1, temporal_mvp=1, weighted_pred=1, inter_num_refs=2)
    by_poc = {r["poc"]: r for r in b_refs}
    coding_pocs = [r["poc"] for r in b_refs]
    outputs = []

    def poll():
        while True:
            img, ud = vp(), C.c_size_t(0)
            e = plug.decode_next_image2(dec, C.byref(img), C.byref(ud), None)
            assert e.code == 0, e.message
            if not img.value:
                return
            outputs.append(([lh._plane(L, img, ch) for ch in (lh.CHANNEL_Y, lh.CHANNEL_CB, lh.CHANNEL_CR)], ud.value))
            L.heif_image_release(img)

    try:
        for k, s in enumerate(b_aus):
            assert plug.push_data2(dec, s, len(s), 7000 + k).code == 0
            poll()
        assert len(outputs) < len(b_aus)                      # the last anchor's B pictures (and the samples the look-ahead holds) were still waiting
        assert plug.flush_data(dec).code == 0
        poll()
        assert len(outputs) == len(b_aus)
        for poc, (planes, ud) in enumerate(outputs):
            assert ud == 7000 + coding_pocs.index(poc)
            for c in range(3):
                np.testing.assert_array_equal(planes[c], by_poc[poc]["planes"][c], err_msg="B track POC %d plane %d" % (poc, c))
    finally:
        plug.free_decoder(dec)


def _p_sequence(n, w=200, h=136, bit_depth=8, **cfg):
    from test_inter_oracle import make_frames
    frames = make_frames(w, h, n, bit_depth)
    cfg = dict(cfg)
    aus = orc.encode_sequence(frames, bit_depth=bit_depth, qp=cfg.pop("qp", 26), global_mv_x=cfg.pop("global_mv_x", -8), global_mv_y=cfg.pop("global_mv_y", -4), **cfg)
    return aus, orc.decode_sequence(aus)


@pytest.mark.parametrize("cfg", [dict(), dict(amp=1, inter_num_refs=3, max_merge_cand=3, parallel_merge_level=4, log2_ctb=4, log2_max_tb=4),
                                 dict(stress=1, amp=1, inter_num_refs=2, lists_modification=1, cabac_init_present=1, num_slices=2, wpp=0),
                                 dict(bit_depth=10, tile_cols=2, tile_rows=2, inter_num_refs=2), dict(w=70, h=42, amp=1, inter_num_refs=2, global_mv_y=17)],
                         ids=["default", "amp_multiref_mer", "stress_slices_listmod", "main10_tiles", "cropped"])
def test_p_pictures_decode_bit_exact_through_the_decoder_object(cfg):
    """IPPP tracks: every sample pushed on its own (the first with the parameter sets), one frame per decode, planes == oracle"""
    from libheif_amd.decoder import HipDecoder
    cfg = dict(cfg)
    aus, refs = _p_sequence(5, w=cfg.pop("w", 200), h=cfg.pop("h", 136), bit_depth=cfg.pop("bit_depth", 8), **cfg)
    d = HipDecoder()
    for k, (au, ref) in enumerate(zip(aus, refs)):
        d.push_data(au)
        img = d.decode_next_image()
        assert img is not None and d.decode_next_image() is None
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="picture %d plane %d" % (k, c))
    d.free()


def test_two_tracks_decode_side_by_side():
    """two decoder instances with their own reference pictures, samples interleaved (libheif decodes tracks independently)"""
    from libheif_amd.decoder import HipDecoder
    a_aus, a_refs = _p_sequence(4, seed=3)
    b_aus, b_refs = _p_sequence(4, w=136, h=104, seed=9, inter_num_refs=2)
    da, db = HipDecoder(), HipDecoder()
    for k in range(4):
        for d, aus, refs in ((da, a_aus, a_refs), (db, b_aus, b_refs)):
            d.push_data(aus[k])
            img = d.decode_next_image()
            for c in range(3):
                np.testing.assert_array_equal(img.planes[c], refs[k]["planes"][c])
    da.free(); db.free()


def test_p_picture_without_its_reference_is_an_error_not_a_misdecode():
    from libheif_amd.decoder import HipDecoder
    from libheif_amd import HipDecError
    aus, _ = _p_sequence(3)
    d = HipDecoder()
    d.push_data(aus[0]); d.decode_next_image()
    d.push_data(aus[2])                       # its RPS names POC 1, which was never pushed
    with pytest.raises(HipDecError):
        d.decode_next_image()
    d.free()


def test_inter_slices_outside_a_sequence_are_refused_loudly():
    """a P slice header (slice_type 1) in place of the I slice of a still: UNSUPPORTED, not a mis-decode"""
    from libheif_amd.decoder import HipDecoder
    from libheif_amd import HipDecError
    s = orc.encode(orc.synth_image(64, 64, 8, 1, seed=3), wpp=0)
    nals = _nals(s)
    out = []
    for x in nals:
        t = (x[4] >> 1) & 63
        if t < 32:
            b = bytearray(x)
            # slice_segment_header of an IDR: first_slice_segment_in_pic_flag(1) no_output_of_prior_pics_flag(1) slice_pic_parameter_set_id ue(v)=1
            # then slice_type ue(v): I = 2 -> '011'; P = 1 -> '010'.  bits: 1 x 1 | 011 ... -> flip the last bit of '011'
            assert (b[6] >> 2) & 7 == 0b011, "unexpected slice header layout"
            b[6] &= ~(1 << 2) & 0xff
            x = bytes(b)
        out.append(x)
    d = HipDecoder()
    d.push_data(b"".join(out))
    with pytest.raises(HipDecError) as e:
        d.decode_next_image()
    assert e.value.code == -4
    d.free()


# ---- B pictures, temporal motion vector prediction, weighted prediction: what libheif's x265 plugin writes for its "lowdelay" and "unrestricted" GOP
#      structures (libheif/plugins/encoder_x265.cc:875-888) ------------------------------------------------------------------------------------------
B_CASES = {
    "lowdelay_tmvp_weighted": dict(temporal_mvp=1, weighted_pred=1, inter_num_refs=3),
    "b1": dict(b_frames=1, temporal_mvp=1),
    "b2_ref_multiref_weighted": dict(b_frames=2, b_ref=1, inter_num_refs=2, temporal_mvp=1, weighted_pred=1, mvd_l1_zero=1, amp=1),
    "b3_slices_listmod": dict(b_frames=3, temporal_mvp=1, num_slices=2, lists_modification=1, cabac_init_present=1, max_merge_cand=4, wpp=0),
    "b_main10_tiles": dict(b_frames=2, b_ref=1, temporal_mvp=1, bit_depth=10, tile_cols=2, tile_rows=2, inter_num_refs=2),
    "b_cropped": dict(b_frames=1, temporal_mvp=1, weighted_pred=1, w=70, h=42, global_mv_y=17, inter_num_refs=2),
    "b_scaling_lists_sps": dict(b_frames=1, temporal_mvp=1, scaling_list=2, inter_intra_pct=30),      # inter matrices (matrixId 3 .. 5) beside the intra ones
    "p_scaling_lists_default": dict(scaling_list=1, inter_num_refs=2, inter_intra_pct=30),
    "b_constrained_intra_pred": dict(b_frames=1, temporal_mvp=1, constrained_intra_pred=1, inter_intra_pct=45, log2_ctb=4, log2_max_tb=4),
    "b_long_term_ref": dict(b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, long_term_ref=1),
    "p_long_term_ref_sps_listmod": dict(temporal_mvp=1, inter_num_refs=3, lists_modification=1, weighted_pred=1, long_term_ref=3),
}


@pytest.mark.parametrize("name", sorted(B_CASES))
def test_b_tmvp_weighted_sequences_decode_bit_exact_in_output_order(name, lookahead):
    """samples pushed in CODING order (as the track stores them); the pictures come out in OUTPUT order (bumping, C.5.2.2), each with the user_data of
    its own sample, each bit-exact against the oracle's picture of that POC"""
    from libheif_amd.decoder import HipDecoder
    cfg = dict(B_CASES[name])
    n = 8
    aus, refs = _p_sequence(n, w=cfg.pop("w", 200), h=cfg.pop("h", 136), bit_depth=cfg.pop("bit_depth", 8), **cfg)
    by_poc = {r["poc"]: r for r in refs}
    coding_pocs = [r["poc"] for r in refs]
    d = HipDecoder()
    got = []
    for k, au in enumerate(aus):
        d.push_data(au)
        r = d.next_picture(user_data=500 + k)
        while r is not None:
            got.append(r)
            r = d.next_picture()
    r = d.next_picture(flush=True)
    while r is not None:
        got.append(r)
        r = d.next_picture(flush=True)
    assert len(got) == n
    for out_idx, (img, ud) in enumerate(got):
        assert ud == 500 + coding_pocs.index(out_idx), (out_idx, ud)          # output order == POC order; user_data of the sample that coded it
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], by_poc[out_idx]["planes"][c], err_msg="%s POC %d plane %d" % (name, out_idx, c))
    d.free()


def _play_track(aus, refs, look_for_error=False):
    """a track the way libheif drives it: push a sample, take every picture that is ready; flush at the end.  Returns [(planes, user_data)] in output order"""
    from libheif_amd.decoder import HipDecoder
    d = HipDecoder()
    got = []
    try:
        for k, au in enumerate(aus):
            d.push_data(au)
            r = d.next_picture(user_data=900 + k)
            while r is not None:
                got.append(r)
                r = d.next_picture()
        r = d.next_picture(flush=True)
        while r is not None:
            got.append(r)
            r = d.next_picture(flush=True)
    finally:
        d.free()
    return got


def test_tracks_side_by_side_share_launch_sets(lookahead):
    """six tracks with different GOP structures, picture sizes and lengths, one host thread and one decoder instance each (as libheif's Track_Visual
    objects): the look-ahead chains of tracks that ask together run as ONE launch set (decoder.hip: ChainCoalescer; step k = step k of every track) and
    every picture of every track still equals the oracle's, in output order, with its own sample's user_data"""
    import threading
    from libheif_amd.decoder import chain_stats
    specs = [dict(n=9, temporal_mvp=1, weighted_pred=1, inter_num_refs=3),
             dict(n=12, b_frames=2, b_ref=1, inter_num_refs=2, temporal_mvp=1),
             dict(n=7, w=136, h=104, b_frames=1, temporal_mvp=1, long_term_ref=1),
             dict(n=10, w=70, h=42, amp=1, inter_num_refs=2, global_mv_y=17),
             dict(n=9, scaling_list=2, b_frames=1, temporal_mvp=1, inter_intra_pct=30),
             dict(n=11, temporal_mvp=0, log2_ctb=5)]
    tracks = []
    for k, cfg in enumerate(specs):
        cfg = dict(cfg)
        n = cfg.pop("n")
        tracks.append(_p_sequence(n, w=cfg.pop("w", 200), h=cfg.pop("h", 136), seed=60 + k, **cfg))
    before = chain_stats()
    for attempt in range(3):      # (whether two threads' chains meet is a matter of timing: the pictures are checked every time, the sharing once)
        results, errors = [None] * len(tracks), []

        def run(t):
            try:
                results[t] = _play_track(*tracks[t])
            except Exception as e:      # noqa: BLE001 - reported below with the track's number
                errors.append((t, repr(e)))

        threads = [threading.Thread(target=run, args=(t,)) for t in range(len(tracks))]
        for th in threads: th.start()
        for th in threads: th.join()
        assert not errors, errors
        for t, (aus, refs) in enumerate(tracks):
            by_poc = {r["poc"]: r for r in refs}
            coding = [r["poc"] for r in refs]
            assert len(results[t]) == len(aus)
            for out_idx, (img, ud) in enumerate(results[t]):
                assert ud == 900 + coding.index(out_idx), (t, out_idx, ud)
                for c in range(3):
                    np.testing.assert_array_equal(img.planes[c], by_poc[out_idx]["planes"][c], err_msg="track %d POC %d plane %d" % (t, out_idx, c))
        after = chain_stats()
        assert after[0] > before[0]
        if not lookahead or after[2] > before[2]:
            break
    else:
        raise AssertionError("no launch set held more than one track's chain in three runs: %r -> %r" % (before, after))


def test_a_corrupt_track_beside_good_ones_fails_alone():
    """one track's sample is damaged (its slice data cut short): the shared launch set is given up, every track runs on its own, the damaged one reports
    the error and the others decode bit-exact"""
    import threading
    from libheif_amd import HipDecError
    good = [_p_sequence(6, seed=70 + k, temporal_mvp=1, inter_num_refs=2) for k in range(3)]
    bad_aus, _ = _p_sequence(6, seed=75)
    bad_aus = list(bad_aus)
    nals = _nals(bad_aus[3])
    last = nals[-1]
    cut = 4 + (len(last) - 4) // 2
    bad_aus[3] = b"".join(nals[:-1]) + (cut - 4).to_bytes(4, "big") + last[4:cut]      # a slice NAL whose data ends early: the device parser runs out of bitstream
    outcomes = {}

    def run(name, aus, refs):
        try:
            outcomes[name] = _play_track(aus, refs)
        except HipDecError as e:
            outcomes[name] = e

    threads = [threading.Thread(target=run, args=("good%d" % k, g[0], g[1])) for k, g in enumerate(good)]
    threads.append(threading.Thread(target=run, args=("bad", bad_aus, None)))
    for th in threads: th.start()
    for th in threads: th.join()
    assert isinstance(outcomes["bad"], HipDecError), outcomes["bad"]
    for k, (aus, refs) in enumerate(good):
        got = outcomes["good%d" % k]
        assert not isinstance(got, Exception), got
        by_poc = {r["poc"]: r for r in refs}
        assert len(got) == len(aus)
        for out_idx, (img, _) in enumerate(got):
            for c in range(3):
                np.testing.assert_array_equal(img.planes[c], by_poc[out_idx]["planes"][c], err_msg="track %d POC %d plane %d" % (k, out_idx, c))


def test_track_entered_at_a_cra_picture_drops_its_rasl_pictures(lookahead):
    """IDR P B B CRA RASL RASL P B B, decoded from the CRA picture on (a seek to a sync sample): the CRA picture's PicOrderCnt comes from its LSBs
    alone, its two RASL pictures predict from a picture that was never decoded and are dropped (8.3.3) - no picture, no error -, everything else equals
    the full track's pictures, in output order with the user_data of its own sample"""
    from test_inter_oracle import make_frames
    from test_inter_emu import parameter_sets
    frames = make_frames(136, 104, 10)
    aus = orc.encode_sequence(frames, qp=27, b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, open_gop=2, seed=3)
    full = {p["poc"]: p for p in orc.decode_sequence(aus)}
    entered = [parameter_sets(aus[0]) + aus[4]] + list(aus[5:])      # CRA, RASL, RASL, P (POC 9), B (7), B (8)
    got = _play_track(entered, None)
    assert [ud for _, ud in got] == [900, 904, 905, 903], [ud for _, ud in got]
    for (img, _), poc in zip(got, [6, 7, 8, 9]):
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], full[poc]["planes"][c], err_msg="PicOrderCnt %d plane %d" % (poc, c))
    # and the whole track from its IDR picture: the CRA picture and its RASL pictures are ordinary pictures there
    got = _play_track(aus, None)
    assert len(got) == 10
    for out_idx, (img, _) in enumerate(got):
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], full[out_idx]["planes"][c], err_msg="from the IDR picture: PicOrderCnt %d plane %d" % (out_idx, c))


@pytest.mark.parametrize("hidden", [3, 2], ids=["an_anchor", "a_reference_b"])
def test_picture_with_pic_output_flag_0_is_decoded_but_never_output(hidden, lookahead):
    """output_flag_present_flag = 1 and one picture with pic_output_flag = 0 (7.4.7.1): it is a reference picture of its neighbours like any other, the
    bumping process never hands it out (C.5.2.2) - the track delivers one picture less, the others unchanged and in output order"""
    aus, refs = _p_sequence(8, b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, hidden_poc=hidden, seed=5)
    by_poc = {r["poc"]: r for r in refs}
    coding = [r["poc"] for r in refs]
    got = _play_track(aus, refs)
    shown = [p for p in range(8) if p != hidden]
    assert [ud for _, ud in got] == [900 + coding.index(p) for p in shown]
    for (img, _), poc in zip(got, shown):
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], by_poc[poc]["planes"][c], err_msg="PicOrderCnt %d plane %d" % (poc, c))


def test_b_sequence_in_coding_order_through_the_legacy_call():
    """hipdec_decoder_decode keeps delivering the picture of the sample just pushed (coding order): the planes are those of that POC"""
    from libheif_amd.decoder import HipDecoder
    aus, refs = _p_sequence(6, b_frames=2, temporal_mvp=1, b_ref=1)
    d = HipDecoder()
    for au, ref in zip(aus, refs):
        d.push_data(au)
        img = d.decode_next_image()
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="POC %d" % ref["poc"])
    d.free()


def test_two_coded_video_sequences_back_to_back_come_out_in_order(lookahead):
    """an IDR picture in the middle of a track starts a new coded video sequence: POCs start over, and everything of the first sequence that is still
    waiting for output precedes it (C.5.2.2); no flush in between"""
    from libheif_amd.decoder import HipDecoder
    a_aus, a_refs = _p_sequence(6, b_frames=2, temporal_mvp=1, seed=3)
    b_aus, b_refs = _p_sequence(5, b_frames=1, b_ref=0, temporal_mvp=1, weighted_pred=1, seed=8)
    expect = [r for _, r in sorted((r["poc"], r) for r in a_refs)] + [r for _, r in sorted((r["poc"], r) for r in b_refs)]
    d = HipDecoder()
    got = []
    for au in a_aus + b_aus:          # (the second sequence's first sample carries its own parameter sets and an IDR picture)
        d.push_data(au)
        r = d.next_picture()
        while r is not None:
            got.append(r[0])
            r = d.next_picture()
    r = d.next_picture(flush=True)
    while r is not None:
        got.append(r[0])
        r = d.next_picture(flush=True)
    assert len(got) == len(expect)
    for k, (img, ref) in enumerate(zip(got, expect)):
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="output %d (POC %d) plane %d" % (k, ref["poc"], c))
    d.free()


def test_whole_track_pushed_at_once_is_split_into_access_units(lookahead):
    """libde265 takes any number of pictures per push (decoder_libde265.cc:322-368 only frames NAL units): a host that pushes a whole track in one
    call gets every picture, in output order; the first access unit is the still-image case, the rest goes through the look-ahead in chains"""
    from libheif_amd.decoder import HipDecoder
    aus, refs = _p_sequence(11, b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2)
    expect = [r for _, r in sorted((r["poc"], r) for r in refs)]
    d = HipDecoder()
    d.push_data(b"".join(aus))
    got = []
    r = d.next_picture(flush=True)
    while r is not None:
        got.append(r[0])
        r = d.next_picture(flush=True)
    assert len(got) == len(expect)
    for k, (img, ref) in enumerate(zip(got, expect)):
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="output %d plane %d" % (k, c))
    d.free()


def test_a_picture_pushed_in_pieces_is_one_sample():
    """heif_plugin.h:113-115: push_data may be called several times for one picture - parameter sets, then the slice segments one by one"""
    from libheif_amd.decoder import HipDecoder
    aus, refs = _p_sequence(3, num_slices=3, wpp=0)
    d = HipDecoder()
    for au, ref in zip(aus, refs):
        for x in _nals(au):
            d.push_data(x)
        img = d.decode_next_image()
        assert img is not None and d.decode_next_image() is None
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c])
    d.free()


# ---- f3 through the REAL libheif: an image-sequence file (moov / trak), heif_track_decode_next_image() until End_of_sequence -----------------------
# libheif's default decoding options convert every decoded image to the sRGB nclx (context.cc:1533-1558), so the tracks signal exactly that profile in
# the VUI: the plugin's planes then pass through untouched (as tests/test_plugin_dropin.py does for stills)
SRGB_VUI = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
TRACKS = {
    "ippp": dict(inter_num_refs=2, amp=1),
    "ibbp_tmvp_weighted": dict(b_frames=2, b_ref=1, temporal_mvp=1, weighted_pred=1, inter_num_refs=2),
    "ibp_main10": dict(b_frames=1, temporal_mvp=1, bit_depth=10),
    "intra_only": None,
}


@pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")
@pytest.mark.parametrize("name", sorted(TRACKS))
def test_sequence_track_through_libheif(name, lookahead):
    """libheif/sequences/track_visual.cc:175-330 (Track_Visual::decode_next_image_sample) drives the plugin: one push_data2 per sample, polls of
    decode_next_image2 in between, flush_data behind the last sample.  Every image of the track, in the order libheif delivers them (= output
    order), equals the oracle's picture of that POC; with the look-ahead the plugin answers 'no image yet' until its window is full."""
    from heic_util import build_sequence
    lh.load_hip_plugin()
    n = 13
    if TRACKS[name] is None:
        aus, refs = _samples(6, **SRGB_VUI)
        expect = refs
        bd = 8
    else:
        cfg = dict(TRACKS[name], **SRGB_VUI)
        bd = cfg.pop("bit_depth", 8)
        aus, refs = _p_sequence(n, bit_depth=bd, **cfg)
        expect = [r for _, r in sorted((r["poc"], r) for r in refs)]
    data = build_sequence(aus, 200, 136, bit_depth=bd)
    got = lh.decode_track(data)
    assert len(got) == len(expect)
    for k, (g, r) in enumerate(zip(got, expect)):
        assert g["bit_depth"] == bd
        for c in range(3):
            np.testing.assert_array_equal(g["planes"][c], r["planes"][c], err_msg="%s: image %d plane %d" % (name, k, c))


@pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")
def test_sequence_track_to_rgb_through_libheif():
    """the same call with interleaved RGB requested: libheif's colour conversion runs behind every frame; expectation = the compiled reference
    colour op (ref_harness) on the oracle's planes"""
    import ref_harness as rh
    from heic_util import build_sequence
    lh.load_hip_plugin()
    aus, refs = _p_sequence(6, b_frames=1, temporal_mvp=1, **SRGB_VUI)
    expect = [r for _, r in sorted((r["poc"], r) for r in refs)]
    got = lh.decode_track(build_sequence(aus, 200, 136), lh.COLORSPACE_RGB, lh.CHROMA_RGB)
    assert len(got) == len(expect)
    for k, (g, r) in enumerate(zip(got, expect)):
        exp = rh.convert(r["planes"], 8, rh.CH_420, r["nclx"], rh.CS_RGB, rh.CH_RGB)[0]
        np.testing.assert_array_equal(g["rgb"], exp[:, :200 * 3], err_msg="image %d" % k)
