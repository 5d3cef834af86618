"""Pipelined look-ahead chains (hipdec_set_sequence_pipeline, libheif_amd/csrc/decoder_chains.inc): with D > 1 a chain of a sequence track is only
enqueued when its window is full and its pictures are held back until D chains are in flight - libheif keeps pushing samples while no picture comes
out (sequences/track_visual.cc:200-260) -, so a chain's CABAC launch runs beside the pixel steps of the chains in front of it.  What must not change:
every picture, its place in output order and its sample's user_data (against the oracle), what a corrupt sample does to its own track and to the
tracks beside it, and the plain form (D = 1)."""
import os
import sys
import threading
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import libheif_host as lh
from test_sequence_gpu import _set_lookahead, _p_sequence, _play_track, _nals

pytestmark = pytest.mark.gpu


@pytest.fixture
def pipeline3():
    from libheif_amd.decoder import set_sequence_pipeline
    set_sequence_pipeline(3)
    yield 3
    set_sequence_pipeline(3)      # (the library's default)
    _set_lookahead(32)


def _check(got, aus, refs, what=""):
    by_poc = {r["poc"]: r for r in refs}
    coding = [r["poc"] for r in refs]
    assert len(got) == len(aus), what
    for out_idx, (img, ud) in enumerate(got):
        assert ud == 900 + coding.index(out_idx), (what, out_idx, ud)
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], by_poc[out_idx]["planes"][c], err_msg="%s POC %d plane %d" % (what, out_idx, c))


@pytest.mark.parametrize("cfg", [dict(temporal_mvp=1, weighted_pred=1, inter_num_refs=3),
                                 dict(b_frames=2, b_ref=1, inter_num_refs=2, temporal_mvp=1),
                                 dict(b_frames=1, temporal_mvp=1, long_term_ref=1)], ids=["ippp", "ibbp", "long_term"])
@pytest.mark.parametrize("k", [2, 4])
def test_chains_in_flight_decode_bit_exact_in_output_order(cfg, k, pipeline3):
    """23 pictures, look-ahead k: five to eleven chains, up to three in flight; chain j + 1 predicts from (and takes its collocated motion from) pictures
    of chain j that are still being decoded when it is enqueued"""
    from libheif_amd.decoder import pipeline_stats
    aus, refs = _p_sequence(23, seed=31, **cfg)
    _set_lookahead(k)
    before = pipeline_stats()
    _check(_play_track(aus, refs), aus, refs)
    after = pipeline_stats()
    assert after[0] - before[0] >= 3, (before, after)      # chains were left in flight
    assert after[1] == before[1]                            # and none had to be undone


def test_tracks_side_by_side_with_chains_in_flight(pipeline3):
    from libheif_amd.decoder import pipeline_stats
    specs = [dict(n=17, temporal_mvp=1, inter_num_refs=2), dict(n=20, b_frames=2, b_ref=1, temporal_mvp=1), dict(n=13, w=136, h=104, b_frames=1, temporal_mvp=1),
             dict(n=15, amp=1, inter_num_refs=2)]
    tracks = []
    for t, cfg in enumerate(specs):
        cfg = dict(cfg)
        n = cfg.pop("n")
        tracks.append(_p_sequence(n, w=cfg.pop("w", 200), h=cfg.pop("h", 136), seed=80 + t, **cfg))
    _set_lookahead(3)
    before = pipeline_stats()
    results, errors = [None] * len(tracks), []

    def run(t):
        try:
            results[t] = _play_track(*tracks[t])
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(len(tracks))]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors
    for t, (aus, refs) in enumerate(tracks):
        _check(results[t], aus, refs, "track %d" % t)
    assert pipeline_stats()[0] > before[0]


@pytest.mark.parametrize("damage", ["payload", "cut"])
@pytest.mark.parametrize("where", [3, 9, 15], ids=["second_chain", "middle", "last_chain"])
def test_a_corrupt_sample_in_a_chain_in_flight_fails_its_own_track_only(where, damage, pipeline3):
    """"payload": bytes inside a sample's slice data are damaged - the device's CABAC parser finds out (the substream does not terminate where the slice
    header says) while the chain, and chains built on it, are in flight: they are undone and decoded again the plain way.  "cut": the slice data ends
    early - the host refuses to build the chain; that is reported the plain way too, behind the pictures in flight.  Either way the damaged track gets
    the error after the pictures in front of the damage, and the tracks beside it decode bit-exact"""
    from libheif_amd import HipDecError
    from libheif_amd.decoder import pipeline_stats, HipDecoder
    good = [_p_sequence(16, seed=90 + t, temporal_mvp=1, inter_num_refs=2) for t in range(2)]
    bad_aus, bad_refs = _p_sequence(16, seed=95)
    bad_aus = list(bad_aus)
    nals = _nals(bad_aus[where])
    last = nals[-1]
    if damage == "cut":
        cut = 4 + (len(last) - 4) // 2
        bad_aus[where] = b"".join(nals[:-1]) + (cut - 4).to_bytes(4, "big") + last[4:cut]
    else:
        last = bytearray(last)
        m = 4 + (len(last) - 4) * 2 // 3
        for i in range(m, min(m + 6, len(last) - 2)):
            last[i] ^= 0x5a
        bad_aus[where] = b"".join(nals[:-1]) + bytes(last)
    _set_lookahead(3)
    before = pipeline_stats()
    outcomes = {}

    def run_bad():
        d = HipDecoder()
        got, err = [], None
        try:
            for k, au in enumerate(bad_aus):
                d.push_data(au)
                r = d.next_picture(user_data=900 + k)
                while r is not None:
                    got.append(r)
                    r = d.next_picture()
            r = d.next_picture(flush=True)
            while r is not None:
                got.append(r)
                r = d.next_picture(flush=True)
        except HipDecError as e:
            err = e
        finally:
            d.free()
        outcomes["bad"] = (got, err)

    def run(name, aus, refs):
        try:
            outcomes[name] = _play_track(aus, refs)
        except Exception as e:      # noqa: BLE001
            outcomes[name] = e

    threads = [threading.Thread(target=run, args=("good%d" % t, g[0], g[1])) for t, g in enumerate(good)] + [threading.Thread(target=run_bad)]
    for th in threads: th.start()
    for th in threads: th.join()
    got, err = outcomes["bad"]
    assert isinstance(err, HipDecError), outcomes["bad"]
    by_poc = {r["poc"]: r for r in bad_refs}
    for out_idx, (img, ud) in enumerate(got):       # what came out in front of the error is right (IPPP: output order = coding order)
        assert out_idx < where and ud == 900 + out_idx
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], by_poc[out_idx]["planes"][c])
    for t, (aus, refs) in enumerate(good):
        assert not isinstance(outcomes["good%d" % t], Exception), outcomes["good%d" % t]
        _check(outcomes["good%d" % t], aus, refs, "track %d" % t)
    if damage == "payload":
        assert pipeline_stats()[1] > before[1]      # the failed chain was undone


def test_plain_form_leaves_nothing_in_flight():
    """D = 1: every chain is waited for where it is launched (the form of rounds 5 / 6; a host that wants the least delay)"""
    from libheif_amd.decoder import pipeline_stats, set_sequence_pipeline
    set_sequence_pipeline(1)
    aus, refs = _p_sequence(9, seed=33, temporal_mvp=1)
    _set_lookahead(3)
    before = pipeline_stats()
    try:
        _check(_play_track(aus, refs), aus, refs)
    finally:
        _set_lookahead(32)
        set_sequence_pipeline(3)
    assert pipeline_stats() == before


@pytest.mark.skipif(not lh.available(), reason="oracle/_ref/libheif.so not built")
def test_sequence_track_through_libheif_with_chains_in_flight(pipeline3):
    """the real libheif's track loop (heif_track_decode_next_image) over a pipelined decoder: every frame, in order"""
    import test_sequence_gpu as ts
    _set_lookahead(3)
    name = sorted(ts.TRACKS)[0]
    ts.test_sequence_track_through_libheif.__wrapped__(name, 3) if hasattr(ts.test_sequence_track_through_libheif, "__wrapped__") else ts.test_sequence_track_through_libheif(name, 3)
