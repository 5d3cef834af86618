"""P pictures in the CPU oracle (oracle/hevc_oracle_inter.c) and the test-stream generator (oracle/hevc_testenc_inter.c): SURVEY.md 8 f3,
the samples libheif's sequence tracks push through the decoder plugin (libheif/sequences/track_visual.cc:200-280).

No fixture of the reference holds inter-coded pictures (parity unpinned).  What pins the SYNTAX is the exact round trip: with every coding
unit coded lossless (cu_transquant_bypass) and no skipped units, decoded == source for every sample of every picture - through the
generator's CABAC writer, the oracle's parser, candidate derivation, interpolation and residual path, over the coding-tool matrix."""
import numpy as np
import pytest

from oracle import pyoracle as orc


def shifted(planes, dx, dy):
    h, w = planes[0].shape
    out = []
    for i, p in enumerate(planes):
        sx = 1 if (i == 0 or p.shape[1] == w) else 2      # SubWidthC / SubHeightC of the plane
        sy = 1 if (i == 0 or p.shape[0] == h) else 2
        out.append(np.roll(np.roll(p, dy // sy, 0), dx // sx, 1))
    return out


def make_frames(w, h, n, bit_depth=8, mono=False, seed=5, chroma_format_idc=1):
    f0 = orc.synth_image(w, h, bit_depth, 0 if mono else chroma_format_idc, seed=seed)
    return [shifted(f0, 2 * k, k) for k in range(n)]


CONFIGS = {
    "default": dict(),
    "amp_multiref_mer": dict(amp=1, inter_num_refs=3, max_merge_cand=3, parallel_merge_level=4, log2_ctb=4, log2_max_tb=4),
    "stress_slices_listmod": dict(stress=1, amp=1, inter_num_refs=2, log2_min_cb=4, log2_ctb=5, lists_modification=1, cabac_init_present=1, num_slices=2,
                                  max_transform_hierarchy_depth_inter=0),
    "tiles_wpp": dict(tile_cols=2, tile_rows=2, wpp=1, log2_ctb=4, log2_max_tb=4, inter_num_refs=2, max_merge_cand=1),
    "no_wpp_one_cand": dict(wpp=0, max_merge_cand=2, inter_merge_pct=80, inter_intra_pct=30),
    "min_cb16_nxn": dict(log2_min_cb=4, log2_ctb=6, inter_merge_pct=10, max_transform_hierarchy_depth_inter=2),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_lossless_p_pictures_round_trip_exactly(name):
    frames = make_frames(136, 104, 4)
    aus = orc.encode_sequence(frames, qp=30, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, lossless_pct=100, seed=11, **CONFIGS[name])
    pics = orc.decode_sequence(aus, taps=True)
    for i, p in enumerate(pics):
        assert p["poc"] == i
        for c in range(3):
            np.testing.assert_array_equal(p["planes"][c], frames[i][c], err_msg="%s: picture %d component %d" % (name, i, c))
    assert all((p["map_pred"] == 1).mean() > 0.5 for p in pics[1:])       # the P pictures really are inter coded
    assert (pics[0]["map_pred"] == 0).all()


@pytest.mark.parametrize("bit_depth,mono", [(10, False), (8, True), (12, False)])
def test_lossless_round_trip_other_formats(bit_depth, mono):
    frames = make_frames(72, 56, 3, bit_depth, mono)
    aus = orc.encode_sequence(frames, bit_depth=bit_depth, qp=28, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, lossless_pct=100, amp=1, inter_num_refs=2)
    pics = orc.decode_sequence(aus)
    for i, p in enumerate(pics):
        for c in range(1 if mono else 3):
            np.testing.assert_array_equal(p["planes"][c], frames[i][c])


def test_lossy_sequence_with_skipped_units_decodes_and_tracks_the_source():
    """lossy P pictures (skipped units, deblocking with motion-dependent filtering strength, SAO): every substream must end exactly on its entry
    point (the oracle checks that) and the pictures must stay close to their sources"""
    frames = make_frames(200, 136, 5)
    aus = orc.encode_sequence(frames, qp=24, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=25, inter_num_refs=2, amp=1, sao=1)
    pics = orc.decode_sequence(aus, taps=True)
    for i, p in enumerate(pics):
        mse = float(np.mean((p["planes"][0].astype(np.float64) - frames[i][0]) ** 2))
        assert mse < 600.0, (i, mse)     # (skipped units take a random merge candidate and carry no residual: coarse, but bounded)
    hist = np.bincount(pics[2]["map_pred"].ravel(), minlength=3)
    assert hist[1] > 0 and hist[2] > 0                       # inter and skipped units both occur
    assert pics[3]["mf_ref"][..., 0].max() >= 1                      # the second reference picture is used


def test_no_drift_between_generator_and_decoder():
    """the generator predicts from ITS decoded pictures (deblocked with motion-dependent strength, SAO applied): if the decoder's reference
    pictures differed, the error would accumulate over the sequence; at a fine quantiser it must stay at the quantisation noise"""
    frames = make_frames(200, 136, 6)
    aus = orc.encode_sequence(frames, qp=4, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, inter_num_refs=2, amp=1)
    for i, p in enumerate(orc.decode_sequence(aus)):
        assert float(np.mean((p["planes"][0].astype(np.float64) - frames[i][0]) ** 2)) < 3.0, i


B_CONFIGS = {
    "tmvp_p": dict(temporal_mvp=1, inter_num_refs=2),
    "weighted_p": dict(weighted_pred=1, inter_num_refs=2),
    "b1": dict(b_frames=1),
    "b2_ref_multiref": dict(b_frames=2, b_ref=1, inter_num_refs=2),
    "b_tmvp": dict(b_frames=2, temporal_mvp=1, inter_num_refs=2, b_ref=1, max_merge_cand=5),
    "b_weighted_mvdl1zero": dict(b_frames=1, weighted_pred=1, mvd_l1_zero=1, inter_bi_pct=80),
    "b_everything": dict(b_frames=2, temporal_mvp=1, weighted_pred=1, mvd_l1_zero=1, amp=1, inter_num_refs=2, b_ref=1, lists_modification=1,
                         cabac_init_present=1, num_slices=2, max_merge_cand=4, parallel_merge_level=3),
    "b_tiles_small_ctb": dict(b_frames=3, temporal_mvp=1, tile_cols=2, tile_rows=2, wpp=0, log2_ctb=4, log2_max_tb=4, inter_bi_pct=70),
}


@pytest.mark.parametrize("name", sorted(B_CONFIGS))
def test_lossless_b_tmvp_weighted_round_trip_exactly(name):
    """B pictures (coded after the anchor that follows them: coding order != POC order), temporal motion vector prediction and explicit weighted
    prediction: with every coding unit lossless, each decoded picture equals the source frame of its POC"""
    frames = make_frames(136, 104, 7)
    aus = orc.encode_sequence(frames, qp=30, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, lossless_pct=100, seed=11, **B_CONFIGS[name])
    pics = orc.decode_sequence(aus, taps=True)
    assert sorted(p["poc"] for p in pics) == list(range(7))
    b = B_CONFIGS[name].get("b_frames", 0)
    if b:
        assert [p["poc"] for p in pics][:3] == [0, b + 1, 1]              # the anchor is coded before the B pictures it follows
    for p in pics:
        for c in range(3):
            np.testing.assert_array_equal(p["planes"][c], frames[p["poc"]][c], err_msg="%s: POC %d component %d" % (name, p["poc"], c))
    if b:
        bi = [int(((p["mf_ref"][..., 0] >= 0) & (p["mf_ref"][..., 1] >= 0)).sum()) for p in pics]
        l1_only = [int(((p["mf_ref"][..., 0] < 0) & (p["mf_ref"][..., 1] >= 0)).sum()) for p in pics]
        assert max(bi) > 50 and max(l1_only) > 10, (bi, l1_only)          # bi-predicted and list-1-only blocks both occur


CHROMA_FORMAT_CONFIGS = {
    "p_default": dict(),
    "p_amp_multiref_tmvp": dict(amp=1, inter_num_refs=2, temporal_mvp=1, log2_ctb=4, log2_max_tb=4),
    "b_weighted": dict(b_frames=2, b_ref=1, weighted_pred=1, inter_num_refs=2, inter_bi_pct=70),
    "b_everything": dict(b_frames=2, temporal_mvp=1, weighted_pred=1, mvd_l1_zero=1, amp=1, inter_num_refs=2, b_ref=1, lists_modification=1,
                         num_slices=2, max_merge_cand=4, tile_cols=2, wpp=1),
    "min_cb16_depth2": dict(log2_min_cb=4, log2_ctb=5, max_transform_hierarchy_depth_inter=2, inter_merge_pct=20),
}


@pytest.mark.parametrize("name", sorted(CHROMA_FORMAT_CONFIGS))
@pytest.mark.parametrize("cfi,bit_depth", [(3, 8), (2, 8), (3, 10), (2, 12)])
def test_lossless_round_trip_in_422_and_444(cfi, bit_depth, name):
    """P / B pictures of 4:2:2 and 4:4:4 sequences: the chroma motion vectors of 8.5.3.2.10 (mvLX * 2 / SubWidthC, / SubHeightC: twice the luma
    vector where the chroma plane is not subsampled), chroma prediction blocks of nPbW / SubWidthC x nPbH / SubHeightC, the chroma transform trees
    of those formats inside inter coded units - lossless, so every decoded sample equals the source"""
    frames = make_frames(104, 72, 5, bit_depth, chroma_format_idc=cfi)
    assert frames[0][1].shape == (72 if cfi != 1 else 36, 104 if cfi == 3 else 52)
    aus = orc.encode_sequence(frames, bit_depth=bit_depth, qp=30, global_mv_x=-6, global_mv_y=-3, inter_skip_pct=0, lossless_pct=100, seed=13, **CHROMA_FORMAT_CONFIGS[name])
    pics = orc.decode_sequence(aus, taps=True)
    assert sorted(p["poc"] for p in pics) == list(range(5))
    for p in pics:
        assert p["chroma_format_idc"] == cfi
        for c in range(3):
            np.testing.assert_array_equal(p["planes"][c], frames[p["poc"]][c], err_msg="%s %d: POC %d component %d" % (name, cfi, p["poc"], c))
    assert all((p["map_pred"] == 1).mean() > 0.4 for p in pics if p["poc"] > 0)


@pytest.mark.parametrize("cfi", [2, 3])
def test_lossy_422_444_sequences_do_not_drift(cfi):
    frames = make_frames(136, 104, 6, chroma_format_idc=cfi)
    kw = dict(b_frames=1, temporal_mvp=1, inter_num_refs=2, amp=1, sao=1)
    for p in orc.decode_sequence(orc.encode_sequence(frames, qp=4, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, **kw)):
        for c in range(3):
            assert float(np.mean((p["planes"][c].astype(np.float64) - frames[p["poc"]][c]) ** 2)) < 5.0, (cfi, p["poc"], c)   # (4:2:0 with these parameters: up to 3.3 - quantisation noise of the random QP deltas, not drift)
    pics = orc.decode_sequence(orc.encode_sequence(frames, qp=26, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=25, **kw), taps=True)
    assert any((p["map_pred"] == 2).any() for p in pics)


def test_lossy_b_sequence_no_drift_and_skips():
    frames = make_frames(200, 136, 7)
    kw = dict(b_frames=2, b_ref=1, temporal_mvp=1, weighted_pred=0, inter_num_refs=2, amp=1, sao=1)
    for i, p in enumerate(orc.decode_sequence(orc.encode_sequence(frames, qp=4, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, **kw))):
        assert float(np.mean((p["planes"][0].astype(np.float64) - frames[p["poc"]][0]) ** 2)) < 3.0, (i, p["poc"])
    pics = orc.decode_sequence(orc.encode_sequence(frames, qp=24, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=25, **kw), taps=True)
    for p in pics:
        assert float(np.mean((p["planes"][0].astype(np.float64) - frames[p["poc"]][0]) ** 2)) < 900.0, p["poc"]
    assert any((p["map_pred"] == 2).any() for p in pics if p["poc"] in (1, 2))   # skipped units in B pictures (merge candidates may be bi-predictive)


def test_weighted_prediction_changes_the_prediction():
    """the explicit weights really are applied: the same sequence decoded with the pred_weight_table bits of the stream, against the same motion with
    default weights, differs - and the lossless round trip above is exact with them"""
    frames = make_frames(136, 104, 3)
    a = orc.decode_sequence(orc.encode_sequence(frames, qp=26, inter_skip_pct=0, inter_intra_pct=0, weighted_pred=1, seed=3))
    b = orc.decode_sequence(orc.encode_sequence(frames, qp=26, inter_skip_pct=0, inter_intra_pct=0, weighted_pred=0, seed=3))
    assert any((x["planes"][0] != y["planes"][0]).any() for x, y in zip(a[1:], b[1:]))


def test_single_picture_api_still_refuses_p_slices():
    frames = make_frames(64, 64, 2)
    aus = orc.encode_sequence(frames, qp=30)
    orc.decode(aus[0])                                          # the IDR picture alone is an intra still
    with pytest.raises(orc.OracleError):
        orc.decode(aus[0][:0] + aus[1])                       # a P picture without its parameter sets / outside a sequence
    q = orc.SeqDecoder()
    q.decode(aus[0])
    q.decode(aus[1])
    q.close()


def test_missing_reference_picture_is_an_error():
    frames = make_frames(64, 64, 3)
    aus = orc.encode_sequence(frames, qp=30)
    q = orc.SeqDecoder()
    q.decode(aus[0])
    with pytest.raises(orc.OracleError):
        q.decode(aus[2])                                        # its RPS names POC 1, which was never decoded
    q.close()


@pytest.mark.parametrize("cip", [0, 1])
def test_lossless_round_trip_with_constrained_intra_pred(cip):
    """oracle + generator agree on constrained_intra_pred_flag in P / B pictures: lossless coded pictures come back exactly (the intra blocks
    predict from intra coded neighbours only on both sides), and the flag changes the bitstream"""
    frames = make_frames(136, 104, 4)
    aus = orc.encode_sequence(frames, qp=30, global_mv_x=-8, global_mv_y=-4, lossless_pct=100, inter_skip_pct=0, inter_intra_pct=40, constrained_intra_pred=cip, b_frames=1, temporal_mvp=1)
    pics = sorted(orc.decode_sequence(aus, taps=True), key=lambda p: p["poc"])
    for i, p in enumerate(pics):
        for c in range(3):
            np.testing.assert_array_equal(p["planes"][c], frames[i][c], err_msg="picture %d component %d" % (i, c))
    assert all(0.1 < (p["map_pred"] == 0).mean() < 0.9 for p in pics[1:])      # intra and inter units side by side


@pytest.mark.parametrize("lt", [1, 2, 3])
def test_lossless_round_trip_with_a_long_term_reference_picture(lt):
    """oracle + generator agree on the long-term syntax (slice header by LSBs / with the MSB cycle / SPS candidate), the RPS derivation and the lists"""
    frames = make_frames(136, 104, 7)
    for kw in (dict(), dict(b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2), dict(temporal_mvp=1, inter_num_refs=3, lists_modification=1, weighted_pred=1)):
        aus = orc.encode_sequence(frames, qp=30, global_mv_x=-8, global_mv_y=-4, lossless_pct=100, inter_skip_pct=0, long_term_ref=lt, **kw)
        pics = sorted(orc.decode_sequence(aus), key=lambda p: p["poc"])
        for i, p in enumerate(pics):
            for c in range(3):
                np.testing.assert_array_equal(p["planes"][c], frames[i][c], err_msg="%r picture %d component %d" % (kw, i, c))
        assert aus != orc.encode_sequence(frames, qp=30, global_mv_x=-8, global_mv_y=-4, lossless_pct=100, inter_skip_pct=0, long_term_ref=0, **kw)
