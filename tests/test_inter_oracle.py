"""P pictures in the CPU oracle (oracle/hevc_oracle_inter.c) and the test-stream generator (oracle/hevc_testenc_inter.c): SURVEY.md 8 f3,
the samples libheif's sequence tracks push through the decoder plugin (libheif/sequences/track_visual.cc:200-280).

No fixture of the reference holds inter-coded pictures (parity unpinned).  What pins the SYNTAX is the exact round trip: with every coding
unit coded lossless (cu_transquant_bypass) and no skipped units, decoded == source for every sample of every picture - through the
generator's CABAC writer, the oracle's parser, candidate derivation, interpolation and residual path, over the coding-tool matrix."""
import numpy as np
import pytest

from oracle import pyoracle as orc


def shifted(planes, dx, dy):
    return [np.roll(np.roll(p, dy // (1 if i == 0 else 2), 0), dx // (1 if i == 0 else 2), 1) for i, p in enumerate(planes)]


def make_frames(w, h, n, bit_depth=8, mono=False, seed=5):
    f0 = orc.synth_image(w, h, bit_depth, 0 if mono else 1, seed=seed)
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
    assert pics[3]["mf_ref"].max() >= 1                      # the second reference picture is used


def test_no_drift_between_generator_and_decoder():
    """the generator predicts from ITS decoded pictures (deblocked with motion-dependent strength, SAO applied): if the decoder's reference
    pictures differed, the error would accumulate over the sequence; at a fine quantiser it must stay at the quantisation noise"""
    frames = make_frames(200, 136, 6)
    aus = orc.encode_sequence(frames, qp=4, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=0, inter_num_refs=2, amp=1)
    for i, p in enumerate(orc.decode_sequence(aus)):
        assert float(np.mean((p["planes"][0].astype(np.float64) - frames[i][0]) ** 2)) < 3.0, i


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
