"""P pictures through the DEVICE code on the CPU (tests/emu: parse_core.h with the inter syntax, residual / k_motion / k_mc / reconstruction /
deblocking / SAO kernels compiled for the host) against the oracle, picture by picture: motion field, output planes.  SURVEY.md 8 f3."""
import ctypes as C
import os
import numpy as np
import pytest

from oracle import pyoracle as orc
from test_parse_emu import emu
from test_inter_oracle import make_frames, CONFIGS


def _lib():
    L = emu()
    L.emu_seq_new.restype = C.c_void_p
    L.emu_seq_free.argtypes = [C.c_void_p]
    L.emu_seq_create_picture.restype = C.c_void_p
    L.emu_seq_create_picture.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.emu_seq_commit.argtypes = [C.c_void_p, C.c_void_p]
    L.emu_run_parse.argtypes = [C.c_void_p]
    L.emu_run_pipeline.argtypes = [C.c_void_p, C.c_int]
    L.emu_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.emu_out_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.emu_motion.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.emu_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.emu_seq_create_chain.restype = C.c_void_p
    L.emu_seq_create_chain.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.emu_seq_commit_chain.argtypes = [C.c_void_p, C.c_void_p]
    L.emu_run_pipeline_chain.argtypes = [C.c_void_p]
    L.emu_num_items.argtypes = [C.c_void_p]
    L.emu_item_source.argtypes = [C.c_void_p, C.c_int]
    return L


def parameter_sets(au):
    """the VPS / SPS / PPS NAL units of an access unit in plugin framing: what the decoder instance keeps in front of later samples
    (libheif/codecs/decoder.cc:422: only a chunk's first sample carries them)"""
    out, p = b"", 0
    while p + 4 <= len(au):
        n = int.from_bytes(au[p:p + 4], "big")
        if 32 <= ((au[p + 4] >> 1) & 63) <= 34:
            out += au[p:p + 4 + n]
        p += 4 + n
    return out


def _read_picture(L, b, i, want_motion):
    sz = (C.c_int * 5)()
    L.emu_out_size(b, i, sz)
    w, h, cw, ch, es = list(sz)
    dt = np.uint16 if es == 2 else np.uint8
    planes = []
    for c in range(3 if cw else 1):
        a = np.zeros((h, w) if c == 0 else (ch, cw), dt)
        L.emu_plane(b, i, c, a.ctypes.data)
        planes.append(a)
    pic = {"planes": planes}
    if want_motion:
        info = (C.c_int * 7)()
        L.emu_info(b, i, info)           # coded size: the motion field covers the coded picture
        uw, uh = (info[0] + 3) // 4, (info[1] + 3) // 4
        mv, ref, pred = np.zeros((uh, uw, 2, 2), np.int16), np.zeros((uh, uw, 2), np.int8), np.zeros((uh, uw), np.uint8)
        if L.emu_motion(b, i, mv.ctypes.data, ref.ctypes.data, pred.ctypes.data) == 0:
            pic.update(mf_mv=mv, mf_ref=ref, map_pred=pred)
    return pic


def decode_sequence_emu(aus, want_motion=False, chain=0):
    """chain = 0: one launch set per sample (the instance's state committed after every picture); chain = K: the first sample alone, then K samples
    per launch set the way the decoder's look-ahead runs them (parse + residual over all K, the pixel stages picture by picture)"""
    L = _lib()
    ps = parameter_sets(aus[0])
    aus = [aus[0]] + [ps + a for a in aus[1:]]
    q = C.c_void_p(L.emu_seq_new())
    out = []
    try:
        k = 0
        while k < len(aus):
            err = C.create_string_buffer(512)
            if chain and k > 0:
                group = aus[k:k + chain]
                ptrs = (C.c_char_p * len(group))(*group)
                sizes = (C.c_size_t * len(group))(*[len(a) for a in group])
                b = L.emu_seq_create_chain(q, len(group), ptrs, sizes, err, 512)
                assert b, err.value.decode()
                b = C.c_void_p(b)
                n = L.emu_num_items(b)
                if n:
                    assert L.emu_run_parse(b) == 0, "parse status"
                    assert L.emu_run_pipeline_chain(b) == 0, "pipeline status"
                assert [L.emu_item_source(b, i) for i in range(n)] == list(range(len(group)))      # (no RASL picture is dropped in these streams)
                for i in range(n):
                    out.append(_read_picture(L, b, i, want_motion))
                assert L.emu_seq_commit_chain(q, b) == 0
                k += len(group)
                continue
            au = aus[k]
            b = L.emu_seq_create_picture(q, au, len(au), err, 512)
            assert b, err.value.decode()
            b = C.c_void_p(b)
            assert L.emu_run_parse(b) == 0, "parse status"
            assert L.emu_run_pipeline(b, 15) == 0, "pipeline status"
            out.append(_read_picture(L, b, 0, want_motion))
            assert L.emu_seq_commit(q, b) == 0
            k += 1
    finally:
        L.emu_seq_free(q)
    return out


def check_sequence(aus, name="", chain=0):
    ref = orc.decode_sequence(aus, taps=True)
    got = decode_sequence_emu(aus, want_motion=True, chain=chain)
    assert len(got) == len(ref)
    for i, (r, g) in enumerate(zip(ref, got)):
        if "map_pred" in g:      # a P picture: the motion field first (where the units are inter coded)
            uh, uw = g["map_pred"].shape
            rp = r["map_pred"][:uh, :uw]
            np.testing.assert_array_equal(g["map_pred"], rp, err_msg="%s picture %d: prediction modes" % (name, i))
            inter = rp > 0
            np.testing.assert_array_equal(g["mf_ref"][inter], r["mf_ref"][:uh, :uw][inter], err_msg="%s picture %d: reference indices" % (name, i))
            np.testing.assert_array_equal(g["mf_mv"][inter], r["mf_mv"][:uh, :uw][inter], err_msg="%s picture %d: motion vectors" % (name, i))
        for c in range(len(r["planes"])):
            np.testing.assert_array_equal(g["planes"][c], r["planes"][c], err_msg="%s picture %d plane %d" % (name, i, c))


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_emulated_device_pipeline_decodes_p_pictures(name):
    frames = make_frames(136, 104, 4)
    aus = orc.encode_sequence(frames, qp=26, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=20, seed=21, **CONFIGS[name])
    check_sequence(aus, name)


@pytest.mark.parametrize("kw", [dict(bit_depth=10, amp=1, inter_num_refs=2), dict(lossless_pct=30, amp=1, transform_skip=1, log2_ctb=5),
                                dict(pcm_pct=10, inter_intra_pct=40, cu_qp_delta=1, diff_cu_qp_delta_depth=2, deblock_disable=0, tc_offset_div2=2, beta_offset_div2=-2),
                                dict(deblock_disable=1, sao=0, inter_merge_pct=90), dict(dependent_segments=3, num_slices=2, inter_num_refs=3, wpp=0)],
                         ids=["main10", "lossless_tskip", "pcm_qpdelta_offsets", "no_filters", "dependent_segments"])
def test_emulated_device_pipeline_tool_mix(kw):
    bd = kw.get("bit_depth", 8)
    frames = make_frames(120, 88, 3, bd)
    aus = orc.encode_sequence(frames, qp=24, global_mv_x=6, global_mv_y=-10, seed=5, **kw)
    check_sequence(aus, str(kw))


def test_emulated_monochrome_and_cropped_sequence():
    frames = make_frames(70, 42, 3, 8, mono=True)
    check_sequence(orc.encode_sequence(frames, qp=22, inter_num_refs=2), "mono")
    frames = make_frames(70, 42, 4)        # coded 72 x 48: the reference pictures need the rows below the conformance window
    check_sequence(orc.encode_sequence(frames, qp=22, global_mv_x=3, global_mv_y=17, inter_num_refs=2, amp=1), "cropped")


from test_inter_oracle import B_CONFIGS


@pytest.mark.parametrize("name", sorted(B_CONFIGS))
def test_emulated_device_pipeline_decodes_b_tmvp_weighted(name):
    """B pictures (two lists, bi-prediction, combined merge candidates, coding order != POC order), temporal candidates from the collocated
    picture's motion field, explicit weighted prediction: motion field and samples of every picture against the oracle, in coding order"""
    frames = make_frames(136, 104, 7)
    aus = orc.encode_sequence(frames, qp=26, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=20, seed=21, **B_CONFIGS[name])
    check_sequence(aus, name)


@pytest.mark.parametrize("kw", [dict(b_frames=2, b_ref=1, temporal_mvp=1, weighted_pred=1, bit_depth=10, amp=1, inter_num_refs=2),
                                dict(b_frames=1, temporal_mvp=1, lossless_pct=30, transform_skip=1, log2_ctb=5, mvd_l1_zero=1),
                                dict(b_frames=3, b_ref=1, temporal_mvp=1, weighted_pred=1, dependent_segments=3, num_slices=2, inter_num_refs=3, wpp=0, max_merge_cand=2),
                                dict(b_frames=2, temporal_mvp=1, deblock_disable=0, sao=1, inter_merge_pct=90, inter_bi_pct=90, cu_qp_delta=1)],
                         ids=["main10", "lossless_tskip", "dependent_segments", "merge_heavy"])
def test_emulated_b_tool_mix(kw):
    bd = kw.get("bit_depth", 8)
    frames = make_frames(120, 88, 6, bd)
    aus = orc.encode_sequence(frames, qp=24, global_mv_x=6, global_mv_y=-10, seed=5, **kw)
    check_sequence(aus, str(kw))


def test_emulated_b_monochrome_and_cropped():
    frames = make_frames(70, 42, 5, 8, mono=True)
    check_sequence(orc.encode_sequence(frames, qp=22, inter_num_refs=2, b_frames=1, temporal_mvp=1, weighted_pred=1), "mono")
    frames = make_frames(70, 42, 6)        # coded 72 x 48: references and collocated motion cover the rows below the conformance window
    check_sequence(orc.encode_sequence(frames, qp=22, global_mv_x=3, global_mv_y=17, inter_num_refs=2, amp=1, b_frames=2, b_ref=1, temporal_mvp=1), "cropped")


def test_reference_of_another_format_is_refused_by_the_host():
    """parameter sets that change without an IDR picture: a P picture whose reference was decoded at another size must not reach the kernels (they address
    references with the current picture's geometry)"""
    L = _lib()
    a = orc.encode_sequence(make_frames(136, 104, 2), qp=26)
    b = orc.encode_sequence(make_frames(72, 56, 2), qp=26)
    q = C.c_void_p(L.emu_seq_new())
    try:
        err = C.create_string_buffer(512)
        pic = C.c_void_p(L.emu_seq_create_picture(q, a[0], len(a[0]), err, 512))
        assert pic and L.emu_run_parse(pic) == 0 and L.emu_run_pipeline(pic, 15) == 0 and L.emu_seq_commit(q, pic) == 0
        au = parameter_sets(b[0]) + b[1]                      # the other stream's parameter sets in front of its P picture: POC 1 references POC 0
        assert not L.emu_seq_create_picture(q, au, len(au), err, 512)
        assert b"another format" in err.value
    finally:
        L.emu_seq_free(q)


# ---- chains: the decoder's look-ahead (HIPDEC_SEQ_LOOKAHEAD) parses K samples of a track in ONE launch set; references inside the set ----------
@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("chain", [2, 8])
def test_emulated_chain_p_pictures(name, chain):
    frames = make_frames(136, 104, 6)
    aus = orc.encode_sequence(frames, qp=26, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=20, seed=21, **CONFIGS[name])
    check_sequence(aus, name, chain=chain)


@pytest.mark.parametrize("name", sorted(B_CONFIGS))
@pytest.mark.parametrize("chain", [3, 16])
def test_emulated_chain_b_tmvp_weighted(name, chain):
    """coding order != output order inside a chain; the collocated picture and both lists' references are earlier items of the same launch set"""
    frames = make_frames(136, 104, 8)
    aus = orc.encode_sequence(frames, qp=26, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=20, seed=21, **B_CONFIGS[name])
    check_sequence(aus, name, chain=chain)


def test_emulated_chain_main10_mono_cropped_and_idr_inside():
    frames = make_frames(120, 88, 6, 10)
    check_sequence(orc.encode_sequence(frames, qp=24, global_mv_x=6, global_mv_y=-10, seed=5, bit_depth=10, amp=1, inter_num_refs=2, b_frames=2, b_ref=1,
                                       temporal_mvp=1, weighted_pred=1), "main10", chain=4)
    frames = make_frames(70, 42, 5, 8, mono=True)
    check_sequence(orc.encode_sequence(frames, qp=22, inter_num_refs=2, b_frames=1, temporal_mvp=1, weighted_pred=1), "mono", chain=3)
    frames = make_frames(70, 42, 7)        # coded 72 x 48: the in-batch references are the uncropped copies of the second SAO pass
    check_sequence(orc.encode_sequence(frames, qp=22, global_mv_x=3, global_mv_y=17, inter_num_refs=2, amp=1, b_frames=2, b_ref=1, temporal_mvp=1), "cropped", chain=4)
    # two coded video sequences back to back: the IDR picture of the second one sits inside a chain
    a = orc.encode_sequence(make_frames(136, 104, 4), qp=26, inter_num_refs=2)
    b = orc.encode_sequence(make_frames(136, 104, 4, seed=9), qp=26, inter_num_refs=2)
    both = a + [parameter_sets(b[0]) + x if i else x for i, x in enumerate(b)]
    ref = orc.decode_sequence(a) + orc.decode_sequence(b)
    got = decode_sequence_emu(both, chain=5)
    assert len(got) == len(ref)
    for i, (r, g) in enumerate(zip(ref, got)):
        for c in range(3):
            np.testing.assert_array_equal(g["planes"][c], r["planes"][c], err_msg="two sequences: picture %d plane %d" % (i, c))


def test_chain_steps_follow_the_dependencies():
    """pixel steps break where a picture predicts from an earlier picture of the run, motion steps only where the COLLOCATED picture is in the run:
    an intra-only chain is one step (a plain batch), IPPP without temporal candidates is one motion step, non-reference B pictures share a step"""
    L = _lib()
    L.emu_chain_steps.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def steps(aus, first_alone=True):
        ps = parameter_sets(aus[0])
        q = C.c_void_p(L.emu_seq_new())
        try:
            err = C.create_string_buffer(512)
            b = C.c_void_p(L.emu_seq_create_picture(q, aus[0], len(aus[0]), err, 512))
            assert L.emu_run_parse(b) == 0 and L.emu_run_pipeline(b, 15) == 0 and L.emu_seq_commit(q, b) == 0
            group = [ps + a for a in aus[1:]]
            ptrs = (C.c_char_p * len(group))(*group)
            sizes = (C.c_size_t * len(group))(*[len(a) for a in group])
            b = L.emu_seq_create_chain(q, len(group), ptrs, sizes, err, 512)
            assert b, err.value.decode()
            px, mo = C.c_int(), C.c_int()
            assert L.emu_chain_steps(C.c_void_p(b), C.byref(px), C.byref(mo)) == 0
            return px.value, mo.value
        finally:
            L.emu_seq_free(q)

    frames = make_frames(72, 56, 7)
    intra = [orc.encode(f, qp=30) for f in frames]
    assert steps([intra[0]] + [b"".join(x for x in _split(a) if (x[4] >> 1) & 63 < 32) for a in intra[1:]]) == (1, 1)
    assert steps(orc.encode_sequence(frames, qp=30, temporal_mvp=0)) == (6, 1)          # every P picture predicts from the one before; no collocated picture
    assert steps(orc.encode_sequence(frames, qp=30, temporal_mvp=1)) == (6, 6)
    px, mo = steps(orc.encode_sequence(frames, qp=30, b_frames=2, b_ref=0, temporal_mvp=0))
    assert px < 6 and mo == 1                                                          # the two B pictures between anchors predict from the anchors only


def _split(stream):
    out, p = [], 0
    while p + 4 <= len(stream):
        n = int.from_bytes(stream[p:p + 4], "big")
        out.append(stream[p:p + 4 + n])
        p += 4 + n
    return out


@pytest.mark.parametrize("sl", [1, 2, 3], ids=["default_lists", "sps_lists", "pps_lists"])
def test_emulated_p_b_pictures_with_scaling_lists(sl):
    """scaling_list_enabled_flag in P / B pictures (missing 4): blocks of inter coded units dequantise with the matrices of
    matrixId 3 .. 5 (Table 7-4; the 32x32 inter luma matrix), intra units of the same picture with 0 .. 2 - default lists (Table 7-6 differs
    between the two), lists coded in the SPS, lists coded in the PPS"""
    frames = make_frames(136, 104, 5)
    aus = orc.encode_sequence(frames, qp=24, global_mv_x=-8, global_mv_y=-4, scaling_list=sl, b_frames=1, temporal_mvp=1, inter_intra_pct=30, seed=3 + sl)
    check_sequence(aus, "scaling_list=%d" % sl)
    check_sequence(aus, "scaling_list=%d, chain" % sl, chain=4)


@pytest.mark.parametrize("kw", [dict(), dict(b_frames=2, b_ref=1, temporal_mvp=1, log2_ctb=4, log2_max_tb=4), dict(log2_ctb=5, amp=1, inter_num_refs=2, num_slices=2, wpp=0),
                                dict(tile_cols=2, tile_rows=2, log2_ctb=4, log2_max_tb=4)], ids=["ctb64", "b_ctb16", "slices_ctb32", "tiles_ctb16"])
def test_emulated_constrained_intra_pred_in_p_b_pictures(kw):
    """constrained_intra_pred_flag with P / B slices (missing 4): samples of units that are not intra coded are "not available for
    intra prediction" (8.4.4.2.2) - inside the CTB (inter units never enter the availability map) and across CTB borders (the neighbouring CTBs'
    units are looked up in the prediction-mode map); small CTBs put most neighbours in another CTB"""
    aus = orc.encode_sequence(make_frames(136, 104, 5), qp=24, global_mv_x=-8, global_mv_y=-4, constrained_intra_pred=1, inter_intra_pct=45, inter_skip_pct=10, **kw)
    check_sequence(aus, "constrained_intra_pred %r" % kw)
    check_sequence(aus, "constrained_intra_pred, chain", chain=4)
    # the flag matters in these streams: the same pictures coded without it give another bitstream (another prediction for the intra blocks)
    assert aus != orc.encode_sequence(make_frames(136, 104, 5), qp=24, global_mv_x=-8, global_mv_y=-4, constrained_intra_pred=0, inter_intra_pct=45, inter_skip_pct=10, **kw)


@pytest.mark.parametrize("lt", [1, 2, 3], ids=["slice_lsb", "slice_msb_present", "sps_candidate"])
@pytest.mark.parametrize("kw", [dict(), dict(b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2),
                                dict(temporal_mvp=1, inter_num_refs=3, lists_modification=1, weighted_pred=1, amp=1)], ids=["p", "b_tmvp", "p_multiref_listmod_weighted"])
def test_emulated_long_term_reference_pictures(lt, kw):
    """Long-term reference pictures (missing 4): the IDR picture stays in the DPB as a long-term reference of every later picture, named
    in the slice header by its POC LSBs, by LSBs + delta_poc_msb_cycle_lt, or through a candidate of the SPS (7.3.6.1, 8.3.2); it sits behind the
    short-term pictures in both lists (8.3.4); a motion vector is only predicted between two long-term or two short-term references and never
    scaled towards a long-term one (8.5.3.2.7), the collocated candidate compares the marking its block's reference had when ITS picture was decoded
    (8.5.3.2.9: two bits per unit of the stored motion field)"""
    frames = make_frames(136, 104, 7)
    aus = orc.encode_sequence(frames, qp=26, global_mv_x=-8, global_mv_y=-4, long_term_ref=lt, **kw)
    check_sequence(aus, "long_term_ref=%d %r" % (lt, kw))
    check_sequence(aus, "long_term_ref=%d %r, chain" % (lt, kw), chain=4)


# ---- several tracks' chains in ONE launch set (batch_layout.h: layout_batch_plan_chains; the product's chain coalescer in decoder.hip) ---------------
def decode_tracks_emu(tracks, chains):
    """tracks: access-unit lists; chains[t]: samples of track t per shared launch set.  Every track's first sample is decoded alone, then round after
    round the next chains[t] samples of every track that still has some go into ONE batch: items ordered by (pixel step, track), the RowDesc table
    by (motion step, item); step k of the batch = step k of every track."""
    L = _lib()
    L.emu_seq_create_chains.restype = C.c_void_p
    L.emu_seq_create_chains.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                                        C.c_char_p, C.c_size_t]
    L.emu_seq_commit_chains.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    L.emu_track_items.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.emu_chain_steps.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.emu_free.argtypes = [C.c_void_p]
    T = len(tracks)
    tracks = [[aus[0]] + [parameter_sets(aus[0]) + a for a in aus[1:]] for aus in tracks]
    qs = [C.c_void_p(L.emu_seq_new()) for _ in range(T)]
    out = [[] for _ in range(T)]
    shared, step_counts = [], []
    try:
        err = C.create_string_buffer(512)
        for t in range(T):
            au = tracks[t][0]
            b = L.emu_seq_create_picture(qs[t], au, len(au), err, 512)
            assert b, err.value.decode()
            b = C.c_void_p(b)
            assert L.emu_run_parse(b) == 0 and L.emu_run_pipeline(b, 15) == 0
            out[t].append(_read_picture(L, b, 0, True))
            assert L.emu_seq_commit(qs[t], b) == 0
        pos = [1] * T
        while any(pos[t] < len(tracks[t]) for t in range(T)):
            live = [t for t in range(T) if pos[t] < len(tracks[t])]
            data, first, count = [], [], []
            for t in live:
                group = tracks[t][pos[t]:pos[t] + chains[t]]
                first.append(len(data)); count.append(len(group)); data += group
            n = len(live)
            qarr = (C.c_void_p * n)(*[qs[t] for t in live])
            b = L.emu_seq_create_chains(qarr, n, (C.c_int * n)(*first), (C.c_int * n)(*count), (C.c_char_p * len(data))(*data),
                                        (C.c_size_t * len(data))(*[len(a) for a in data]), err, 512)
            assert b, err.value.decode()
            b = C.c_void_p(b)
            shared.append(b)
            assert L.emu_num_items(b) == len(data)
            assert L.emu_run_parse(b) == 0, "parse status"
            assert L.emu_run_pipeline_chain(b) == 0, "pipeline status"
            px, mo = C.c_int(), C.c_int()
            L.emu_chain_steps(b, C.byref(px), C.byref(mo))
            step_counts.append((px.value, mo.value))
            for k, t in enumerate(live):
                items, samples = (C.c_int * 64)(), (C.c_int * 64)()
                m = L.emu_track_items(b, k, items, samples, 64)
                assert m == count[k] and list(samples[:m]) == list(range(m))
                for i in items[:m]:
                    out[t].append(_read_picture(L, b, i, True))
                pos[t] += count[k]
            assert L.emu_seq_commit_chains(qarr, n, b) == 0
    finally:
        for q in qs:
            L.emu_seq_free(q)
        for b in shared:
            L.emu_free(b)
    return out, step_counts


def _check_tracks(tracks, chains, names):
    got, step_counts = decode_tracks_emu(tracks, chains)
    for t, aus in enumerate(tracks):
        ref = orc.decode_sequence(aus, taps=True)
        assert len(got[t]) == len(ref)
        for i, (r, g) in enumerate(zip(ref, got[t])):
            if "map_pred" in g:
                uh, uw = g["map_pred"].shape
                inter = r["map_pred"][:uh, :uw] > 0
                np.testing.assert_array_equal(g["mf_mv"][inter], r["mf_mv"][:uh, :uw][inter], err_msg="%s picture %d: motion vectors" % (names[t], i))
            for c in range(len(r["planes"])):
                np.testing.assert_array_equal(g["planes"][c], r["planes"][c], err_msg="%s picture %d plane %d" % (names[t], i, c))
    return step_counts


def test_emulated_chains_of_several_tracks_share_one_launch_set():
    """four tracks with different structures, picture sizes and chain lengths side by side: low-delay P with two references and temporal candidates,
    B pictures with a reference B, an intra-only track, weighted P with CTB 32 - every picture of every track equals the oracle's"""
    specs = [("ippp_tmvp", (136, 104), 7, dict(inter_num_refs=2, temporal_mvp=1), 3),
             ("ibbp", (72, 56), 8, dict(b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2), 4),
             ("weighted_ctb32", (104, 72), 6, dict(weighted_pred=1, log2_ctb=5, amp=1), 2),
             ("ippp_no_tmvp", (136, 104), 5, dict(temporal_mvp=0), 6)]
    tracks, chains, names = [], [], []
    for k, (name, (w, h), n, kw, chain) in enumerate(specs):
        frames = make_frames(w, h, n)
        tracks.append(orc.encode_sequence(frames, qp=27, global_mv_x=-8, global_mv_y=4, inter_skip_pct=20, seed=40 + k, **kw))
        chains.append(chain); names.append(name)
    intra = [orc.encode(f, qp=30) for f in make_frames(72, 56, 4)]
    tracks.append([intra[0]] + [b"".join(x for x in _split(a) if (x[4] >> 1) & 63 < 32) for a in intra[1:]])
    chains.append(3); names.append("intra_only")
    step_counts = _check_tracks(tracks, chains, names)
    # the batch has as many steps as its deepest track: the B track's first four samples (P B B P: the second anchor predicts from the first) need more
    # than one pixel step, the three P samples of the first track three
    assert step_counts[0][0] >= 3 and step_counts[0][1] >= 2


def test_emulated_chains_of_equal_tracks_keep_the_step_count_of_one():
    """8 copies of one IPPP track in one launch set have the steps of one track: step k holds the k-th picture of all of them"""
    frames = make_frames(72, 56, 6)
    aus = orc.encode_sequence(frames, qp=30, temporal_mvp=1, inter_num_refs=2, seed=3)
    step_counts = _check_tracks([aus] * 8, [5] * 8, ["copy%d" % i for i in range(8)])
    assert step_counts == _check_tracks([aus], [5], ["alone"]) and step_counts[0][0] == 5


def test_emulated_chains_of_tracks_with_a_cropped_a_monochrome_and_a_restarting_track():
    """one launch set with: a cropped track (its in-batch references are the uncropped copies of the second SAO pass, whose slots move with the item
    order), a monochrome track (no chroma waves in the shared ReconWave table), and a track whose second coded video sequence starts INSIDE the chain
    (an IDR picture in the middle: PicOrderCnt starts over, the pictures before it stop being references)"""
    cropped = orc.encode_sequence(make_frames(70, 42, 7), qp=22, global_mv_x=3, global_mv_y=17, inter_num_refs=2, amp=1, b_frames=2, b_ref=1, temporal_mvp=1, seed=4)
    mono = orc.encode_sequence(make_frames(72, 56, 6, 8, mono=True), qp=24, inter_num_refs=2, temporal_mvp=1, weighted_pred=1, seed=5)
    a = orc.encode_sequence(make_frames(136, 104, 4), qp=26, inter_num_refs=2, temporal_mvp=1)
    b = orc.encode_sequence(make_frames(136, 104, 4, seed=9), qp=26, inter_num_refs=2, temporal_mvp=1)
    both = a + [parameter_sets(b[0]) + x if i else x for i, x in enumerate(b)]
    got, _ = decode_tracks_emu([cropped, mono, both], [4, 5, 7])
    for t, (name, refs) in enumerate((("cropped", orc.decode_sequence(cropped)), ("mono", orc.decode_sequence(mono)),
                                      ("two sequences", orc.decode_sequence(a) + orc.decode_sequence(b)))):
        assert len(got[t]) == len(refs)
        for i, (r, g) in enumerate(zip(refs, got[t])):
            for c in range(len(r["planes"])):
                np.testing.assert_array_equal(g["planes"][c], r["planes"][c], err_msg="%s: picture %d plane %d" % (name, i, c))


# ---- open GOP: a CRA picture with RASL pictures (oracle/hevc_testenc.h: open_gop) ---------------------------------------------------------------------
def _open_gop_track():
    frames = make_frames(136, 104, 10)
    return orc.encode_sequence(frames, qp=27, b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, open_gop=2, seed=3)


def _slice_types(au):
    return [(x[4] >> 1) & 63 for x in _split(au) if (x[4] >> 1) & 63 < 32]


@pytest.mark.parametrize("chain", [0, 16])
def test_emulated_open_gop_track_from_its_idr_picture(chain):
    """IDR P B B CRA RASL RASL P B B: decoded from the start, the CRA picture is an ordinary intra picture with a reference picture set and its RASL
    pictures are ordinary B pictures (NoRaslOutputFlag = 0): every picture equals the oracle's"""
    aus = _open_gop_track()
    assert [t for a in aus for t in _slice_types(a)] == [19, 1, 1, 1, 21, 9, 9, 1, 1, 1]
    check_sequence(aus, "open gop", chain=chain)


def test_emulated_track_that_starts_at_a_cra_picture_drops_its_rasl_pictures():
    """the same track entered at the CRA picture (a seek; libheif pushes from a sync sample): NoRaslOutputFlag = 1, the two RASL pictures reference a
    picture that was never decoded and are dropped (8.3.3) - they are no items of the chain -, everything else decodes exactly as in the full track:
    the CRA picture's PicOrderCnt comes from its LSBs alone, the pictures behind it find their references by PicOrderCnt"""
    L = _lib()
    aus = _open_gop_track()
    full = {p["poc"]: p for p in orc.decode_sequence(aus)}
    ps = parameter_sets(aus[0])
    entered = [ps + a for a in aus[4:]]          # CRA, RASL, RASL, P, B, B
    q = C.c_void_p(L.emu_seq_new())
    try:
        err = C.create_string_buffer(512)
        b = L.emu_seq_create_picture(q, entered[0], len(entered[0]), err, 512)
        assert b, err.value.decode()
        b = C.c_void_p(b)
        assert L.emu_run_parse(b) == 0 and L.emu_run_pipeline(b, 15) == 0
        cra = _read_picture(L, b, 0, False)
        assert L.emu_seq_commit(q, b) == 0
        for c in range(3):
            np.testing.assert_array_equal(cra["planes"][c], full[6]["planes"][c], err_msg="the CRA picture, plane %d" % c)
        # a RASL picture on its own is refused with "no image" (the decoder object then moves on to the next sample) ...
        lone = L.emu_seq_create_picture(q, entered[1], len(entered[1]), err, 512)
        assert not lone and b"RASL" in err.value, err.value
        # ... and inside a chain it simply is no item
        group = entered[1:]
        ptrs = (C.c_char_p * len(group))(*group)
        sizes = (C.c_size_t * len(group))(*[len(a) for a in group])
        b = L.emu_seq_create_chain(q, len(group), ptrs, sizes, err, 512)
        assert b, err.value.decode()
        b = C.c_void_p(b)
        n = L.emu_num_items(b)
        assert [L.emu_item_source(b, i) for i in range(n)] == [2, 3, 4]
        assert L.emu_run_parse(b) == 0 and L.emu_run_pipeline_chain(b) == 0
        for i, poc in enumerate([9, 7, 8]):
            pic = _read_picture(L, b, i, False)
            for c in range(3):
                np.testing.assert_array_equal(pic["planes"][c], full[poc]["planes"][c], err_msg="PicOrderCnt %d plane %d" % (poc, c))
        assert L.emu_seq_commit_chain(q, b) == 0
    finally:
        L.emu_seq_free(q)
