"""GPU parity of the HIP HEVC decode path (through the C ABI) against the CPU oracle: bit-exact
planes on seeded synthetic streams over the coding-tool matrix, plus the reference's real fixtures
where /root/reference is present (never on the GPU box)."""
import os
import numpy as np
import pytest

from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(),
    dict(wpp=0),
    dict(stress=1),
    dict(stress=1, wpp=0, log2_ctb=4, log2_max_tb=4),
    dict(stress=1, log2_ctb=5, log2_max_tb=5),
    dict(tile_cols=2, tile_rows=2, wpp=0),
    dict(tile_cols=3, tile_rows=2, wpp=1, loop_filter_across_tiles=0),
    dict(num_slices=3, loop_filter_across_slices=0),
    dict(num_slices=4, wpp=0, stress=1),
    dict(transform_skip=1, stress=1),
    dict(lossless_pct=30),
    dict(bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16),
    dict(log2_ctb=5, log2_min_cb=4, log2_max_tb=5, max_transform_hierarchy_depth_intra=2, stress=1),
    dict(sao=0, deblock_disable=1),
    dict(cb_qp_offset=3, cr_qp_offset=-4, beta_offset_div2=2, tc_offset_div2=-2, qp=34),
    dict(qp=12, stress=1, zero_residual_pct=30),
    dict(sign_data_hiding=0, cu_qp_delta=0, strong_intra_smoothing=0),
    dict(qp=40),
    dict(pcm_pct=25),
    dict(pcm_pct=30, pcm_loop_filter_disabled=1, stress=1, bit_depth=10),
    dict(pcm_pct=40, log2_ctb=5, log2_max_tb=4, wpp=0, lossless_pct=20),
    dict(dependent_segments=3, wpp=0, stress=1),
    dict(dependent_segments=4, num_slices=2, wpp=0, log2_ctb=4, log2_max_tb=4, loop_filter_across_slices=0),
    dict(dependent_segments=3, wpp=1),
]


def _decode_gpu(stream):
    from libheif_amd.decoder import HipDecoder
    d = HipDecoder()
    d.push_data(stream)
    img = d.decode_next_image()
    assert d.decode_next_image() is None
    d.free()
    return img


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: ",".join("%s=%s" % kv for kv in c.items()) or "default")
@pytest.mark.parametrize("size", [(200, 136), (64, 64)])
def test_decode_matches_oracle(cfg, size):
    bd = cfg.get("bit_depth", 8)
    planes = orc.synth_image(size[0], size[1], bd, 1, seed=3 + size[0])
    stream = orc.encode(planes, **cfg)
    ref = orc.decode(stream)
    img = _decode_gpu(stream)
    assert (img.info["width"], img.info["height"]) == (ref["width"], ref["height"])
    assert img.nclx == ref["nclx"]
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="component %d" % c)


def test_monochrome_and_cropped_sizes():
    for w, h, cf in [(75, 41, 0), (70, 42, 1), (8, 8, 1), (136, 24, 1)]:
        planes = orc.synth_image(w, h, 8, cf, seed=5)
        stream = orc.encode(planes)
        ref = orc.decode(stream)
        img = _decode_gpu(stream)
        assert len(img.planes) == len(ref["planes"])
        for c in range(len(ref["planes"])):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c])


def test_intermediate_maps_match_oracle():
    """the parse kernel's per-unit maps against the oracle's taps (same bit layout)."""
    from libheif_amd.decoder import Batch
    planes = orc.synth_image(200, 136, 8, 1, seed=11)
    stream = orc.encode(planes, stress=1, transform_skip=1, lossless_pct=10)
    ref = orc.decode(stream, taps=True)
    b = Batch([stream])
    b.run(); b.status()
    m = b.maps(0)
    np.testing.assert_array_equal(m["log2_cb"], ref["map_log2_cb"])
    np.testing.assert_array_equal(m["log2_tb"], ref["map_log2_tb"])
    np.testing.assert_array_equal(m["intra_luma"], ref["map_intra_luma"])
    np.testing.assert_array_equal(m["intra_chroma"], ref["map_intra_chroma"])
    np.testing.assert_array_equal(m["qp_y"], ref["map_qp_y"])
    np.testing.assert_array_equal(m["flags"] & 0x7f, ref["map_flags"] & 0x7f)
    for c in range(3):
        np.testing.assert_array_equal(b.tap(0, c), ref["post_deblock"][c])


def test_batch_of_independent_items_matches_single_decodes():
    from libheif_amd.decoder import Batch
    streams = []
    for i in range(6):
        planes = orc.synth_image(128 + 8 * (i % 3), 72 + 8 * (i % 2), 8, 1, seed=100 + i)
        streams.append(orc.encode(planes, wpp=i % 2, stress=i % 3 == 0, tile_cols=1 + i % 2, seed=i))
    b = Batch(streams)
    b.run(); b.status()
    for i, s in enumerate(streams):
        ref = orc.decode(s)
        got = b.planes(i)
        for c in range(3):
            np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))
    t = b.timing_us()
    assert t["total"] > 0


@pytest.mark.parametrize("waves", [1, 2, 3, 5])
def test_fewer_parser_waves_than_substreams(waves, monkeypatch):
    """large batches deal a picture's substreams round-robin to W < n waves (WPP rows r, r+W, ... per wave)"""
    from libheif_amd.decoder import Batch
    monkeypatch.setenv("HIPDEC_WAVES_PER_PICTURE", str(waves))
    monkeypatch.setenv("HIPDEC_RECON_WAVES_PER_PICTURE", str(waves))     # same dealing for the reconstruction wavefronts
    streams = [orc.encode(orc.synth_image(264, 456, 8, 1, seed=70), log2_ctb=5, log2_max_tb=5),            # 15 WPP rows
               orc.encode(orc.synth_image(200, 136, 8, 1, seed=71), tile_cols=3, tile_rows=2, wpp=0),       # 6 tiles
               orc.encode(orc.synth_image(200, 136, 8, 1, seed=72), num_slices=3, wpp=1),
               orc.encode(orc.synth_image(136, 200, 8, 1, seed=73), wpp=0)]
    b = Batch(streams)
    b.run(); b.status()
    for i, s in enumerate(streams):
        ref = orc.decode(s)
        got = b.planes(i)
        for c in range(3):
            np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))


def test_fused_rgb_after_decode_matches_oracle_chain():
    from libheif_amd.decoder import Batch
    planes = orc.synth_image(200, 136, 8, 1, seed=21)
    for vui, expect_int in (((1, 13, 6, 1), True), ((1, 13, 6, 0), False), (None, False)):
        kw = dict(vui_primaries=vui[0], vui_transfer=vui[1], vui_matrix=vui[2], vui_full_range=vui[3]) if vui else {}
        stream = orc.encode(planes, **kw)
        ref = orc.decode(stream)
        b = Batch([stream]); b.run(); b.status()
        rgb = b.to_rgb(0, 10)
        y, cb, cr = ref["planes"]
        if expect_int:
            exp = orc.color_420_to_rgb24(y, cb, cr, ref["nclx"]).reshape(136, -1)
        else:
            r, g, bb = orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, 1, ref["nclx"])
            exp = orc.color_rgb_planar_to_interleaved8(r, g, bb).reshape(136, -1)
        np.testing.assert_array_equal(rgb, exp)


def test_errors_are_loud():
    from libheif_amd.decoder import HipDecoder
    from libheif_amd import HipDecError
    planes = orc.synth_image(64, 64, 8, 1, seed=2)
    stream = orc.encode(planes)
    d = HipDecoder()
    with pytest.raises(HipDecError) as e:
        d.push_data(stream[:len(stream) - 5])       # truncated framing -> End_of_data
    assert e.value.code == -2
    assert d.decode_next_image() is None            # nothing pushed -> no image
    d = HipDecoder()
    nals, p = [], 0
    while p < len(stream):
        n = int.from_bytes(stream[p:p + 4], "big"); nals.append(stream[p:p + 4 + n]); p += 4 + n
    # two coded pictures in one push (libde265 takes any number of NAL units per push, decoder_libde265.cc:322-368): the front end splits them
    # into access units; the first is the still, the second follows as a sample of a sequence - one picture per decode call, then nothing
    d.push_data(stream + b"".join(x for x in nals if (x[4] >> 1) & 63 < 32))
    ref = orc.decode(stream)
    for _ in range(2):
        img = d.decode_next_image()
        assert img is not None
        for c in range(3):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c])
    assert d.decode_next_image() is None
    d = HipDecoder(max_image_size_pixels=1000)      # security limit before any allocation
    d.push_data(stream)
    with pytest.raises(HipDecError) as e:
        d.decode_next_image()
    assert e.value.code == -5
    # corrupt slice data: the device must report a desynchronised substream, not hang or crash
    bad = bytearray(stream)
    for k in range(len(bad) - 60, len(bad) - 20):
        bad[k] ^= 0xA5
    d = HipDecoder()
    d.push_data(bytes(bad))
    try:
        d.decode_next_image()
    except HipDecError as ex:
        assert ex.code in (-8, -3)


def test_reference_fixtures_match_oracle():
    """the reference's x265-coded items (committed plugin-framed streams, tests/golden/ref_*.hevc — the GPU box has no
    /root/reference) against the oracle run live: planes AND the parser's unit maps"""
    import glob
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    paths = [p for p in sorted(glob.glob(os.path.join(gold, "ref_*.hevc"))) if "ref_reject_" not in p]
    assert len(paths) >= 14
    for p in paths:
        s = open(p, "rb").read()
        ref = orc.decode(s)
        img = _decode_gpu(s)
        assert len(img.planes) == len(ref["planes"])
        for c in range(len(ref["planes"])):
            np.testing.assert_array_equal(img.planes[c], ref["planes"][c], err_msg="%s component %d" % (os.path.basename(p), c))


@pytest.mark.parametrize("size,cfg", [((3840, 2160), dict()), ((1920, 1080), dict(tile_cols=4, tile_rows=2, wpp=0)), ((1024, 1024), dict(stress=1))],
                         ids=["4k_wpp", "1080p_tiles", "grid_tile_stress"])
def test_full_size_stills_match_oracle(size, cfg):
    """BASELINE.json's full sizes (config 1 still, config 5 still, config 3 tile): bit-exact planes and the
    fused RGB of the whole picture; a batch of replicas must reproduce the same planes (idempotence)."""
    import hashlib
    from libheif_amd.decoder import Batch
    from tools import streamgen
    stream = streamgen.make_stream(size[0], size[1], 1, 8, **cfg)
    ref = orc.decode(stream)
    b = Batch([stream, stream, stream])
    b.run(); b.status()
    b.run(); b.status()      # a second run over the same arena (control words are re-zeroed every run)
    for i in range(3):
        got = b.planes(i)
        for c in range(3):
            np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))
    assert b.info(0)["num_substreams"] == ref["n_substreams"]


def test_corrupt_streams_never_hang_or_crash():
    """random corruption of the slice data (and of the headers): every decode returns — success with some
    picture, or a loud error — and the device stays usable (all device-side loops and waits are bounded)."""
    from libheif_amd.decoder import HipDecoder
    from libheif_amd import HipDecError
    rng = np.random.default_rng(123)
    base = orc.encode(orc.synth_image(200, 136, 8, 1, seed=31), stress=1)
    base_t = orc.encode(orc.synth_image(200, 136, 8, 1, seed=32), tile_cols=2, tile_rows=2, wpp=0)
    outcomes = {"ok": 0, "error": 0}
    for trial in range(24):
        s = bytearray(base if trial % 2 == 0 else base_t)
        lo = 0 if trial % 6 == 5 else 120        # mostly slice data, sometimes parameter sets / slice header too
        for _ in range(int(rng.integers(1, 12))):
            s[int(rng.integers(lo, len(s)))] = int(rng.integers(0, 256))
        d = HipDecoder()
        try:
            d.push_data(bytes(s))
            d.decode_next_image()
            outcomes["ok"] += 1
        except HipDecError as e:
            assert e.code in (-2, -3, -4, -5, -7, -8), e
            outcomes["error"] += 1
        d.free()
    assert outcomes["error"] > 0
    # the device still decodes correctly afterwards
    ref = orc.decode(base)
    d = HipDecoder(); d.push_data(base); img = d.decode_next_image(); d.free()
    for c in range(3):
        np.testing.assert_array_equal(img.planes[c], ref["planes"][c])


def test_pool_scheduler_batch_and_error_path(monkeypatch):
    """throughput-mode scheduling on the device (rows as pool tasks with suspend / resume): a mixed batch decodes
    bit-exactly, a corrupt member fails the batch loudly without hanging, and the next batch is unaffected."""
    from libheif_amd.decoder import Batch
    from libheif_amd import HipDecError
    monkeypatch.setenv("HIPDEC_PARSE_POOL", "1")
    monkeypatch.setenv("HIPDEC_POOL_WAVES", "48")          # fewer waves than rows: rows queue up and get parked
    cfgs = [dict(), dict(log2_ctb=4, log2_max_tb=4, stress=1), dict(tile_cols=2, tile_rows=2, wpp=1), dict(num_slices=3), dict(wpp=0)]
    streams = [orc.encode(orc.synth_image(264 + 8 * (i % 4), 200 + 8 * (i % 3), 8, 1, seed=300 + i), **cfgs[i % len(cfgs)]) for i in range(24)]
    b = Batch(streams)
    for _ in range(2):
        b.run(); b.status()
    for i, s in enumerate(streams):
        ref = orc.decode(s)
        got = b.planes(i)
        for c in range(3):
            np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))
    bad = bytearray(streams[5])
    for k in range(len(bad) // 2, len(bad) // 2 + 40):
        bad[k] ^= 0xC3
    b2 = Batch(streams[:5] + [bytes(bad)] + streams[6:])
    b2.run()
    with pytest.raises(HipDecError) as e:
        b2.status()
    assert e.value.code == -8
    b3 = Batch(streams[:4])
    b3.run(); b3.status()
    for c in range(3):
        np.testing.assert_array_equal(b3.planes(2)[c], orc.decode(streams[2])["planes"][c])


def test_concurrent_decoders_share_launch_sets():
    """libheif decodes grid tiles on worker threads, one plugin decoder instance each (grid.cc:405-453): concurrent
    decode calls are coalesced into shared batches; every instance must still get exactly its own picture, and a
    corrupt or odd-one-out item must fail / succeed alone."""
    import threading
    from libheif_amd.decoder import HipDecoder, coalesce_stats
    from libheif_amd._capi import HipDecError
    items = []
    for i in range(12):
        w, h = [(128, 64), (64, 128), (200, 136), (72, 40)][i % 4]
        planes = orc.synth_image(w, h, 8, 1, seed=100 + i)
        items.append(orc.encode(planes, qp=22 + i, stress=i & 1))
    items.append(orc.encode(orc.synth_image(64, 64, 10, 1, seed=7), bit_depth=10))    # cannot share an 8-bit batch
    items.append(orc.encode(orc.synth_image(80, 48, 8, 0, seed=8)))                   # monochrome
    bad = bytearray(items[2]); bad[len(bad) * 2 // 3] ^= 0x5a; bad[len(bad) * 2 // 3 + 7] ^= 0xff
    refs = [orc.decode(s) for s in items]
    bad_alone = None
    try:
        _decode_gpu(bytes(bad))
    except HipDecError as e:
        bad_alone = e.code
    items.append(bytes(bad))
    for rnd in range(3):
        r0, _, s0 = coalesce_stats()
        out = [None] * len(items)
        gate = threading.Barrier(len(items))

        def work(k):
            d = HipDecoder()
            d.push_data(items[k])
            gate.wait()
            try:
                out[k] = d.decode_next_image()
            except HipDecError as e:
                out[k] = e
            d.free()

        th = [threading.Thread(target=work, args=(k,)) for k in range(len(items))]
        [t.start() for t in th]
        [t.join() for t in th]
        for k, ref in enumerate(refs):
            assert not isinstance(out[k], Exception), (k, out[k])
            assert len(out[k].planes) == len(ref["planes"])
            for c in range(len(ref["planes"])):
                np.testing.assert_array_equal(out[k].planes[c], ref["planes"][c], err_msg="item %d comp %d" % (k, c))
        if bad_alone is not None:
            assert isinstance(out[-1], HipDecError) and out[-1].code == bad_alone
        r1, _, s1 = coalesce_stats()
        assert r1 - r0 == len(items)
    # the barrier releases all threads together: at least some rounds must have shared a launch set
    assert coalesce_stats()[2] > 0


def test_batch_colour_stage_in_one_launch():
    """hipdec_batch_to_rgb_all (every item's colour conversion in ONE kernel launch) against the per-item entry point:
    mixed picture sizes, and mixed planner decisions (integer op for full range, float op for limited range)."""
    from libheif_amd.decoder import Batch
    specs = [((128, 64), dict(vui_matrix=6, vui_primaries=1, vui_transfer=13, vui_full_range=1)),
             ((200, 136), dict(vui_matrix=1, vui_primaries=1, vui_transfer=1, vui_full_range=0)),
             ((70, 42), dict(vui_matrix=6, vui_primaries=1, vui_transfer=13, vui_full_range=1)),
             ((64, 128), dict())]
    streams = [orc.encode(orc.synth_image(w, h, 8, 1, seed=60 + i), **cfg) for i, ((w, h), cfg) in enumerate(specs)]
    b = Batch(streams)
    b.run(); b.status()
    single = [b.to_rgb(i, 10) for i in range(len(streams))]
    b.alloc_rgb(10)
    for _ in range(2):                 # the second call reuses the uploaded parameter blocks
        b.to_rgb_all()
        for i in range(len(streams)):
            np.testing.assert_array_equal(b.rgb(i), single[i], err_msg="item %d" % i)
    b.alloc_rgb(11)                    # other layout (RGBA): new outputs -> new parameter blocks
    b.to_rgb_all()
    for i in range(len(streams)):
        rgba = b.rgb(i).reshape(single[i].shape[0], -1, 4)
        np.testing.assert_array_equal(rgba[:, :, :3].reshape(single[i].shape), single[i])
        assert (rgba[:, :, 3] == 255).all()


def test_a_stream_of_batches_recycles_one_arena():
    """hipdec_batch_create_recycling: batch k+1 is created (host parsing, staging, upload) while batch k is still in flight and takes
    over its arena; every batch's planes are consumed through the colour stage before its successor exists; a retired batch still
    reports its status but refuses plane reads loudly"""
    from libheif_amd.decoder import Batch
    from libheif_amd import HipDecError
    from tools import streamgen
    sets = [[streamgen.make_stream(1280, 720, 40 + 8 * j + i, 8) for i in range(8)] for j in range(4)]   # upload region > 4 MiB: async path
    first = Batch(sets[0])
    first.alloc_rgb(10)
    rgb_state = first.rgb_state()
    cur, results = first, []
    for j in range(4):
        cur.run()
        cur.to_rgb_all()
        nxt = Batch(sets[j + 1], recycle=cur) if j + 1 < 4 else None     # created while `cur` may still be decoding
        if nxt is not None:
            nxt.use_rgb(rgb_state)
        cur.status()                                                      # the status word was copied back behind cur's kernels
        if nxt is not None:
            with pytest.raises(HipDecError):
                cur.planes(0)
        # the RGB buffers are shared: read them back before the successor's colour stage overwrites them (stream order makes the
        # successor's kernels wait for nothing but its own upload, so synchronise through status() above and read now)
        results.append([np.array(cur.rgb(i), copy=True) for i in (0, 7)])
        if nxt is None:
            planes_last = [cur.planes(i) for i in (0, 7)]
        prev, cur = cur, nxt
        prev.free() if nxt is not None else None
    for j in range(4):
        for n, i in enumerate((0, 7)):
            ref = orc.decode(sets[j][i])
            nclx = tuple(ref["nclx"])
            if nclx[3] and (6 if nclx[2] == 2 else nclx[2]) not in (0, 8):
                want = orc.color_420_to_rgb24(ref["planes"][0], ref["planes"][1], ref["planes"][2], nclx).reshape(720, -1)
            else:
                r, g, b = orc.color_ycbcr_to_rgb_planar(ref["planes"][0], ref["planes"][1], ref["planes"][2], 8, 1, nclx)
                want = orc.color_rgb_planar_to_interleaved8(r, g, b).reshape(720, -1)
            np.testing.assert_array_equal(results[j][n], want, err_msg="batch %d item %d" % (j, i))
    for n, i in enumerate((0, 7)):
        ref = orc.decode(sets[3][i])
        for c in range(3):
            np.testing.assert_array_equal(planes_last[n][c], ref["planes"][c])


@pytest.mark.parametrize("cfg", [dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1), dict(vui_primaries=1, vui_transfer=1, vui_matrix=1, vui_full_range=0),
                                 dict(bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)], ids=["srgb", "bt709_limited", "main10_unfused"])
def test_run_rgb_fused_sao_colour_equals_the_two_step_path(cfg):
    """hipdec_batch_run_rgb: planes and RGB of the fused SAO + colour kernel (8-bit -> RGB24) — and of the unfused fallback (Main10 ->
    RRGGBB) — equal hipdec_batch_run + hipdec_batch_to_rgb_all, which tests/test_color_gpu.py and the parity tests pin to the oracle"""
    from libheif_amd.decoder import Batch
    bd = cfg.get("bit_depth", 8)
    out_chroma = 10 if bd == 8 else 14
    streams = [orc.encode(orc.synth_image(w, h, bd, 1, seed=80 + k), **cfg) for k, (w, h) in enumerate([(456, 264), (200, 136), (70, 42), (1288, 728)])]
    groups = [[streams[0]], [streams[1], streams[1]], [streams[2]], [streams[3]]]
    for g in groups:
        a = Batch(g); a.alloc_rgb(out_chroma); a.run(); a.to_rgb_all(); a.status()
        f = Batch(g); f.alloc_rgb(out_chroma); f.run_rgb(); f.status()
        f.run_rgb(); f.status()                                # and again over the same arena
        for i in range(len(g)):
            np.testing.assert_array_equal(f.rgb(i), a.rgb(i))
            ref = orc.decode(g[i])
            for c in range(3):
                np.testing.assert_array_equal(f.planes(i)[c], ref["planes"][c])
        t = f.kernel_timing_us()
        assert (t["colour"] == 0.0) == (bd == 8)               # fused: no separate colour kernel was timed
        a.free(); f.free()


@pytest.mark.parametrize("size", [(128, 32), (136, 66), (264, 98), (392, 130), (256, 64), (520, 34)], ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("cfg", [dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1, stress=1), dict(stress=1, qp=34), dict(vui_matrix=1, vui_full_range=0, qp=20)],
                         ids=["int88", "float_default", "float_bt709"])
def test_run_rgb_lean_kernel_at_picture_and_output_borders(size, cfg):
    """k_sao_rgb_lean (filter_kernels.hip): pictures whose tiles are ALL border tiles in one way or another - one tile column (the picture's left and
    right border in the same tile), widths that are multiples of 8 but not of the 128-sample tile (quads outside the output), heights that end inside a
    tile, edge offsets against the picture's four borders - planes against the oracle, RGB against the two-step path (k_sao + the batched colour kernel)"""
    from libheif_amd.decoder import Batch
    w, h = size
    streams = [orc.encode(orc.synth_image(w, h, 8, 1, seed=300 + k), **cfg) for k in range(2)]
    a = Batch(streams); a.alloc_rgb(10); a.run(); a.to_rgb_all(); a.status()
    f = Batch(streams); f.alloc_rgb(10); f.run_rgb(); f.status()
    for i in range(len(streams)):
        ref = orc.decode(streams[i])
        for c in range(3):
            np.testing.assert_array_equal(f.planes(i)[c], ref["planes"][c], err_msg="item %d component %d" % (i, c))
        np.testing.assert_array_equal(f.rgb(i), a.rgb(i), err_msg="item %d" % i)
    a.free(); f.free()
