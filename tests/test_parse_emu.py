"""CPU check of the device parser's logic: libheif_amd/csrc/parse_core.h compiled for the host with the
64 lanes emulated (tests/emu) against the oracle's taps — unit maps, coefficient levels, SAO
parameters and exact substream termination — over the coding-tool matrix.  The emulation is test
infrastructure; the product runs the same source on the GPU (tests/test_decode_gpu.py)."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

from oracle import pyoracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_emu(target=None):
    """make -C tests/emu [target] under a file lock: pytest-xdist workers (and tests/test_product_on_emulator.py) ask for it at the same time"""
    import fcntl
    with open(os.path.join(HERE, "emu", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")] + ([target] if target else []))


def emu():
    global _LIB
    if _LIB is None:
        build_emu()
        L = C.CDLL(os.path.join(HERE, "emu", "libparse_emu.so"))
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.emu_free.argtypes = [C.c_void_p]
        L.emu_run_parse.argtypes = [C.c_void_p]
        L.emu_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.emu_maps.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.emu_coeffs.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
        L.emu_sao.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
        _LIB = L
    return _LIB


def run_emu(streams):
    L = emu()
    n = len(streams)
    arr = (C.c_char_p * n)(*streams)
    sizes = (C.c_size_t * n)(*[len(s) for s in streams])
    err = C.create_string_buffer(512)
    h = L.emu_create(n, arr, sizes, err, 512)
    assert h, err.value.decode()
    status = L.emu_run_parse(h)
    out = []
    if status != 0:   # maps / coefficients of a desynchronised parse are garbage: do not walk them
        L.emu_free(h)
        return status, out
    for i in range(n):
        info = (C.c_int * 7)()
        L.emu_info(h, i, info)
        w, hgt, ctb_w, ctb_h, log2_ctb, cf, nsub = list(info)
        uw, uh = (w + 3) // 4, (hgt + 3) // 4
        names = ["log2_tb", "log2_cb", "intra_luma", "intra_chroma", "qp_y", "flags"]
        maps = [np.zeros((uh, uw), np.int8 if k == "qp_y" else np.uint8) for k in names]
        L.emu_maps(h, i, *[m.ctypes.data for m in maps])
        ch, cw = (hgt if cf in (2, 3) else hgt // 2), (w if cf == 3 else w // 2)      # coded chroma plane size (4:2:2: half as wide, as tall)
        coef = [np.zeros((hgt, w), np.int32), np.zeros((ch, cw), np.int32), np.zeros((ch, cw), np.int32)]
        assert L.emu_coeffs(h, i, *[c.ctypes.data for c in coef]) == 0
        nctb = ctb_w * ctb_h
        st, sc, so = np.zeros((nctb, 3), np.uint8), np.zeros((nctb, 3), np.uint8), np.zeros((nctb, 3, 4), np.int16)
        L.emu_sao(h, i, st.ctypes.data, sc.ctypes.data, so.ctypes.data)
        out.append(dict(zip(names, maps), coef=coef, sao_type=st, sao_cls=sc, sao_off=so, cf=cf, nsub=nsub))
    L.emu_free(h)
    return status, out


def check_against_oracle(stream, got):
    ref = orc.decode(stream, taps=True)
    np.testing.assert_array_equal(got["log2_cb"], ref["map_log2_cb"])
    np.testing.assert_array_equal(got["log2_tb"], ref["map_log2_tb"])
    np.testing.assert_array_equal(got["intra_luma"], ref["map_intra_luma"])
    np.testing.assert_array_equal(got["intra_chroma"], ref["map_intra_chroma"])
    np.testing.assert_array_equal(got["qp_y"], ref["map_qp_y"])
    # 4:2:2: a unit's cbf_cb / cbf_cr bits describe ONE of the two chroma blocks (the lower block's flags sit in the neighbouring unit), the oracle's
    # tap has their union in every unit of the block: the coefficient comparison below covers them
    fmask = 0x79 if got["cf"] == 2 else 0x7f
    np.testing.assert_array_equal(got["flags"] & fmask, ref["map_flags"] & fmask)
    for c in range(3 if got["cf"] else 1):
        np.testing.assert_array_equal(got["coef"][c], ref["coeff"][c], err_msg="coefficients of component %d" % c)
    ncomp = 3 if got["cf"] else 1
    np.testing.assert_array_equal(got["sao_type"][:, :ncomp], ref["sao_type"][:, :ncomp])
    on = got["sao_type"][:, :ncomp] != 0
    np.testing.assert_array_equal(got["sao_cls"][:, :ncomp][on], ref["sao_band_or_class"][:, :ncomp][on])
    np.testing.assert_array_equal(got["sao_off"][:, :ncomp][on], ref["sao_offset"][:, :ncomp][on])
    assert got["nsub"] == ref["n_substreams"]


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
    dict(qp=4, stress=1),
    dict(pcm_pct=20),
    dict(pcm_pct=30, pcm_loop_filter_disabled=1, bit_depth=10, stress=1),
    dict(pcm_pct=40, wpp=0, log2_ctb=5, log2_max_tb=5, lossless_pct=20),
    dict(dependent_segments=3, wpp=0),                                        # dependent slice segments: contexts / QP / left neighbour continue
    dict(dependent_segments=4, wpp=0, num_slices=2, stress=1, log2_ctb=4, log2_max_tb=4),
    dict(dependent_segments=3, wpp=1, stress=1),                              # ... starting at CTB row starts under WPP
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: ",".join("%s=%s" % kv for kv in c.items()) or "default")
@pytest.mark.parametrize("size", [(200, 136), (64, 64), (328, 72)])
def test_parser_emulation_matches_oracle(cfg, size):
    bd = cfg.get("bit_depth", 8)
    planes = orc.synth_image(size[0], size[1], bd, 1, seed=3 + size[0])
    stream = orc.encode(planes, **cfg)
    status, got = run_emu([stream])
    assert status == 0, "device status 0x%x" % status
    check_against_oracle(stream, got[0])


def test_parser_emulation_monochrome_and_batch():
    streams = []
    for i, (w, h, cf) in enumerate([(75, 41, 0), (70, 42, 1), (8, 8, 1), (136, 24, 1), (264, 200, 1)]):
        streams.append(orc.encode(orc.synth_image(w, h, 8, cf, seed=5 + i), wpp=i % 2, stress=i % 3 == 0))
    status, got = run_emu(streams)
    assert status == 0
    for s, g in zip(streams, got):
        check_against_oracle(s, g)


def test_parser_emulation_reports_desync():
    stream = bytearray(orc.encode(orc.synth_image(128, 128, 8, 1, seed=9)))
    for k in range(len(stream) - 80, len(stream) - 30):
        stream[k] ^= 0x5A
    status, _ = run_emu([bytes(stream)])
    assert status != 0


def test_parser_emulation_reference_fixtures(reference_dir):
    from heic_util import HeicFile
    for rel in ("examples/example.heic", "tests/data/rainbow-451x461.heic", "tests/data/with-alpha-512x512.heic"):
        f = HeicFile(os.path.join(reference_dir, rel))
        for iid in f.hevc_items():
            s = f.plugin_stream(iid)
            status, got = run_emu([s])
            assert status == 0, rel
            check_against_oracle(s, got[0])


@pytest.mark.parametrize("yield_ctbs", [0, 1, 3])
@pytest.mark.parametrize("cfg", [dict(), dict(stress=1, log2_ctb=4, log2_max_tb=4), dict(tile_cols=3, tile_rows=2, wpp=1),
                                 dict(num_slices=3), dict(wpp=0), dict(bit_depth=10)],
                         ids=["default", "ctb16", "tiles_wpp", "slices", "nowpp", "main10"])
def test_parser_pool_scheduler_emulation(cfg, yield_ctbs, monkeypatch):
    """throughput-mode scheduling: rows as tasks of a work pool with suspend / resume through HBM state.  The forced
    yield makes rows interleave so that real dependency suspensions, wake-ups and state restores happen."""
    monkeypatch.setenv("HIPDEC_PARSE_POOL", "1")
    monkeypatch.setenv("HIPDEC_POOL_YIELD", str(yield_ctbs))
    streams = [orc.encode(orc.synth_image(w, h, cfg.get("bit_depth", 8), 1, seed=50 + i), **cfg) for i, (w, h) in enumerate([(264, 200), (200, 136), (328, 72)])]
    status, got = run_emu(streams)
    assert status == 0, "device status 0x%x" % status
    for s, g in zip(streams, got):
        check_against_oracle(s, g)


def _unescape_like_the_spec(buf):
    """7.4.2: 00 00 03 -> 00 00 inside a NAL unit; a trailing 03 (no byte behind it) is data"""
    out, zeros, i = bytearray(), 0, 0
    while i < len(buf):
        b = buf[i]
        if zeros >= 2 and b == 3 and i + 1 < len(buf):
            zeros = 0; i += 1
            continue
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
        i += 1
    return bytes(out)


@pytest.mark.parametrize("seed", range(6))
def test_byte_reader_skips_emulation_prevention_across_windows_and_resumes(seed):
    """the 256-byte-window reader of parse_core.h (candidate detection once per window, short path in clean windows, per-byte
    path elsewhere) against the definition, on buffers dense in 00 00 03 / 00 03 / 03 patterns placed around the window
    boundaries, from every kind of start offset, with and without suspend / resume in between"""
    import ctypes as C
    import random
    lib = emu()
    lib.emu_read_bytes.restype = C.c_int
    lib.emu_read_bytes.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32]
    rng = random.Random(seed)
    n = 256 * 9
    if seed == 0:
        body = bytearray(rng.randrange(1, 256) for _ in range(n))              # clean windows only ...
        for at in (254, 255, 256, 510, 511, 512, 513, 767, 1024 + 1, 1280 - 3, 1535, 2047, 2048):
            body[at:at + 3] = b"\x00\x00\x03"                                    # ... except patterns on the seams
    elif seed == 1:
        body = bytearray(rng.choice(b"\x00\x00\x00\x03\x03\x01\x02\xff") for _ in range(n))   # dense
    else:
        body = bytearray(rng.randrange(256) for _ in range(n))
        for _ in range(60):
            at = rng.randrange(n - 4)
            body[at:at + rng.choice([3, 4, 5])] = rng.choice([b"\x00\x00\x03", b"\x00\x00\x03\x00\x00\x03", b"\x00\x03", b"\x00\x00\x00\x03\x01"])[:5]
    buf = bytes(body) + bytes(1024)
    out = C.create_string_buffer(n + 16)
    for start, end in [(0, n), (5, n - 7), (255, 1300), (256, 1024), (257, 770), (511, 515), (700, 701), (1000, 1000 + 256), (3, 256 * 8 + 1)]:
        want = _unescape_like_the_spec(buf[start:end])
        for resume_every in (0, 1, 7, 255, 256, 300):
            got = lib.emu_read_bytes(buf, start, end, out, n + 16, resume_every)
            assert out.raw[:got] == want, (seed, start, end, resume_every)


def test_missing_dependent_segment_is_a_device_error():
    """a slice whose middle (dependent) segment NAL is missing: the first segment's substream ends early (end_of_slice_segment_flag = 1 before the
    CTB count the headers imply) -> DEV_ERR_TERMINATE, never a mis-decode"""
    s = orc.encode(orc.synth_image(200, 136, 8, 1, seed=4), dependent_segments=3, wpp=0)
    nals, p = [], 0
    while p < len(s):
        n = int.from_bytes(s[p:p + 4], "big"); nals.append(s[p:p + 4 + n]); p += 4 + n
    sl = [x for x in nals if (x[4] >> 1) & 63 < 32]
    broken = b"".join([x for x in nals if (x[4] >> 1) & 63 >= 32] + [sl[0], sl[2]])
    try:
        status, _ = run_emu([broken])
    except AssertionError:
        return          # refused by the host front end: also fine
    assert status != 0


def test_context_init_tables_of_the_device_parser_equal_the_oracles():
    """The initValue tables (9.3.2.2, tables 9-5 .. 9-37) are typed in twice, once per side - the oracle's flat table per initType
    (oracle/hevc_oracle.c) and the device's per-group / per-lane layout (csrc/parse_tables.h).  Same numbers, context by context, for
    slice_type I and both P initTypes; the padding lanes hold 154."""
    L = emu()
    L.emu_ctx_init_value.argtypes = [C.c_int] * 3
    O = orc.lib()
    n_ctx = 155
    tab_i = list((C.c_uint8 * n_ctx).in_dll(O, "hevc_cabac_init_I"))
    tab_p = list((C.c_uint8 * (2 * n_ctx)).in_dll(O, "hevc_cabac_init_P"))
    oracle = [tab_i, tab_p[:n_ctx], tab_p[n_ctx:]]
    # (group, first lane, count, first oracle context): the enums of parse_tables.h against those of hevc_oracle_internal.h
    layout = [(0, 0, 62, 0),        # sao_merge .. coded_sub_block_flag: same order on both sides
              (0, 62, 1, 134),      # cbf_cb / cbf_cr at trafoDepth 4
              (1, 0, 42, 62),       # sig_coeff_flag
              (1, 44, 6, 128),      # coeff_abs_level_greater2
              (2, 0, 24, 104),      # coeff_abs_level_greater1
              (2, 24, 20, 135)]     # cu_skip_flag .. rqt_root_cbf, inter_pred_idc (P / B slices)
    covered = set()
    for table in range(3):
        used = {g: set() for g in range(3)}
        for g, lane0, count, ctx0 in layout:
            for k in range(count):
                if table == 0 and ctx0 >= 135:
                    continue    # the I table of the device carries no inter contexts
                assert L.emu_ctx_init_value(table, g, lane0 + k) == oracle[table][ctx0 + k], (table, g, lane0 + k, ctx0 + k)
                used[g].add(lane0 + k)
                covered.add(ctx0 + k)
        for g in range(3):
            for lane in range(64):
                if lane not in used[g] and not (g == 1 and lane in (42, 43)):
                    assert L.emu_ctx_init_value(table, g, lane) == 154, (table, g, lane)
    assert covered == set(range(n_ctx))
