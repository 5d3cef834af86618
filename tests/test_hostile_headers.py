"""Hostile parameter sets and slice headers through hipdec_probe() (host-only, no GPU).

The plugin's input is untrusted (HEIC files from anywhere), so every syntax element that is narrowed, used as an index or as
a loop bound must be range-checked while it is still unsigned.  The streams here are written bit by bit; each one used to
crash, overflow or exhaust memory (negative tile widths from column_width_minus1 = 0xFFFFFFF9, negative
sps index, num_tile_columns_minus1 = 0xFFFFFFFE -> std::length_error, 2^31 entry points with a zero-width offset field,
log2 size deltas that invert the transform-size range, negative conformance offsets, 2^22 x 2^22 pictures).  The bar: a
negative status code and a message, never an exception, abort or out-of-bounds access (tools/emu_asan_fuzz.sh runs the same
generator under AddressSanitizer)."""
import ctypes as C
import pytest

import libheif_amd
from libheif_amd._capi import ImageInfo


class Bits:
    def __init__(self):
        self.b = []

    def u(self, n, v):
        for i in range(n - 1, -1, -1):
            self.b.append((v >> i) & 1)
        return self

    def ue(self, v):
        v += 1
        n = v.bit_length()
        return self.u(n - 1, 0).u(n, v)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def rbsp(self):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        raw = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
        out, zeros = bytearray(), 0
        for x in raw:                       # emulation prevention
            if zeros >= 2 and x <= 3:
                out.append(3); zeros = 0
            out.append(x)
            zeros = zeros + 1 if x == 0 else 0
        return bytes(out)


def nal(t, payload):
    body = bytes([t << 1, 1]) + payload
    return len(body).to_bytes(4, "big") + body


def sps(**o):
    g = lambda k, d: o.get(k, d)
    b = Bits()
    b.u(4, 0).u(3, 0).u(1, 1)
    b.u(2, 0).u(1, 0).u(5, 3).u(32, 1 << 28).u(48, 0).u(8, 90)     # profile_tier_level, no sub-layers
    b.ue(g("sps_id", 0)).ue(g("chroma_format_idc", 1))
    if g("chroma_format_idc", 1) == 3:
        b.u(1, g("separate_colour_plane", 0))
    b.ue(g("width", 64)).ue(g("height", 64))
    conf = g("conf", None)
    b.u(1, 1 if conf else 0)
    if conf:
        for v in conf: b.ue(v)
    b.ue(g("bd_luma_m8", 0)).ue(g("bd_chroma_m8", 0)).ue(g("log2_poc_m4", 4))
    b.u(1, 1).ue(0).ue(0).ue(0)
    b.ue(g("log2_min_cb_m3", 0)).ue(g("log2_diff_cb", 3)).ue(g("log2_min_tb_m2", 0)).ue(g("log2_diff_tb", 3))
    b.ue(g("th_inter", 1)).ue(g("th_intra", 1))
    b.u(1, 0)                       # scaling_list_enabled
    b.u(1, 0).u(1, 1).u(1, 0)       # amp, sao, pcm
    b.ue(g("num_st_rps", 0))
    b.u(1, 0).u(1, 0).u(1, 1)       # long-term, temporal mvp, strong intra smoothing
    b.u(1, 0).u(1, 0)               # vui, extension
    return nal(33, b.rbsp())


def pps(**o):
    g = lambda k, d: o.get(k, d)
    b = Bits()
    b.ue(g("pps_id", 0)).ue(g("sps_id", 0))
    b.u(1, 0).u(1, 0).u(3, 0).u(1, 0).u(1, 0)
    b.ue(0).ue(0).se(g("init_qp_m26", 0))
    b.u(1, 0).u(1, 0)
    b.u(1, 1 if "cu_qp_depth" in o else 0)
    if "cu_qp_depth" in o: b.ue(o["cu_qp_depth"])
    b.se(g("cb_off", 0)).se(g("cr_off", 0))
    b.u(1, 0).u(2, 0).u(1, 0)
    tiles = g("tiles", None)
    b.u(1, 1 if tiles else 0).u(1, g("wpp", 0))
    if tiles:
        b.ue(tiles["cols_m1"]).ue(tiles["rows_m1"]).u(1, 1 if tiles.get("uniform", 1) else 0)
        for v in tiles.get("col_w_m1", []): b.ue(v)
        for v in tiles.get("row_h_m1", []): b.ue(v)
        b.u(1, 1)
    b.u(1, 1)                       # loop filter across slices
    b.u(1, 0).u(1, 0).u(1, 0).ue(0).u(1, 0)
    rext = g("range_ext", None)     # pps_range_extension(): (cross_component_prediction, chroma_qp_offset_list_enabled, log2_sao_offset_scale_luma, _chroma)
    b.u(1, 1 if rext else 0)
    if rext:
        b.u(1, 1).u(7, 0)
        b.u(1, rext[0]).u(1, rext[1]).ue(rext[2]).ue(rext[3])
    return nal(34, b.rbsp())


def idr(**o):
    """slice segment header for the parameter sets above (sao on, no deblocking control, lf across slices on)"""
    g = lambda k, d: o.get(k, d)
    b = Bits()
    b.u(1, 1).u(1, 0).ue(g("pps_id", 0)).ue(2)
    b.u(1, 1).u(1, 1)               # sao luma / chroma
    b.se(g("qp_delta", 0))
    b.u(1, 1)                       # slice_loop_filter_across_slices_enabled_flag (sao on)
    if "entry" in o:
        n, len_m1, offs = o["entry"]
        b.ue(n)
        if n: b.ue(len_m1)
        for v in offs: b.u((len_m1 + 1) if len_m1 < 32 else 0, v)
    return nal(19, b.rbsp() + bytes(g("payload", 32)))


def probe(data, max_px=0):
    lib = libheif_amd.load_library()
    lib.hipdec_probe.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.POINTER(ImageInfo)]
    info = ImageInfo()
    rc = lib.hipdec_probe(data, len(data), max_px, C.byref(info))
    return rc, lib.hipdec_last_error().decode()


def test_the_handwritten_baseline_is_accepted():
    rc, msg = probe(sps() + pps() + idr())
    assert rc == 0, msg
    rc, msg = probe(sps(width=128) + pps(wpp=1, tiles=dict(cols_m1=1, rows_m1=0, uniform=0, col_w_m1=[0])) + idr(entry=(1, 7, [5])))
    assert rc == 0, msg


U32 = 0xFFFFFFFE   # the largest value a 32-bit ue(v) can carry

HOSTILE = {
    "pps sps_id wraps negative": sps() + pps(sps_id=0xFFFFFFD6) + idr(),
    "pps sps_id 16": sps() + pps(sps_id=16) + idr(),
    "pps id 64": sps() + pps(pps_id=64) + idr(pps_id=64),
    "num_tile_columns_minus1 = 2^32-2": sps() + pps(tiles=dict(cols_m1=U32, rows_m1=0)) + idr(),
    "num_tile_rows_minus1 = 2^32-2": sps() + pps(tiles=dict(cols_m1=0, rows_m1=U32)) + idr(),
    "column_width_minus1 = 0xFFFFFFF9": sps(width=256) + pps(tiles=dict(cols_m1=2, rows_m1=0, uniform=0, col_w_m1=[0xFFFFFFF9, 3])) + idr(entry=(2, 7, [1, 1])),
    "column widths exceed the picture": sps(width=256) + pps(tiles=dict(cols_m1=2, rows_m1=0, uniform=0, col_w_m1=[2, 2])) + idr(entry=(2, 7, [1, 1])),
    "row heights exceed the picture": sps(height=128) + pps(tiles=dict(cols_m1=0, rows_m1=1, uniform=0, row_h_m1=[1])) + idr(entry=(1, 7, [1])),
    "more tile columns than CTBs": sps() + pps(tiles=dict(cols_m1=3, rows_m1=0)) + idr(),
    "2^31 entry points": sps() + pps(wpp=1) + idr(entry=(1 << 31, U32, [])),
    "entry points, offset_len_minus1 = 2^32-2": sps(height=128) + pps(wpp=1) + idr(entry=(1, U32, [])),
    "entry point beyond the NAL": sps(height=128) + pps(wpp=1) + idr(entry=(1, 31, [0xFFFFFFF0])),
    "log2_diff_max_min_tb huge": sps(log2_diff_tb=U32) + pps() + idr(),
    "min tb above min cb": sps(log2_min_tb_m2=2, log2_diff_tb=0) + pps() + idr(),
    "log2_min_cb huge": sps(log2_min_cb_m3=U32) + pps() + idr(),
    "ctb of 128": sps(log2_min_cb_m3=3, log2_diff_cb=1) + pps() + idr(),
    "bit depth wraps negative": sps(bd_luma_m8=U32) + pps() + idr(),
    "bit depth 17": sps(bd_luma_m8=9, bd_chroma_m8=9) + pps() + idr(),
    "transform hierarchy depth huge": sps(th_intra=U32) + pps() + idr(),
    "transform hierarchy depth 5": sps(th_intra=5) + pps() + idr(),
    "conformance offset wraps negative": sps(conf=(U32, 0, 0, 0)) + pps() + idr(),
    "conformance window eats the picture": sps(conf=(16, 16, 0, 0)) + pps() + idr(),
    "picture 2^22 x 2^22": sps(width=1 << 22, height=1 << 22) + pps() + idr(),
    "picture 65528 x 65528": sps(width=65528, height=65528) + pps() + idr(),
    "diff_cu_qp_delta_depth wraps negative": sps() + pps(cu_qp_depth=U32) + idr(),
    "chroma qp offset 13": sps() + pps(cb_off=13) + idr(),
    "init_qp out of range": sps() + pps(init_qp_m26=80) + idr(),
    "chroma_format_idc 7": sps(chroma_format_idc=7) + pps() + idr(),
    "65 short-term RPS": sps(num_st_rps=65) + pps() + idr(),
    "4:4:4 with separate colour planes": sps(chroma_format_idc=3, separate_colour_plane=1) + pps() + idr(),
    "4:2:2 conformance window eats the picture (x in chroma units)": sps(chroma_format_idc=2, conf=(16, 16, 0, 0)) + pps() + idr(),
    "4:2:2 conformance window eats the picture (y in luma rows)": sps(chroma_format_idc=2, conf=(0, 0, 32, 32)) + pps() + idr(),
    "cross-component prediction": sps() + pps(range_ext=(1, 0, 0, 0)) + idr(),
    "chroma qp offset lists": sps() + pps(range_ext=(0, 1, 0, 0)) + idr(),
    "sao offset scale": sps() + pps(range_ext=(0, 0, 0, 2)) + idr(),
    "sps id 16": sps(sps_id=16) + pps() + idr(),
}


@pytest.mark.parametrize("name", sorted(HOSTILE))
def test_hostile_headers_are_rejected_cleanly(name):
    rc, msg = probe(HOSTILE[name])
    assert rc < 0 and msg, (name, rc, msg)
    assert rc in (-3, -4, -5), (name, rc, msg)      # bitstream / unsupported / limit: never a crash, never "memory"


def test_other_chroma_formats_pass_the_front_end_with_their_own_window_units():
    # (left, right, top, bottom) in chroma units: x doubles for 4:2:2, y never does; the last window leaves one luma row
    for cfi, conf in ((2, (1, 2, 3, 4)), (3, (1, 2, 3, 4)), (2, (0, 0, 31, 32))):
        rc, msg = probe(sps(chroma_format_idc=cfi, conf=conf) + pps() + idr())
        assert rc == 0, (cfi, conf, msg)


def test_pps_range_extension_that_enables_nothing_is_accepted():
    rc, msg = probe(sps() + pps(range_ext=(0, 0, 0, 0)) + idr())
    assert rc == 0, msg
    from oracle import pyoracle as orc
    with pytest.raises(orc.OracleError) as e:                       # the oracle parses it the same way (then runs out of slice data)
        orc.decode(sps() + pps(range_ext=(0, 0, 0, 0)) + idr())
    assert "unsupported" not in str(e.value)
    with pytest.raises(orc.OracleError) as e:
        orc.decode(sps() + pps(range_ext=(1, 0, 0, 0)) + idr())
    assert "cross-component" in str(e.value)


def test_size_limit_is_applied_before_any_table_is_sized():
    rc, msg = probe(sps(width=16384, height=16384) + pps() + idr(), max_px=1 << 20)
    assert rc == -5, msg
