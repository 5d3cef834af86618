"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY: may be imported
from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from libheif_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c") or f.endswith(".h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


class _Pic(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("chroma_format_idc", C.c_int),
        ("bit_depth_luma", C.c_int), ("bit_depth_chroma", C.c_int),
        ("cwidth", C.c_int), ("cheight", C.c_int),
        ("plane", C.POINTER(C.c_uint16) * 3),
        ("colour_primaries", C.c_int), ("transfer_characteristics", C.c_int),
        ("matrix_coeffs", C.c_int), ("full_range_flag", C.c_int),
        ("coded_width", C.c_int), ("coded_height", C.c_int),
        ("ccoded_width", C.c_int), ("ccoded_height", C.c_int),
        ("pre_deblock", C.POINTER(C.c_uint16) * 3),
        ("post_deblock", C.POINTER(C.c_uint16) * 3),
        ("final_coded", C.POINTER(C.c_uint16) * 3),
        ("coeff", C.POINTER(C.c_int32) * 3),
        ("map_stride", C.c_int), ("map_height", C.c_int),
        ("map_log2_tb", C.POINTER(C.c_uint8)), ("map_log2_cb", C.POINTER(C.c_uint8)),
        ("map_intra_luma", C.POINTER(C.c_uint8)), ("map_intra_chroma", C.POINTER(C.c_uint8)),
        ("map_qp_y", C.POINTER(C.c_int8)), ("map_flags", C.POINTER(C.c_uint8)),
        ("ctb_log2", C.c_int), ("ctbs_w", C.c_int), ("ctbs_h", C.c_int),
        ("sao_type", C.POINTER(C.c_uint8)), ("sao_band_or_class", C.POINTER(C.c_uint8)),
        ("sao_offset", C.POINTER(C.c_int16)),
        ("n_bins_ctx", C.c_uint64), ("n_bins_bypass", C.c_uint64), ("n_substreams", C.c_int),
        ("poc", C.c_int), ("map_pred", C.POINTER(C.c_uint8)), ("mf_mv", C.POINTER(C.c_int16)), ("mf_ref", C.POINTER(C.c_int8)),
    ]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.hevc_oracle_decode.restype = C.c_int
        _LIB.hevc_oracle_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(_Pic), C.c_char_p, C.c_size_t]
        _LIB.hevc_oracle_free_picture.argtypes = [C.POINTER(_Pic)]
    return _LIB


class OracleError(RuntimeError):
    pass


def _arr(ptr, shape, dtype):
    if not ptr:
        return None
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True).reshape(shape)


def decode(stream: bytes, taps: bool = False) -> dict:
    """Decode one length-prefixed-NAL HEVC intra picture.  Returns a dict of numpy arrays."""
    L = lib()
    pic = _Pic()
    err = C.create_string_buffer(512)
    rc = L.hevc_oracle_decode(stream, len(stream), 1 if taps else 0, C.byref(pic), err, 512)
    if rc != 0:
        raise OracleError(err.value.decode("latin1"))
    return _picture_dict(L, pic, taps)


class SeqDecoder:
    """hevc_oracle_seq_*: one access unit per decode() call in decoding order (the samples of a track); P slices are decoded, parameter
    sets and the decoded picture buffer persist between calls."""

    def __init__(self):
        L = lib()
        L.hevc_oracle_seq_new.restype = C.c_void_p
        L.hevc_oracle_seq_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(_Pic), C.c_char_p, C.c_size_t]
        L.hevc_oracle_seq_free.argtypes = [C.c_void_p]
        self._L = L
        self._h = C.c_void_p(L.hevc_oracle_seq_new())

    def decode(self, stream: bytes, taps: bool = False) -> dict:
        pic = _Pic()
        err = C.create_string_buffer(512)
        rc = self._L.hevc_oracle_seq_decode(self._h, stream, len(stream), 1 if taps else 0, C.byref(pic), err, 512)
        if rc != 0:
            raise OracleError(err.value.decode("latin1"))
        return _picture_dict(self._L, pic, taps)

    def close(self):
        if self._h:
            self._L.hevc_oracle_seq_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_sequence(streams, taps=False):
    q = SeqDecoder()
    try:
        return [q.decode(s, taps) for s in streams]
    finally:
        q.close()


def _picture_dict(L, pic, taps):
    try:
        nc = 3 if pic.chroma_format_idc else 1
        out = {
            "width": pic.width, "height": pic.height, "chroma_format_idc": pic.chroma_format_idc,
            "bit_depth_luma": pic.bit_depth_luma, "bit_depth_chroma": pic.bit_depth_chroma,
            "nclx": (pic.colour_primaries, pic.transfer_characteristics, pic.matrix_coeffs, pic.full_range_flag),
            "coded_size": (pic.coded_width, pic.coded_height),
            "n_bins_ctx": pic.n_bins_ctx, "n_bins_bypass": pic.n_bins_bypass, "n_substreams": pic.n_substreams,
            "poc": pic.poc,
            "planes": [],
        }
        for c in range(nc):
            w, h = (pic.width, pic.height) if c == 0 else (pic.cwidth, pic.cheight)
            out["planes"].append(_arr(pic.plane[c], (h, w), np.uint16))
        if taps:
            for name in ("pre_deblock", "post_deblock", "final_coded", "coeff"):
                lst = []
                for c in range(nc):
                    w, h = (pic.coded_width, pic.coded_height) if c == 0 else (pic.ccoded_width, pic.ccoded_height)
                    lst.append(_arr(getattr(pic, name)[c], (h, w), np.int32 if name == "coeff" else np.uint16))
                out[name] = lst
            ms, mh = pic.map_stride, pic.map_height
            for name, dt in (("map_log2_tb", np.uint8), ("map_log2_cb", np.uint8), ("map_intra_luma", np.uint8),
                             ("map_intra_chroma", np.uint8), ("map_qp_y", np.int8), ("map_flags", np.uint8)):
                out[name] = _arr(getattr(pic, name), (mh, ms), dt)
            nctb = pic.ctbs_w * pic.ctbs_h
            out["ctb_log2"] = pic.ctb_log2
            out["ctbs"] = (pic.ctbs_w, pic.ctbs_h)
            out["sao_type"] = _arr(pic.sao_type, (nctb, 3), np.uint8)
            out["sao_band_or_class"] = _arr(pic.sao_band_or_class, (nctb, 3), np.uint8)
            out["sao_offset"] = _arr(pic.sao_offset, (nctb, 3, 4), np.int16)
            if pic.map_pred:
                out["map_pred"] = _arr(pic.map_pred, (mh, ms), np.uint8)
                out["mf_mv"] = _arr(pic.mf_mv, (mh, ms, 2, 2), np.int16)    # [row, column, list, component]
                out["mf_ref"] = _arr(pic.mf_ref, (mh, ms, 2), np.int8)
        return out
    finally:
        L.hevc_oracle_free_picture(C.byref(pic))


# ------------------------------------------------------------------------------------------------
# colour-stage oracle (oracle/color_oracle.c)
# ------------------------------------------------------------------------------------------------
def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def _p16(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def color_coeffs(has_nclx, matrix, primaries):
    out = (C.c_float * 4)()
    lib().color_oracle_coeffs(int(has_nclx), int(matrix), int(primaries), out)
    return [float(v) for v in out]


def color_420_to_rgb24(y, cb, cr, nclx=None, alpha=False):
    """a9.  nclx = (primaries, transfer, matrix, full_range) or None."""
    y, cb, cr = _u16(y), _u16(cb), _u16(cr)
    h, w = y.shape
    bpp = 4 if alpha else 3
    out = np.zeros((h, w * bpp), np.uint8)
    lib().color_oracle_420_to_rgb24(_p16(y), w, _p16(cb), cb.shape[1], _p16(cr), cr.shape[1], w, h,
                                    int(nclx is not None), nclx[2] if nclx else 2, nclx[0] if nclx else 2,
                                    out.ctypes.data_as(C.POINTER(C.c_uint8)), w * bpp, int(alpha))
    return out.reshape(h, w, bpp)


def color_ycbcr_to_rgb_planar(y, cb, cr, bpp, chroma, nclx=None):
    """a10.  Returns (R, G, B) uint16 arrays."""
    y, cb, cr = _u16(y), _u16(cb), _u16(cr)
    h, w = y.shape
    r, g, b = (np.zeros((h, w), np.uint16) for _ in range(3))
    lib().color_oracle_ycbcr_to_rgb_planar(_p16(y), w, _p16(cb), cb.shape[1], _p16(cr), cr.shape[1], w, h, bpp, chroma,
                                           int(nclx is not None), nclx[2] if nclx else 2, nclx[0] if nclx else 2,
                                           nclx[3] if nclx else 1, _p16(r), _p16(g), _p16(b), w)
    return r, g, b


def color_rgb_planar_to_interleaved8(r, g, b, alpha=False):
    r, g, b = _u16(r), _u16(g), _u16(b)
    h, w = r.shape
    bpp = 4 if alpha else 3
    out = np.zeros((h, w * bpp), np.uint8)
    lib().color_oracle_rgb_planar_to_interleaved8(_p16(r), _p16(g), _p16(b), w, w, h,
                                                  out.ctypes.data_as(C.POINTER(C.c_uint8)), w * bpp, int(alpha))
    return out.reshape(h, w, bpp)


def color_420_to_rrggbb(y, cb, cr, bpp, nclx=None, little_endian=True):
    """a12.  Returns uint8 array (h, w*6)."""
    y, cb, cr = _u16(y), _u16(cb), _u16(cr)
    h, w = y.shape
    out = np.zeros((h, w * 6), np.uint8)
    lib().color_oracle_420_to_rrggbb(_p16(y), w, _p16(cb), cb.shape[1], _p16(cr), cr.shape[1], w, h, bpp,
                                     int(nclx is not None), nclx[2] if nclx else 2, nclx[0] if nclx else 2,
                                     nclx[3] if nclx else 1, out.ctypes.data_as(C.POINTER(C.c_uint8)), w * 6,
                                     int(little_endian))
    return out


def color_bilinear_420_to_444(plane, w, h):
    """a13 for one chroma plane; (w, h) is the luma size."""
    plane = _u16(plane)
    out = np.zeros((h, w), np.uint16)
    lib().color_oracle_bilinear_420_to_444(_p16(plane), plane.shape[1], w, h, _p16(out), w)
    return out


def color_bilinear_422_to_444(plane, w, h):
    """Op_YCbCr422_bilinear_to_YCbCr444 for one chroma plane; (w, h) is the luma size."""
    plane = _u16(plane)
    out = np.zeros((h, w), np.uint16)
    lib().color_oracle_bilinear_422_to_444(_p16(plane), plane.shape[1], w, h, _p16(out), w)
    return out


def color_to_sdr(plane, bits):
    plane = _u16(plane)
    h, w = plane.shape
    out = np.zeros((h, w), np.uint16)
    lib().color_oracle_to_sdr(_p16(plane), w, w, h, bits, _p16(out), w)
    return out


# ------------------------------------------------------------------------------------------------
# test-stream generator (oracle/hevc_testenc.c)
# ------------------------------------------------------------------------------------------------
class _EncParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "width", "height", "bit_depth", "chroma_format_idc", "log2_ctb", "log2_min_cb", "log2_min_tb", "log2_max_tb",
        "max_transform_hierarchy_depth_intra", "qp", "wpp", "tile_cols", "tile_rows", "num_slices", "sao",
        "deblock_disable", "beta_offset_div2", "tc_offset_div2", "sign_data_hiding", "cu_qp_delta",
        "diff_cu_qp_delta_depth", "transform_skip", "lossless_pct", "pcm_pct", "pcm_loop_filter_disabled",
        "strong_intra_smoothing", "scaling_list", "cb_qp_offset", "cr_qp_offset", "loop_filter_across_tiles",
        "loop_filter_across_slices", "vui_primaries", "vui_transfer", "vui_matrix", "vui_full_range")] + \
        [("seed", C.c_uint32), ("stress", C.c_int), ("zero_residual_pct", C.c_int), ("dependent_segments", C.c_int)] + \
        [(n, C.c_int) for n in (
            "inter_num_refs", "inter_skip_pct", "inter_intra_pct", "inter_merge_pct", "amp", "max_merge_cand", "parallel_merge_level",
            "max_transform_hierarchy_depth_inter", "cabac_init_present", "lists_modification", "global_mv_x", "global_mv_y",
            "b_frames", "b_ref", "inter_bi_pct", "temporal_mvp", "weighted_pred", "mvd_l1_zero", "constrained_intra_pred", "long_term_ref", "open_gop", "hidden_poc")]


ENC_DEFAULTS = dict(bit_depth=8, chroma_format_idc=1, log2_ctb=6, log2_min_cb=3, log2_min_tb=2, log2_max_tb=5,
                    max_transform_hierarchy_depth_intra=1, qp=27, wpp=1, tile_cols=1, tile_rows=1, num_slices=1, sao=1,
                    deblock_disable=0, beta_offset_div2=0, tc_offset_div2=0, sign_data_hiding=1, cu_qp_delta=1,
                    diff_cu_qp_delta_depth=1, transform_skip=0, lossless_pct=0, pcm_pct=0, pcm_loop_filter_disabled=0,
                    strong_intra_smoothing=1, scaling_list=0, cb_qp_offset=0, cr_qp_offset=0, loop_filter_across_tiles=1,
                    loop_filter_across_slices=1, vui_primaries=1, vui_transfer=13, vui_matrix=-1, vui_full_range=0,
                    seed=1, stress=0, zero_residual_pct=0, dependent_segments=0,
                    inter_num_refs=1, inter_skip_pct=20, inter_intra_pct=10, inter_merge_pct=40, amp=0, max_merge_cand=5, parallel_merge_level=2,
                    max_transform_hierarchy_depth_inter=1, cabac_init_present=0, lists_modification=0, global_mv_x=0, global_mv_y=0,
                    b_frames=0, b_ref=0, inter_bi_pct=50, temporal_mvp=0, weighted_pred=0, mvd_l1_zero=0, constrained_intra_pred=0, long_term_ref=0, open_gop=0, hidden_poc=0)


def synth_image(width, height, bit_depth=8, chroma_format_idc=1, seed=1):
    """Seeded band-limited noise + gradients + a few hard edges (SURVEY.md §8d synthetic content)."""
    rng = np.random.default_rng(seed)
    hi = (1 << bit_depth) - 1

    def plane(w, h, amp):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        base = 0.5 + 0.25 * np.sin(xx / (37.0 + seed % 7)) * np.cos(yy / 53.0) + 0.2 * (xx / max(w, 1) - 0.5)
        lo = rng.standard_normal(((h + 15) // 16 + 1, (w + 15) // 16 + 1)).astype(np.float32)
        lo = np.kron(lo, np.ones((16, 16), np.float32))[:h, :w]
        for ax in (0, 1):  # 9-tap box filter, zero padded ("same")
            pad = [(0, 0), (0, 0)]
            pad[ax] = (5, 4)
            cs = np.cumsum(np.pad(lo.astype(np.float64), pad), axis=ax)
            n = lo.shape[ax]
            hi_s = [slice(None), slice(None)]; lo_s = [slice(None), slice(None)]
            hi_s[ax] = slice(9, 9 + n); lo_s[ax] = slice(0, n)
            lo = ((cs[tuple(hi_s)] - cs[tuple(lo_s)]) / 9.0).astype(np.float32)
        fine = rng.standard_normal((h, w)).astype(np.float32)
        img = base + amp * (0.12 * lo + 0.02 * fine)
        nrect = 4
        for _ in range(nrect):
            x0, y0 = int(rng.integers(0, max(1, w - 8))), int(rng.integers(0, max(1, h - 8)))
            x1, y1 = min(w, x0 + int(rng.integers(8, max(9, w // 3)))), min(h, y0 + int(rng.integers(8, max(9, h // 3))))
            img[y0:y1, x0:x1] += float(rng.uniform(-0.25, 0.25))
        return np.clip(img * hi + 0.5, 0, hi).astype(np.uint16)

    planes = [plane(width, height, 1.0)]
    if chroma_format_idc:
        cw = width if chroma_format_idc == 3 else (width + 1) // 2
        ch = (height + 1) // 2 if chroma_format_idc == 1 else height
        planes += [plane(cw, ch, 0.5), plane(cw, ch, 0.5)]
    return planes


def encode(planes, **kw):
    """Encode planes ([Y] or [Y, Cb, Cr], uint16 arrays at display size) into a plugin-framed stream."""
    prm = dict(ENC_DEFAULTS)
    prm.update(kw)
    h, w = planes[0].shape
    prm.setdefault("width", w)
    prm.setdefault("height", h)
    if len(planes) == 3:   # the chroma format follows from the plane shapes (a one-row picture is 4:2:0 unless the widths say otherwise)
        ch, cw = planes[1].shape
        prm["chroma_format_idc"] = 3 if (cw == w and w > 1) else (2 if (ch == h and h > 1) else 1)
    else:
        prm["chroma_format_idc"] = 0
    st = _EncParams()
    for k, v in prm.items():
        setattr(st, k, int(v))
    L = lib()
    L.hevc_testenc_encode.restype = C.c_int
    ps = [np.ascontiguousarray(p, dtype=np.uint16) for p in planes]
    while len(ps) < 3:
        ps.append(ps[0])
    arr = (C.POINTER(C.c_uint16) * 3)(*[p.ctypes.data_as(C.POINTER(C.c_uint16)) for p in ps])
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    err = C.create_string_buffer(512)
    rc = L.hevc_testenc_encode(C.byref(st), arr, C.byref(out), C.byref(n), err, 512)
    if rc != 0:
        raise OracleError(err.value.decode("latin1"))
    data = bytes(np.ctypeslib.as_array(out, shape=(n.value,)))
    L.hevc_testenc_free.argtypes = [C.POINTER(C.c_uint8)]
    L.hevc_testenc_free(out)
    return data


def encode_sequence(frames, **kw):
    """frames: list of plane lists ([Y] or [Y, Cb, Cr] at display size, all of one shape).  Frame 0 becomes an IDR intra picture, the
    others P pictures.  Returns one plugin-framed access unit per frame (the first carries the parameter sets)."""
    prm = dict(ENC_DEFAULTS)
    prm.update(kw)
    h, w = frames[0][0].shape
    prm.setdefault("width", w)
    prm.setdefault("height", h)
    if len(frames[0]) == 3:   # the chroma format follows from the plane shapes, as in encode()
        ch, cw = frames[0][1].shape
        prm["chroma_format_idc"] = 3 if (cw == w and w > 1) else (2 if (ch == h and h > 1) else 1)
    else:
        prm["chroma_format_idc"] = 0
    st = _EncParams()
    for k, v in prm.items():
        setattr(st, k, int(v))
    L = lib()
    n = len(frames)
    keep = []
    ptrs = (C.POINTER(C.c_uint16) * (3 * n))()
    for f, planes in enumerate(frames):
        ps = [np.ascontiguousarray(p, dtype=np.uint16) for p in planes]
        while len(ps) < 3:
            ps.append(ps[0])
        keep.append(ps)
        for c in range(3):
            ptrs[3 * f + c] = ps[c].ctypes.data_as(C.POINTER(C.c_uint16))
    outs = (C.POINTER(C.c_uint8) * n)()
    sizes = (C.c_size_t * n)()
    err = C.create_string_buffer(512)
    L.hevc_testenc_encode_seq.restype = C.c_int
    rc = L.hevc_testenc_encode_seq(C.byref(st), n, ptrs, outs, sizes, err, 512)
    if rc != 0:
        raise OracleError(err.value.decode("latin1"))
    L.hevc_testenc_free.argtypes = [C.POINTER(C.c_uint8)]
    res = []
    for f in range(n):
        res.append(bytes(np.ctypeslib.as_array(outs[f], shape=(sizes[f],))))
        L.hevc_testenc_free(outs[f])
    return res
