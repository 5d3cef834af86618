"""Host-side mirror of libheif's colour-conversion entry point for the HEIC hot path.

`convert_colorspace()` follows libheif/color-conversion/colorconversion.cc:490-623: it derives the
input ColorState from the planes + nclx, and picks the operation chain the reference's Dijkstra
planner picks for the in-scope cases, executing it as ONE fused HIP kernel per
chain through the C ABI.  Arguments use the reference's vocabulary (heif_chroma numeric values,
nclx tuples, preferred_chroma_upsampling_algorithm / only_use_preferred_chroma_algorithm).
"""
import ctypes as C
import numpy as np
from ._capi import DeviceBuffer, Nclx, check, load_library, HipDecError

CHROMA_420, CHROMA_422, CHROMA_444 = 1, 2, 3
CHROMA_RGB, CHROMA_RGBA = 10, 11
CHROMA_RRGGBB_BE, CHROMA_RRGGBBAA_BE, CHROMA_RRGGBB_LE, CHROMA_RRGGBBAA_LE = 12, 13, 14, 15
UPSAMPLING_NEAREST, UPSAMPLING_BILINEAR = 1, 2


def _nclx_struct(nclx):
    if nclx is None:
        return Nclx(0, 2, 2, 2, 1)
    return Nclx(1, int(nclx[0]), int(nclx[1]), int(nclx[2]), int(nclx[3]))


def replaced_nclx(nclx):
    """nclx_profile::replace_undefined_values_with_sRGB_defaults (libheif/nclx.cc:360-373) on (primaries, transfer, matrix, full_range)"""
    if nclx is None:
        return (1, 13, 6, 1)
    p, t, m, f = nclx
    return (1 if p == 2 else p, 13 if t == 2 else t, 6 if m == 2 else m, f)


def _planning_matrix(nclx):
    """nclx_profile::replace_undefined_values_with_sRGB_defaults (libheif/nclx.cc:360-373) — only the
    *planner* sees these; the ops read the image's own profile."""
    if nclx is None:
        return 6, 1  # nclx_profile::undefined() -> sRGB defaults: matrix 6, full range
    m = nclx[2]
    return (6 if m == 2 else m), int(nclx[3])


_BILINEAR = {CHROMA_420: "Op_YCbCr420_bilinear_to_YCbCr444", CHROMA_422: "Op_YCbCr422_bilinear_to_YCbCr444"}


def plan(bpp, chroma, nclx, target_chroma, upsampling=UPSAMPLING_BILINEAR, only_preferred=False):
    """Names of the reference ops the planner would chain (cost 11 each -> fewest steps wins)."""
    matrix, full = _planning_matrix(nclx)
    if matrix in (11, 14):
        raise HipDecError(-4, "Unsupported color conversion (matrix_coefficients %d), as in the reference" % matrix)
    nn_allowed = not (only_preferred and upsampling != UPSAMPLING_NEAREST)
    if chroma == 0:     # heif_chroma_monochrome (no alpha plane here): Op_mono_to_RGB24_32, 8-bit only
        if bpp == 8 and target_chroma in (CHROMA_RGB, CHROMA_RGBA):
            return ["Op_mono_to_RGB24_32"]
        raise HipDecError(-4, "this monochrome conversion is left to the stock ops")
    if target_chroma in (CHROMA_RGB, CHROMA_RGBA):
        if bpp > 8:
            # the reference's search ends on one of two chains (tests/test_color_emu.py checks every state against the compiled pipeline):
            # Op_to_sdr_planes first when the 8-bit chain behind it is the 4:2:0 integer op or when the preferred upsampling runs anyway,
            # otherwise the generic op at the input depth and Op_to_sdr_planes on R, G, B
            int_op = chroma == CHROMA_420 and nn_allowed and full and matrix not in (0, 8)
            if int_op or (chroma != CHROMA_444 and not nn_allowed):
                return ["Op_to_sdr_planes"] + plan(8, chroma, nclx, target_chroma, upsampling, only_preferred)
            return ["Op_YCbCr_to_RGB<u16>", "Op_to_sdr_planes", "Op_RGB_to_RGB24_32"]
        if chroma == CHROMA_420 and nn_allowed and full and matrix not in (0, 8):
            return ["Op_YCbCr420_to_RGB24" if target_chroma == CHROMA_RGB else "Op_YCbCr420_to_RGB32"]
        if chroma != CHROMA_444 and not nn_allowed:
            return [_BILINEAR[chroma], "Op_YCbCr_to_RGB<u8>", "Op_RGB_to_RGB24_32"]
        return ["Op_YCbCr_to_RGB<u8>", "Op_RGB_to_RGB24_32"]
    if target_chroma in (CHROMA_RRGGBB_BE, CHROMA_RRGGBB_LE):
        if bpp <= 8:
            raise HipDecError(-4, "8-bit to RRGGBB needs Op_to_hdr_planes, outside the hot path")
        if chroma == CHROMA_420 and nn_allowed and matrix not in (0, 8):
            return ["Op_YCbCr420_to_RRGGBBaa"]
        # everything else: the generic float op on the 16-bit planes, the interleave, the swap for little endian (checked against the compiled
        # reference pipeline for 4:2:0 / 4:2:2 / 4:4:4 in tests/test_color_emu.py)
        chain = [_BILINEAR[chroma]] if (chroma != CHROMA_444 and not nn_allowed) else []
        chain += ["Op_YCbCr_to_RGB<u16>", "Op_RGB_HDR_to_RRGGBBaa_BE"]
        return chain + (["Op_RRGGBBaa_swap_endianness"] if target_chroma == CHROMA_RRGGBB_LE else [])
    if target_chroma == CHROMA_444:
        if chroma == CHROMA_420 and upsampling == UPSAMPLING_BILINEAR:
            return ["Op_YCbCr420_bilinear_to_YCbCr444"]
    raise HipDecError(-4, "conversion outside the HEIC hot path")


class DevicePlanes:
    """Decoded Y/Cb/Cr planes resident in HBM (tight strides)."""

    def __init__(self, y, cb, cr, bpp):
        dt = np.uint16 if bpp > 8 else np.uint8
        self.bpp = bpp
        self.shape = y.shape
        self.cshape = cb.shape
        self.esize = 2 if bpp > 8 else 1
        self.bufs = [DeviceBuffer.from_numpy(np.ascontiguousarray(p, dtype=dt)) for p in (y, cb, cr)]


def convert_colorspace(planes, bpp, chroma, nclx, target_chroma, upsampling=UPSAMPLING_BILINEAR, only_preferred=False):
    """planes: [Y, Cb, Cr] numpy arrays (host) — uploaded, converted on the GPU, result downloaded.
    Returns a numpy array: (h, w*bytes_per_pixel) uint8 for interleaved targets, or a list of three
    planes for CHROMA_444."""
    lib = load_library()
    steps = plan(bpp, chroma, nclx, target_chroma, upsampling, only_preferred)
    h, w = planes[0].shape
    dp = DevicePlanes(planes[0], planes[1], planes[2], bpp)
    ns = _nclx_struct(nclx)
    es = dp.esize
    yb, cbb, crb = dp.bufs
    ys, cs = w * es, dp.cshape[1] * es
    keep = [dp]
    if steps and steps[0] == "Op_to_sdr_planes":
        new = []
        for buf, shp in ((yb, dp.shape), (cbb, dp.cshape), (crb, dp.cshape)):
            o = DeviceBuffer(shp[0] * shp[1])
            check(lib.hipdec_color_to_sdr(buf.ptr, shp[1] * 2, shp[1], shp[0], bpp, o.ptr, shp[1], None))
            new.append(o)
        yb, cbb, crb = new
        keep.append(new)
        bpp, es = 8, 1
        ys, cs = w, dp.cshape[1]
        steps = steps[1:]
    cur_chroma = chroma
    if steps and steps[0] in _BILINEAR.values():
        new = []
        up = lib.hipdec_color_bilinear_420_to_444 if steps[0] == _BILINEAR[CHROMA_420] else lib.hipdec_color_bilinear_422_to_444
        for buf in (cbb, crb):
            o = DeviceBuffer(w * h * es)
            check(up(buf.ptr, cs, w, h, bpp, o.ptr, w * es, None))
            new.append(o)
        cbb, crb = new
        keep.append(new)
        cs = w * es
        cur_chroma = CHROMA_444
        steps = steps[1:]
        if not steps:
            check(lib.hipdec_stream_synchronize(None))
            dt = np.uint16 if bpp > 8 else np.uint8
            return [yb.to_numpy((h, w), dt), cbb.to_numpy((h, w), dt), crb.to_numpy((h, w), dt)]
    name = steps[0]
    if len(keep) > 1:
        # an op that is not the first of its chain reads the profile the pipeline attached to the intermediate image: the ColorState's, with
        # unspecified values replaced by the sRGB defaults (colorconversion.cc:475, nclx.cc:360-373)
        ns = _nclx_struct(replaced_nclx(nclx))
    if name in ("Op_YCbCr420_to_RGB24", "Op_YCbCr420_to_RGB32"):
        alpha = name.endswith("32")
        bppx = 4 if alpha else 3
        out = DeviceBuffer(w * h * bppx)
        check(lib.hipdec_color_420_to_rgb24(yb.ptr, ys, cbb.ptr, cs, crb.ptr, cs, w, h, C.byref(ns), out.ptr, w * bppx, int(alpha), None))
    elif name == "Op_YCbCr_to_RGB<u8>":
        alpha = target_chroma == CHROMA_RGBA
        bppx = 4 if alpha else 3
        out = DeviceBuffer(w * h * bppx)
        check(lib.hipdec_color_ycbcr_to_rgb24_float(yb.ptr, ys, cbb.ptr, cs, crb.ptr, cs, w, h, cur_chroma, C.byref(ns), out.ptr, w * bppx, int(alpha), None))
    elif name == "Op_YCbCr420_to_RRGGBBaa":
        bppx = 6
        out = DeviceBuffer(w * h * 6)
        check(lib.hipdec_color_420_to_rrggbb(yb.ptr, ys, cbb.ptr, cs, crb.ptr, cs, w, h, bpp, C.byref(ns), out.ptr, w * 6,
                                             int(target_chroma == CHROMA_RRGGBB_LE), None))
    elif name == "Op_YCbCr_to_RGB<u16>" and len(steps) > 1 and steps[1] == "Op_to_sdr_planes":
        alpha = target_chroma == CHROMA_RGBA
        bppx = 4 if alpha else 3
        out = DeviceBuffer(w * h * bppx)
        check(lib.hipdec_color_hdr_to_rgb24(yb.ptr, ys, cbb.ptr, cs, crb.ptr, cs, w, h, bpp, cur_chroma, C.byref(ns), out.ptr, w * bppx, int(alpha), 0, None))
    elif name == "Op_YCbCr_to_RGB<u16>":
        bppx = 6
        out = DeviceBuffer(w * h * 6)
        check(lib.hipdec_color_ycbcr_to_rrggbb_float(yb.ptr, ys, cbb.ptr, cs, crb.ptr, cs, w, h, bpp, cur_chroma, C.byref(ns), out.ptr, w * 6,
                                                     int(target_chroma == CHROMA_RRGGBB_LE), None))
    else:
        raise HipDecError(-4, "unplanned op " + name)
    check(lib.hipdec_stream_synchronize(None))
    res = out.to_numpy((h, w * bppx), np.uint8)
    del keep
    return res
