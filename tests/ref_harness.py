"""Test tooling: ctypes binding of oracle/_ref/libref_harness.so — the REAL reference colour
pipeline (libheif convert_colorspace) compiled from /root/reference by oracle/Makefile.ref.
The prebuilt .so travels to the GPU box; /root/reference itself is never read at test time."""
import ctypes as C
import os
import numpy as np

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref")
_LIB = None

CS_YCBCR, CS_RGB = 0, 1
CH_420, CH_422, CH_444 = 1, 2, 3
CH_RGB, CH_RGBA, CH_RRGGBB_BE, CH_RRGGBBAA_BE, CH_RRGGBB_LE, CH_RRGGBBAA_LE = 10, 11, 12, 13, 14, 15
UPS_NN, UPS_BILINEAR = 1, 2


def available():
    return os.path.exists(os.path.join(_REF, "libref_harness.so"))


def lib():
    global _LIB
    if _LIB is None:
        C.CDLL(os.path.join(_REF, "libheif.so"), mode=C.RTLD_GLOBAL)
        _LIB = C.CDLL(os.path.join(_REF, "libref_harness.so"))
        _LIB.ref_convert_colorspace.restype = C.c_int
    return _LIB


CS_MONOCHROME = 2


def convert(planes, bpp, in_chroma, nclx, target_colorspace, target_chroma, out_bpp=0,
            upsampling=UPS_BILINEAR, only_preferred=False, in_colorspace=CS_YCBCR):
    """planes: [Y, Cb, Cr] numpy arrays.  nclx: (primaries, transfer, matrix, full_range) or None.
    Returns a list of numpy arrays (uint8 rows for interleaved, uint8/uint16 planes otherwise)."""
    h, w = planes[0].shape
    dt = np.uint16 if bpp > 8 else np.uint8
    ins = [np.ascontiguousarray(p, dtype=dt) for p in planes]
    ptrs = (C.c_void_p * len(ins))(*[p.ctypes.data for p in ins])
    outs = [np.zeros(w * h * 8 + 64, np.uint8) for _ in range(4)]
    optrs = (C.c_void_p * 4)(*[o.ctypes.data for o in outs])
    info = (C.c_int * 12)()
    n = lib().ref_convert_colorspace(w, h, bpp, in_colorspace, in_chroma, ptrs, len(ins),
                                     int(nclx is not None), *(nclx if nclx else (2, 2, 2, 1)),
                                     target_colorspace, target_chroma, out_bpp, upsampling, int(only_preferred),
                                     optrs, info)
    if n < 0:
        raise RuntimeError("reference convert_colorspace failed with libheif error code %d" % -n)
    res = []
    for i in range(n):
        pw, ph, row = info[i * 3], info[i * 3 + 1], info[i * 3 + 2]
        a = outs[i][:ph * row].reshape(ph, row)
        if row == pw * 2 and target_chroma < 10:
            a = a.view(np.uint16)
        res.append(a.copy())
    return res
