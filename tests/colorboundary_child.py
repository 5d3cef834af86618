"""Child process of tests/test_color_boundary.py: drives ONE build of the reference libheif (HIPDEC_TEST_LIBHEIF) with libheifhip.so
loaded as decoder plugin, decodes the HEIC files given on the command line to the requested output and stores pixels + the colour
boundary's counters in an .npz.  A process of its own, because the stock and the HIP-colour build of libheif export the same symbols."""
import ctypes as C
import json
import sys
import numpy as np

import libheif_host as lh
import libheif_amd


def main():
    jobs = json.load(open(sys.argv[1]))
    L = lh.load_hip_plugin()
    hip = libheif_amd.load_library()
    hip.hipdec_color_boundary_stats.restype = None
    hip.hipdec_color_boundary_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
    hip.hipdec_image_ops_stats.restype = None
    hip.hipdec_image_ops_stats.argtypes = [C.POINTER(C.c_uint64)] * 2
    hip.hipdec_resident_rgb_stats.restype = None
    hip.hipdec_resident_rgb_stats.argtypes = [C.POINTER(C.c_uint64)] * 2
    out = {}
    for j in jobs:
        data = open(j["heic"], "rb").read()
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        hip.hipdec_color_boundary_stats(C.byref(a), C.byref(b), C.byref(c))
        before = (a.value, b.value, c.value)
        t, g = C.c_uint64(), C.c_uint64()
        hip.hipdec_image_ops_stats(C.byref(t), C.byref(g))
        ops_before = (t.value, g.value)
        rp, rs = C.c_uint64(), C.c_uint64()
        hip.hipdec_resident_rgb_stats(C.byref(rp), C.byref(rs))
        rgb_before = (rp.value, rs.value)
        res = lh.decode(data, j["colorspace"], j["chroma"], max_threads=j.get("threads"))
        hip.hipdec_color_boundary_stats(C.byref(a), C.byref(b), C.byref(c))
        hip.hipdec_image_ops_stats(C.byref(t), C.byref(g))
        if "rgb" in res:
            out[j["name"] + ".rgb"] = res["rgb"]
        else:
            for k, p in enumerate(res["planes"]):
                out[j["name"] + ".plane%d" % k] = p
        out[j["name"] + ".stats"] = np.array([a.value - before[0], b.value - before[1], c.value - before[2]], np.int64)
        hip.hipdec_resident_rgb_stats(C.byref(rp), C.byref(rs))
        out[j["name"] + ".rgbres"] = np.array([rp.value - rgb_before[0], rs.value - rgb_before[1]], np.int64)   # pictures decoded with RGB beside the planes, conversions served from it
        out[j["name"] + ".ops"] = np.array([t.value - ops_before[0], g.value - ops_before[1]], np.int64)   # transforms, grid canvases
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
