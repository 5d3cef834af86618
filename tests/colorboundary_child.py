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
    out = {}
    for j in jobs:
        data = open(j["heic"], "rb").read()
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        hip.hipdec_color_boundary_stats(C.byref(a), C.byref(b), C.byref(c))
        before = (a.value, b.value, c.value)
        res = lh.decode(data, j["colorspace"], j["chroma"], max_threads=j.get("threads"))
        hip.hipdec_color_boundary_stats(C.byref(a), C.byref(b), C.byref(c))
        out[j["name"] + ".rgb"] = res["rgb"]
        out[j["name"] + ".stats"] = np.array([a.value - before[0], b.value - before[1], c.value - before[2]], np.int64)
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
