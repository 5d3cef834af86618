"""Occupancy statistics of the lane-per-substream parser (libheif_amd/csrc/parse_lanes_kernel.hip) from its CPU emulation: iterations of
a wave's loop, populated syntax states and busy lanes per iteration.  Usage: python tools/lanes_stats.py [n_pictures] [w h] [qp]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import streamgen

STATES = ("S_CTB S_CQT S_SPLIT S_SPLIT_R S_CU S_TQB_R S_PART_R S_PREV_R S_IPM S_MPM1_R S_MPM2_R S_REMMODE_R S_CHROMA_R S_CHROMA2_R "
          "S_TSPLIT_R S_CBFCB_R S_CBFCR_R S_TT S_CBFL_R S_QPD0_R S_QPD1_R S_QPD_EGP_R S_QPD_EGS_R S_QPD_SIGN_R "
          "S_TS_R S_LASTX_R S_LASTY_R S_LASTXS_R S_LASTYS_R S_CSBF_R S_SIG_R S_G1_R S_G2_R S_SIGN_R S_REM_R S_SB S_RES S_EOS_R S_EOS2_R S_DONE").split()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
    qp = int(sys.argv[4]) if len(sys.argv) > 4 else 27
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libparse_emu_stats.so"])
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libparse_emu_stats.so"))
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.emu_run_parse_lanes.argtypes = [C.c_void_p]
    streams = streamgen.make_streams([(w, h, 1000 + i, 8, dict(qp=qp)) for i in range(n)])
    arr = (C.c_char_p * n)(*streams)
    sizes = (C.c_size_t * n)(*[len(s) for s in streams])
    err = C.create_string_buffer(512)
    b = L.emu_create(n, arr, sizes, err, 512)
    assert b, err.value
    L.emu_lanes_stats_reset()
    st = L.emu_run_parse_lanes(b)
    out = (C.c_uint64 * 142)()
    L.emu_lanes_stats_get(out)
    it, busy, wait, live, distinct, mx = out[0:6]
    px = n * w * h
    print("status", st, "pictures", n, "%dx%d" % (w, h), "qp", qp)
    print("wave iterations total %d, longest wave %d; per iteration: live lanes %.1f, waiting %.1f, decoding %.1f, populated states %.1f"
          % (it, mx, live / it, wait / it, busy / it, distinct / it))
    print("lane-steps per pixel %.3f" % (sum(out[70:134]) / px))
    print("state: iterations populated (%% of iterations), lane share")
    tot = sum(out[70:134])
    for i, name in enumerate(STATES):
        if out[6 + i]:
            print("  %-12s %5.1f %%   %5.1f %%" % (name, 100.0 * out[6 + i] / it, 100.0 * out[70 + i] / tot))
    print("request kinds (lane-steps):", dict(zip(["none", "ctx", "bypass", "terminate", "remaining"], [out[134 + i] for i in range(5)])))


if __name__ == "__main__":
    main()
