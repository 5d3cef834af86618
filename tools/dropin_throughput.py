"""Drop-in throughput: T application threads, each calling heif_decode_image() on its own 4K HEIC through the UNMODIFIED reference
libheif (oracle/_ref/libheif.so, built from /root/reference by oracle/Makefile.ref) with libheifhip.so loaded as the decoder plugin —
the usage of /root/reference/tests/test-race.go:73-106 and libheif/image-items/image_item.cc:1276.  Host to host: HEIC bytes in host
memory -> heif_image planes in host memory (YCbCr 4:2:0), so PCIe, libheif's container parsing and plane allocation are all inside.
Reports Gpixel/s, the coalescer's launch sets (decoder instances that shared one set of kernel launches) per thread count.

usage: python tools/dropin_throughput.py [--threads 64,256] [--files 64] [--seconds 6] [--json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(threads_list=(64, 256), n_files=64, seconds=6.0, w=3840, h=2160, qp=27, rgb=False, quiet=True, libheif="libheif.so", direct=False):
    """Runs the C harness (tools/dropin_host.c: pthreads, no Python in the timed loop) once per thread count."""
    import subprocess
    import tempfile
    from tools import streamgen
    import heic_util as hu
    import libheif_amd
    ref = os.path.join(ROOT, "oracle", "_ref", libheif)
    if not os.path.exists(ref):
        raise RuntimeError("oracle/_ref/%s is not built (make -C oracle ref)" % libheif)
    exe = os.path.join(ROOT, "build", "dropin_host")
    src = os.path.join(ROOT, "tools", "dropin_host.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-pthread", src, "-ldl", "-o", exe])
    cfg = dict(wpp=1, qp=qp, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    streams = streamgen.make_streams([(w, h, 5000 + i, 8, cfg) for i in range(n_files)])
    tmp = tempfile.mkdtemp(prefix="hipdec_dropin_")
    paths = []
    for i, s in enumerate(streams):
        pth = os.path.join(tmp, "f%04d.heic" % i)
        with open(pth, "wb") as f:
            f.write(s if direct else hu.build_heic([(s, w, h)]))
        paths.append(pth)
    results = []
    for T in threads_list:
        r = subprocess.run([exe, ref, libheif_amd.library_path(), str(T), str(seconds), "2" if direct else ("1" if rgb else "0")] + paths, capture_output=True, text=True, timeout=seconds * 4 + 120)
        sys.stderr.write(r.stderr)
        if r.returncode != 0:
            results.append({"threads": T, "error": (r.stderr or r.stdout)[-400:]})
            continue
        n, dt, mpx, req, sets, failed, mean_ms, p95_ms = r.stdout.split()[-8:]
        rec = {"threads": T, "decodes": int(n), "seconds": float(dt), "mpixel_s": float(mpx), "decoder_requests": int(req), "launch_sets": int(sets),
               "stills_per_launch_set": round(int(req) / max(1, int(sets)), 1), "call_ms_mean": float(mean_ms), "call_ms_p95": float(p95_ms)}
        results.append(rec)
        if not quiet:
            print(json.dumps(rec), flush=True)
    for pth in paths:
        os.unlink(pth)
    os.rmdir(tmp)
    return {"workload": "%d distinct %dx%d 8-bit 4:2:0 HEIC files (QP %d, WPP), T pthreads x heif_decode_image() -> %s through the real libheif (%s) + plugin, host to host"
                        % (n_files, w, h, qp, "RGB24" if rgb else "YCbCr planes", libheif),
            "coalesce_window_us": int(os.environ.get("HIPDEC_COALESCE_WINDOW_US", "2000")), "runs": results}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="64,256")
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--rgb", action="store_true")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--direct", action="store_true", help="the threads call the plugin's C ABI themselves (no libheif, planes into per-thread buffers): isolates the host side")
    ap.add_argument("--libheif", default="libheif.so", help="build of the reference under oracle/_ref: libheif.so (stock) or libheif_hipcolor.so (HIP colour op)")
    a = ap.parse_args()
    out = measure([int(x) for x in a.threads.split(",")], a.files, a.seconds, rgb=a.rgb, quiet=a.json, libheif=a.libheif, direct=a.direct)
    if a.json:
        print(json.dumps(out))
