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


def measure(threads_list=(64, 256), n_files=64, seconds=6.0, w=3840, h=2160, qp=27, rgb=False, quiet=True):
    from tools import streamgen
    import heic_util as hu
    import libheif_host as lh
    from libheif_amd.decoder import coalesce_stats
    cfg = dict(wpp=1, qp=qp, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    streams = streamgen.make_streams([(w, h, 5000 + i, 8, cfg) for i in range(n_files)])
    heics = [hu.build_heic([(s, w, h)]) for s in streams]
    L = lh.load_hip_plugin()
    cs, ch = (lh.COLORSPACE_RGB, lh.CHROMA_RGB) if rgb else (lh.COLORSPACE_YCBCR, lh.CHROMA_420)

    def one(data):
        ctx = L.heif_context_alloc()
        try:
            lh.check(L.heif_context_read_from_memory_without_copy(ctx, data, len(data), None))
            hd = C.c_void_p()
            lh.check(L.heif_context_get_primary_image_handle(ctx, C.byref(hd)))
            img = C.c_void_p()
            try:
                lh.check(L.heif_decode_image(hd, C.byref(img), cs, ch, None))
            finally:
                if img:
                    L.heif_image_release(img)
                L.heif_image_handle_release(hd)
        finally:
            L.heif_context_free(ctx)

    one(heics[0])     # warm-up: HIP runtime, code objects, arena pool
    results = []
    for T in threads_list:
        stop = [False]
        counts = [0] * T
        errors = []
        start_evt = threading.Event()

        def worker(k):
            start_evt.wait()
            i = k
            while not stop[0]:
                try:
                    one(heics[i % len(heics)])
                except Exception as e:   # noqa
                    errors.append(repr(e)); return
                counts[k] += 1
                i += T

        ths = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(T)]
        for t in ths:
            t.start()
        c0 = coalesce_stats()
        t0 = time.perf_counter()
        start_evt.set()
        time.sleep(seconds)
        stop[0] = True
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0      # includes the decodes in flight at `stop`: every counted decode completed inside dt
        c1 = coalesce_stats()
        n = sum(counts)
        r = {"threads": T, "decodes": n, "seconds": round(dt, 3), "mpixel_s": round(n * w * h / dt / 1e6, 1),
             "ms_per_decode_per_thread": round(dt * 1e3 * T / max(1, n), 1),
             "decoder_requests": c1[0] - c0[0], "launch_sets": c1[1] - c0[1],
             "stills_per_launch_set": round((c1[0] - c0[0]) / max(1, c1[1] - c0[1]), 1), "errors": errors[:3]}
        results.append(r)
        if not quiet:
            print(json.dumps(r), flush=True)
    return {"workload": "%d distinct %dx%d 8-bit 4:2:0 HEIC files (QP %d, WPP), T threads x heif_decode_image() -> %s through the real libheif + plugin, host to host"
                        % (n_files, w, h, qp, "RGB24 (colour stage per the loaded libheif build)" if rgb else "YCbCr planes"),
            "coalesce_window_us": int(os.environ.get("HIPDEC_COALESCE_WINDOW_US", "2000")), "runs": results}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="64,256")
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--rgb", action="store_true")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    out = measure([int(x) for x in a.threads.split(",")], a.files, a.seconds, rgb=a.rgb, quiet=a.json)
    if a.json:
        print(json.dumps(out))
