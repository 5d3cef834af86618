"""Does a host call block while two batches alternate with hipdec_set_stage_overlap(1)?  Host-side wall time of every run_rgb() call (dev tool, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libheif_amd
from libheif_amd.decoder import Batch
from libheif_amd._capi import check
from tools import streamgen
lib = libheif_amd.load_library()
check(lib.hipdec_init(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
streams = streamgen.make_streams([(3840, 2160, 1000 + i, 8, dict(qp=27, wpp=1)) for i in range(32)])
lib.hipdec_set_stage_overlap(1)
bs = []
for j in range(2):
    b = Batch([streams[(i + j) % 32] for i in range(n)]); b.alloc_rgb(10); bs.append(b)
for b in bs: b.run_rgb()
check(lib.hipdec_stream_synchronize(None))
for b in bs: b.status()
t0 = time.perf_counter()
for step in range(3):
    for j, b in enumerate(bs):
        t1 = time.perf_counter(); b.run_rgb(); t2 = time.perf_counter()
        print("step %d batch %d: run_rgb returned after %.1f ms (at %.1f ms)" % (step, j, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
for b in bs: b.status()
print("all done at %.1f ms" % ((time.perf_counter() - t0) * 1e3))
