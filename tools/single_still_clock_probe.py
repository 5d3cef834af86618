"""One 4K still (latency mode: 34 waves on 256 CUs) with and without other work on the chip: does the lone still run at a lower clock?  (dev tool, GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libheif_amd
from libheif_amd.decoder import Batch
from libheif_amd._capi import check
from tools import streamgen
lib = libheif_amd.load_library()
check(lib.hipdec_init(0))
s = streamgen.make_stream(3840, 2160, seed=1000, bit_depth=8, qp=27, wpp=1)
b = Batch([s]); b.alloc_rgb(10)
b.timing_slots(8)

def one(tag):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); b.run_rgb(); b.status(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-34s wall ms per still: %s" % (tag, " ".join("%.1f" % t for t in ts)), flush=True)

one("alone")
side = torch.cuda.Stream()
x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for load, n in (("bf16 GEMM 4096^3 stream", 400), ("elementwise add stream", 3000)):
    with torch.cuda.stream(side):
        for _ in range(n):
            if load.startswith("bf16"): y = x @ x
            else: y = x + 1
    one("beside " + load)
    torch.cuda.synchronize()
one("alone again")
