import sys, os, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as orc
from libheif_amd.decoder import Batch, HipDecoder
import libheif_amd

def p(*a):
    print(*a, flush=True)

kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = int(v)
w = kw.pop("w", 200); h = kw.pop("h", 136)
planes = orc.synth_image(w, h, kw.get("bit_depth", 8), 1, seed=3)
stream = orc.encode(planes, **kw)
ref = orc.decode(stream, taps=True)
p("stream", len(stream), "subs", ref["n_substreams"])
b = Batch([stream])
p("batch created", b.info(0))
b.run()
p("run launched")
try:
    b.status()
    p("status ok")
except Exception as e:
    p("status error:", e)
p("timing", b.timing_us())
m = b.maps(0)
for k, rk in (("log2_cb", "map_log2_cb"), ("log2_tb", "map_log2_tb"), ("intra_luma", "map_intra_luma"), ("intra_chroma", "map_intra_chroma"), ("qp_y", "map_qp_y")):
    bad = np.argwhere(m[k] != ref[rk])
    p(k, "mismatches", len(bad), bad[:5].tolist())
bad = np.argwhere((m["flags"] & 0x7f) != (ref["map_flags"] & 0x7f))
p("flags mismatches", len(bad), bad[:5].tolist(), [(int(m["flags"][tuple(x)]), int(ref["map_flags"][tuple(x)])) for x in bad[:5]])
for c in range(3):
    t = b.tap(0, c)
    bad = np.argwhere(t != ref["post_deblock"][c])
    p("deblocked comp", c, "mismatches", len(bad), bad[:5].tolist())
got = b.planes(0)
for c in range(3):
    bad = np.argwhere(got[c] != ref["planes"][c])
    p("final comp", c, "mismatches", len(bad), bad[:5].tolist())
