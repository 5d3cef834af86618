import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tools import streamgen
from libheif_amd.decoder import Batch
s = streamgen.make_stream(3840, 2160, seed=1001, bit_depth=8, wpp=1, qp=27)
b = Batch([s]); b.run(); b.status()
b.run(); b.status()
print("timing", b.kernel_timing_us())
