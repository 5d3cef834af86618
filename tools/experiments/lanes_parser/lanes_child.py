"""Child process of tests/test_decode_gpu.py::test_lane_parser_on_the_gpu: HIPDEC_PARSE_LANES is read once per process, so the
lane-per-substream parser (libheif_amd/csrc/parse_lanes_kernel.hip) is exercised in a process of its own.  Decodes a batch of small stills of
mixed tools / sizes through it and compares every plane with the oracle."""
import os
import sys
import numpy as np

assert os.environ.get("HIPDEC_PARSE_LANES") == "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as orc
from libheif_amd.decoder import Batch

CFGS = [dict(), dict(wpp=0, stress=1), dict(tile_cols=2, tile_rows=2, wpp=0), dict(num_slices=3), dict(transform_skip=1, stress=1),
        dict(lossless_pct=30), dict(qp=12, stress=1), dict(log2_ctb=4, log2_max_tb=4, stress=1), dict(sign_data_hiding=0, cu_qp_delta=0)]
streams, refs = [], []
for i in range(72):   # more pictures than a wave has lanes: lanes of one wave hold the same row of different pictures
    w, h = [(200, 136), (64, 64), (328, 72), (136, 200)][i % 4]
    s = orc.encode(orc.synth_image(w, h, 8, 1, seed=300 + i), **CFGS[i % len(CFGS)])
    streams.append(s)
    refs.append(orc.decode(s))
b = Batch(streams)
b.run()
b.status()   # raises on a device error
for i, ref in enumerate(refs):
    got = b.planes(i)
    for c in range(3):
        np.testing.assert_array_equal(got[c], ref["planes"][c], err_msg="still %d component %d" % (i, c))
print("lane parser: %d stills bit-exact" % len(refs))
