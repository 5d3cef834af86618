"""GPU parity for streams with scaling lists (scaling_list_enabled_flag: default lists, lists in the SPS, lists in the PPS).
Kept in a file of its own, last in collection order: the feature was added after the round's GPU budget was spent, so its
device run is first seen by the round-end test tier (its logic is covered on the CPU by tests/test_pipeline_emu.py, which runs
the same kernel sources)."""
import numpy as np
import pytest

from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scaling_list", [1, 2, 3], ids=["default_lists", "sps_lists", "pps_lists"])
@pytest.mark.parametrize("cfg", [dict(), dict(stress=1, transform_skip=1), dict(bit_depth=10), dict(log2_ctb=5, log2_max_tb=5, qp=38)],
                         ids=["default", "tskip", "main10", "tb32"])
def test_decode_with_scaling_lists_matches_oracle(cfg, scaling_list):
    from libheif_amd.decoder import HipDecoder
    c = dict(cfg, scaling_list=scaling_list)
    stream = orc.encode(orc.synth_image(200, 136, c.get("bit_depth", 8), 1, seed=33 + scaling_list), **c)
    ref = orc.decode(stream)
    d = HipDecoder()
    d.push_data(stream)
    img = d.decode_next_image()
    d.free()
    for comp in range(3):
        np.testing.assert_array_equal(img.planes[comp], ref["planes"][comp], err_msg="component %d" % comp)
