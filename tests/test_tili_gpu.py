"""the 'tili' tiled-image item (/root/reference/libheif/image-items/tiled.cc:1032-1048) reaches the same decoder plugin
as 'hvc1' items and grids — per tile, with ONE shared decoder configuration ('hvcC' inside the item's 'tilC' property) in front of each
tile's slice data.  A 'tili' file written by tests/heic_util.py:build_tili goes through the REAL libheif (the build with
HEIF_ENABLE_EXPERIMENTAL_FEATURES, the only one that instantiates 'tili' items: image_item.cc:204-208, box.cc:761-765) + libheifhip.so."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

import heic_util as hu
import libheif_host as lh
from oracle import pyoracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)


def _tili(rows, cols, tw, th, **cfg):
    streams = [orc.encode(orc.synth_image(tw, th, cfg.get("bit_depth", 8), 1, seed=70 + i), **dict(CFG, **cfg)) for i in range(rows * cols)]
    return streams, hu.build_tili(streams, rows, cols, tw, th, cols * tw - 4, rows * th - 8, bit_depth=cfg.get("bit_depth", 8))


def test_tili_writer_is_parsed_by_the_reference(tmp_path):
    """CPU: the experimental build of the reference opens the file and reports the item's size (no decoder needed)"""
    if not lh.available("libheif_experimental.so"):
        pytest.skip("oracle/_ref/libheif_experimental.so not built")
    _, f = _tili(2, 3, 128, 64)
    code = "import sys; sys.path.insert(0, %r); import libheif_host as lh; print(lh.primary_size(open(sys.argv[1], 'rb').read()))" % HERE
    p = tmp_path / "t.heic"
    p.write_bytes(f)
    env = dict(os.environ, HIPDEC_TEST_LIBHEIF="libheif_experimental.so", PYTHONPATH=os.pathsep.join([os.path.join(HERE, ".."), HERE]))
    r = subprocess.run([sys.executable, "-c", code, str(p)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.strip() == "(380, 120)"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(bit_depth=10), dict(wpp=0, stress=1)], ids=["main", "main10", "stress"])
def test_tili_tiles_through_libheif_and_the_plugin_match_the_oracle(tmp_path, cfg):
    if not lh.available("libheif_experimental.so"):
        pytest.fail("oracle/_ref/libheif_experimental.so missing on the GPU box")
    rows, cols, tw, th = 2, 3, 128, 64
    streams, f = _tili(rows, cols, tw, th, **cfg)
    path = tmp_path / "tiled.heic"
    path.write_bytes(f)
    tiles = [(0, 0), (2, 1), (1, 0), (1, 1)]
    jf, of = str(tmp_path / "job.json"), str(tmp_path / "out.npz")
    json.dump({"file": str(path), "tiles": tiles}, open(jf, "w"))
    env = dict(os.environ, HIPDEC_TEST_LIBHEIF="libheif_experimental.so", PYTHONPATH=os.pathsep.join([os.path.join(HERE, ".."), HERE]))
    r = subprocess.run([sys.executable, os.path.join(HERE, "tili_child.py"), jf, of], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = np.load(of)
    assert tuple(out["size"]) == (cols * tw - 4, rows * th - 8)
    for tx, ty in tiles:
        assert int(out["decodes_%d_%d" % (tx, ty)][0]) == 1           # exactly one plugin decode per requested tile
        ref = orc.decode(streams[ty * cols + tx])
        for c in range(3):
            np.testing.assert_array_equal(out["t%d_%d_c%d" % (tx, ty, c)], ref["planes"][c], err_msg="tile %d,%d component %d" % (tx, ty, c))
    assert int(out["whole_image_error_code"][0]) != 0                # the reference's own rule: 'tili' is accessed per tile
