"""Generates the committed golden vectors: seeded synthetic HEVC intra streams (test encoder) and the
SHA-256 of the planes the CPU oracle decodes from them, plus the hashes of the oracle's decode of the
reference's own HEVC fixtures (only hashes: the .heic files stay in /root/reference).
Run from the repo root in the build container:  python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc  # noqa: E402

CASES = [
    ("default_200x136", 200, 136, 8, dict()),
    ("stress_ctb16_136x72", 136, 72, 8, dict(stress=1, wpp=0, log2_ctb=4, log2_max_tb=4)),
    ("tiles3x2_wpp_264x136", 264, 136, 8, dict(tile_cols=3, tile_rows=2, wpp=1, loop_filter_across_tiles=0)),
    ("slices3_200x136", 200, 136, 8, dict(num_slices=3, loop_filter_across_slices=0)),
    ("tskip_lossless_128x128", 128, 128, 8, dict(transform_skip=1, lossless_pct=20, stress=1)),
    ("main10_bt2020pq_200x136", 200, 136, 10, dict(bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)),
    ("qp12_highrate_128x72", 128, 72, 8, dict(qp=12, stress=1)),
    ("cropped_70x42", 70, 42, 8, dict()),
]


def plane_hashes(planes):
    return [hashlib.sha256(p.astype("<u2").tobytes()).hexdigest() for p in planes]


def main():
    index = {"streams": {}, "reference_fixtures": {}}
    for name, w, h, bd, cfg in CASES:
        stream = orc.encode(orc.synth_image(w, h, bd, 1, seed=11), **cfg)
        with open(os.path.join(HERE, name + ".hevc"), "wb") as f:
            f.write(stream)
        ref = orc.decode(stream)
        index["streams"][name] = {"width": ref["width"], "height": ref["height"], "bit_depth": ref["bit_depth_luma"], "nclx": list(ref["nclx"]),
                                  "stream_sha256": hashlib.sha256(stream).hexdigest(), "planes_sha256": plane_hashes(ref["planes"])}
    refdir = "/root/reference"
    if os.path.isdir(refdir):
        from heic_util import HeicFile
        for rel in ("examples/example.heic", "tests/data/rainbow-451x461.heic", "tests/data/with-alpha-512x512.heic"):
            f = HeicFile(os.path.join(refdir, rel))
            for iid in f.hevc_items():
                ref = orc.decode(f.plugin_stream(iid))
                index["reference_fixtures"]["%s#%d" % (rel, iid)] = {"width": ref["width"], "height": ref["height"],
                                                                     "planes_sha256": plane_hashes(ref["planes"])}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print("wrote %d streams, %d reference fixture hashes" % (len(index["streams"]), len(index["reference_fixtures"])))


if __name__ == "__main__":
    main()
