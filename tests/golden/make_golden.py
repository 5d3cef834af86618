"""Generates the committed golden vectors: seeded synthetic HEVC intra streams (test encoder) and the
SHA-256 of the planes the CPU oracle decodes from them, plus — the independent inputs — the plugin-framed HEVC
streams ([u32 BE length][NAL]..., what libheif hands to push_data2, libheif/codecs/decoder.cc:275-308) of every HEVC
item in the reference's own x265-coded fixtures and fuzz corpus:
  ref_*.hevc        items the oracle decodes (examples/example.heic, tests/data/*.heic, fuzzing/data/corpus/*.hei[cf]),
                    with the oracle's plane hashes;
  ref_reject_*.hevc corpus items the oracle refuses (deliberately malformed fuzz cases): the HIP front end must refuse them
                    too or fail loudly on the device — never hang.
The GPU box has no /root/reference, so these extracted streams are what puts real-encoder bitstreams through the compiled
gfx950 kernels (tests/test_golden.py).  Run from the repo root in the build container:  python tests/golden/make_golden.py"""
import glob
import re
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
    # chroma formats of round 3 (a sixth element: chroma_format_idc)
    ("c422_main10_200x136", 200, 136, 10, dict(bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16, stress=1), 2),
    ("c422_scaling_cropped_70x41", 70, 41, 8, dict(scaling_list=2, transform_skip=1), 2),
    ("c444_stress_depth4_136x72", 136, 72, 8, dict(stress=1, log2_ctb=6, log2_min_cb=3, log2_min_tb=2, log2_max_tb=5, max_transform_hierarchy_depth_intra=4), 3),
    ("c444_pcm_lossless_scaling_128x72", 128, 72, 8, dict(pcm_pct=20, lossless_pct=20, scaling_list=1, qp=34), 3),
]


def plane_hashes(planes):
    return [hashlib.sha256(p.astype("<u2").tobytes()).hexdigest() for p in planes]


def main():
    index = {"streams": {}, "reference_fixtures": {}}
    for case in CASES:
        name, w, h, bd, cfg = case[:5]
        stream = orc.encode(orc.synth_image(w, h, bd, case[5] if len(case) > 5 else 1, seed=11), **cfg)
        with open(os.path.join(HERE, name + ".hevc"), "wb") as f:
            f.write(stream)
        ref = orc.decode(stream)
        index["streams"][name] = {"width": ref["width"], "height": ref["height"], "bit_depth": ref["bit_depth_luma"], "nclx": list(ref["nclx"]),
                                  "stream_sha256": hashlib.sha256(stream).hexdigest(), "planes_sha256": plane_hashes(ref["planes"])}
    refdir = "/root/reference"
    index["reference_streams"] = {}
    index["reference_rejects"] = {}
    if os.path.isdir(refdir):
        from heic_util import HeicFile
        for old in glob.glob(os.path.join(HERE, "ref_*.hevc")):
            os.remove(old)
        rels = ["examples/example.heic", "tests/data/rainbow-451x461.heic", "tests/data/with-alpha-512x512.heic"]
        rels += sorted(os.path.relpath(p, refdir) for p in glob.glob(os.path.join(refdir, "fuzzing/data/corpus/*.hei[cf]")))
        seen = set()
        per_file = {}
        for rel in rels:
            try:
                f = HeicFile(os.path.join(refdir, rel))
                items = f.hevc_items()
            except Exception:
                continue                      # not an ISOBMFF/HEVC file this test-side box reader handles
            for iid in items:
                try:
                    stream = f.plugin_stream(iid)
                except Exception:
                    continue
                sha = hashlib.sha256(stream).hexdigest()
                if sha in seen:
                    continue
                seen.add(sha)
                stem = re.sub(r"[^A-Za-z0-9]+", "_", os.path.splitext(os.path.basename(rel))[0]).strip("_")[-40:]
                try:
                    ref = orc.decode(stream)
                except orc.OracleError as e:
                    per_file[rel] = per_file.get(rel, 0) + 1
                    if len(stream) > 4096 or per_file[rel] > 3:       # the corpus repeats near-identical items: three per file
                        continue
                    name = "ref_reject_%s_%d" % (stem, iid)
                    index["reference_rejects"][name] = {"source": "%s#%d" % (rel, iid), "stream_sha256": sha, "oracle_error": str(e)[:120]}
                else:
                    name = "ref_%s_%d" % (stem, iid)
                    index["reference_streams"][name] = {"source": "%s#%d" % (rel, iid), "width": ref["width"], "height": ref["height"],
                                                        "bit_depth": ref["bit_depth_luma"], "chroma_format_idc": ref["chroma_format_idc"],
                                                        "nclx": list(ref["nclx"]), "n_substreams": ref["n_substreams"],
                                                        "stream_sha256": sha, "planes_sha256": plane_hashes(ref["planes"])}
                with open(os.path.join(HERE, name + ".hevc"), "wb") as out:
                    out.write(stream)
        for rel in ("examples/example.heic", "tests/data/rainbow-451x461.heic", "tests/data/with-alpha-512x512.heic"):
            f = HeicFile(os.path.join(refdir, rel))
            for iid in f.hevc_items():
                ref = orc.decode(f.plugin_stream(iid))
                index["reference_fixtures"]["%s#%d" % (rel, iid)] = {"width": ref["width"], "height": ref["height"],
                                                                     "planes_sha256": plane_hashes(ref["planes"])}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print("wrote %d synthetic streams, %d reference streams, %d reference rejects, %d reference fixture hashes" %
          (len(index["streams"]), len(index["reference_streams"]), len(index["reference_rejects"]), len(index["reference_fixtures"])))


if __name__ == "__main__":
    main()
