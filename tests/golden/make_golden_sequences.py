"""Generates the committed golden SEQUENCE tracks: seeded synthetic HEVC tracks with P and B pictures from the test generator - the tools of
8.5.3 (merge / AMVP / temporal candidates / weighted prediction / AMP / list modification), long-term reference pictures, constrained intra prediction,
scaling lists in inter pictures - stored as their access units in DECODING order ([u32 BE length][access unit in plugin framing] ..., the first one with
the parameter sets) plus the SHA-256 of every decoded picture's planes by PicOrderCnt as the CPU oracle decodes them.

Who is held to these hashes (tests/test_golden_sequences.py, tests/test_reference_decoder_pin.py): the oracle itself (a regression pin for
hevc_oracle_inter.c), the device code under the CPU emulation, the HIP decoder on the GPU box (no generator needed there: the fixtures travel), and - the
moment an HEVC decoder plugin that is not ours is loadable by the reference libheif - that decoder, through an image-sequence file and
heif_track_decode_next_image(): the pin for inter prediction, which no stream of the reference's own fixtures exercises.
The tracks signal the sRGB nclx in their VUI, so that libheif hands the decoder's planes through unconverted (context.cc:1533-1558).
Run from the repo root:  python tests/golden/make_golden_sequences.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc  # noqa: E402
from test_inter_oracle import make_frames  # noqa: E402

SRGB_VUI = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
# name, width, height, pictures, generator parameters
CASES = [
    ("ippp_2refs_tmvp_weighted", 136, 104, 8, dict(inter_num_refs=2, temporal_mvp=1, weighted_pred=1)),
    ("ippp_amp_multiref_mer_ctb16", 136, 104, 7, dict(amp=1, inter_num_refs=3, max_merge_cand=3, parallel_merge_level=4, log2_ctb=4, log2_max_tb=4)),
    ("ibbp_bref_tmvp_weighted", 136, 104, 9, dict(b_frames=2, b_ref=1, inter_num_refs=2, temporal_mvp=1, weighted_pred=1, mvd_l1_zero=1, amp=1)),
    ("b3_slices_listmod", 136, 104, 9, dict(b_frames=3, temporal_mvp=1, num_slices=2, lists_modification=1, cabac_init_present=1, max_merge_cand=4, wpp=0)),
    ("b_main10_tiles", 136, 104, 7, dict(b_frames=2, b_ref=1, temporal_mvp=1, bit_depth=10, tile_cols=2, tile_rows=2, inter_num_refs=2)),
    ("b_cropped_weighted_70x42", 70, 42, 7, dict(b_frames=1, temporal_mvp=1, weighted_pred=1, global_mv_y=17, inter_num_refs=2)),
    ("b_long_term_ref", 136, 104, 8, dict(b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, long_term_ref=1)),
    ("p_long_term_ref_sps_listmod", 136, 104, 8, dict(temporal_mvp=1, inter_num_refs=3, lists_modification=1, weighted_pred=1, long_term_ref=3)),
    ("b_constrained_intra_pred", 136, 104, 7, dict(b_frames=1, temporal_mvp=1, constrained_intra_pred=1, inter_intra_pct=45, log2_ctb=4, log2_max_tb=4)),
    ("b_scaling_lists_sps", 136, 104, 7, dict(b_frames=1, temporal_mvp=1, scaling_list=2, inter_intra_pct=30)),
    ("p_lossless_tskip_ctb32", 104, 72, 6, dict(lossless_pct=30, amp=1, transform_skip=1, log2_ctb=5)),
    # open GOP: IDR P B B CRA RASL RASL P B B - from the IDR picture on, the CRA picture and its RASL pictures are ordinary pictures
    ("ibbp_open_gop_cra_rasl", 136, 104, 10, dict(b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, open_gop=2)),
]


def pack(aus):
    return b"".join(len(a).to_bytes(4, "big") + a for a in aus)


def unpack(blob):
    out, p = [], 0
    while p < len(blob):
        n = int.from_bytes(blob[p:p + 4], "big")
        out.append(blob[p + 4:p + 4 + n])
        p += 4 + n
    return out


def plane_hashes(planes):
    return [hashlib.sha256(p.astype("<u2").tobytes()).hexdigest() for p in planes]


def main():
    index = {}
    for k, (name, w, h, n, cfg) in enumerate(CASES):
        cfg = dict(cfg, **SRGB_VUI)
        bd = cfg.get("bit_depth", 8)
        frames = make_frames(w, h, n, bd)
        aus = orc.encode_sequence(frames, qp=cfg.pop("qp", 27), global_mv_x=cfg.pop("global_mv_x", -8), global_mv_y=cfg.pop("global_mv_y", -4), inter_skip_pct=20,
                                  seed=300 + k, **cfg)
        pics = orc.decode_sequence(aus)
        blob = pack(aus)
        with open(os.path.join(HERE, "seq_" + name + ".hevcs"), "wb") as f:
            f.write(blob)
        index[name] = {"width": pics[0]["width"], "height": pics[0]["height"], "bit_depth": pics[0]["bit_depth_luma"], "chroma_format_idc": pics[0]["chroma_format_idc"],
                       "samples": len(aus), "stream_sha256": hashlib.sha256(blob).hexdigest(), "coding_order_pocs": [p["poc"] for p in pics],
                       "pictures_sha256": {str(p["poc"]): plane_hashes(p["planes"]) for p in pics}}
        print("%-32s %d samples, %6d bytes, pocs %s" % (name, len(aus), len(blob), index[name]["coding_order_pocs"]))
    with open(os.path.join(HERE, "golden_sequences.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
