"""Randomised sweep of P / B sequences through the emulated device pipeline against the oracle (CPU only): random picture sizes, GOP structures
(IPPP, B pictures between anchors, B pictures as references), numbers of reference pictures, TMVP, explicit weights, list modification, AMP,
merge-candidate limits, parallel merge levels, slices / tiles / WPP / dependent segments, lossless / PCM / transform-skip units, bit depths,
monochrome; the motion field (both lists) and every plane of every picture compared bit by bit.
usage: python tools/emu_random_sweep_inter.py <seed> <count> [procs] [tracks]     ("tracks": several random tracks side by side, their chains in one batch)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_case(rng):
    log2_ctb = rng.choice([4, 5, 6, 6])
    cfg = dict(log2_ctb=log2_ctb, log2_min_cb=rng.choice([3, min(4, log2_ctb)]), qp=rng.choice([4, 16, 24, 30, 38]))
    cfg["log2_max_tb"] = rng.choice([x for x in (3, 4, 5) if x <= log2_ctb] or [log2_ctb])
    cfg["max_transform_hierarchy_depth_inter"] = rng.choice([d for d in (0, 1, 2) if d <= log2_ctb - 2])
    cfg["wpp"] = rng.choice([0, 1])
    if rng.random() < 0.3:
        cfg["tile_cols"], cfg["tile_rows"] = rng.choice([1, 2, 3]), rng.choice([1, 2])
        cfg["loop_filter_across_tiles"] = rng.choice([0, 1])
    if rng.random() < 0.4:
        cfg["num_slices"] = rng.choice([2, 3])
        cfg["loop_filter_across_slices"] = rng.choice([0, 1])
    if rng.random() < 0.25 and cfg.get("tile_cols", 1) * cfg.get("tile_rows", 1) == 1:
        cfg["dependent_segments"] = rng.choice([2, 3])
    if rng.random() < 0.2:
        cfg["pcm_pct"] = rng.choice([10, 30])
    if rng.random() < 0.25:
        cfg["lossless_pct"] = rng.choice([10, 40, 100])
    cfg["transform_skip"] = rng.choice([0, 0, 1])
    cfg["bit_depth"] = rng.choice([8, 8, 10, 12])
    cfg["stress"] = rng.choice([0, 1])
    cfg["sao"] = rng.choice([0, 1, 1])
    cfg["deblock_disable"] = rng.choice([0, 0, 0, 1])
    cfg["cu_qp_delta"] = rng.choice([0, 1])
    cfg["amp"] = rng.choice([0, 1]) if cfg["log2_min_cb"] < log2_ctb or log2_ctb >= 4 else 0
    cfg["inter_num_refs"] = rng.choice([1, 2, 3, 4])
    cfg["inter_skip_pct"] = rng.choice([0, 20, 50])
    cfg["inter_intra_pct"] = rng.choice([0, 10, 40])
    cfg["inter_merge_pct"] = rng.choice([10, 40, 90])
    cfg["max_merge_cand"] = rng.choice([1, 2, 3, 4, 5])
    cfg["parallel_merge_level"] = rng.choice([2, 2, 3, 4, log2_ctb])
    cfg["cabac_init_present"] = rng.choice([0, 1])
    cfg["lists_modification"] = rng.choice([0, 1])
    cfg["b_frames"] = rng.choice([0, 0, 1, 2, 3])
    cfg["b_ref"] = rng.choice([0, 1])
    cfg["inter_bi_pct"] = rng.choice([20, 50, 90])
    cfg["temporal_mvp"] = rng.choice([0, 1, 1])
    cfg["weighted_pred"] = rng.choice([0, 0, 1])
    cfg["mvd_l1_zero"] = rng.choice([0, 1])
    cfg["global_mv_x"], cfg["global_mv_y"] = rng.choice([0, -8, 6, 21]), rng.choice([0, -4, 10, -17])
    # round 5: long-term reference pictures (three syntax forms), constrained intra prediction, scaling lists in P / B pictures
    cfg["long_term_ref"] = rng.choice([0, 0, 1, 2, 3])
    cfg["constrained_intra_pred"] = rng.choice([0, 0, 1])
    cfg["scaling_list"] = rng.choice([0, 0, 0, 1, 2, 3])
    w, h = rng.choice([16, 40, 64, 72, 136, 200]), rng.choice([16, 24, 42, 64, 72, 104])
    mono = rng.random() < 0.15
    if not mono and (h & 1):
        h += 1
    n = rng.choice([3, 4, 6, 8])
    # an open GOP (a CRA picture with RASL pictures in the middle of the track) where the structure allows one
    if cfg["b_frames"] and not cfg["long_term_ref"] and rng.random() < 0.4:
        k = rng.choice([1, 2])
        if (cfg["b_frames"] + 1) * k < n:
            cfg["open_gop"] = k
    return w, h, mono, n, cfg


def run_case(args):
    seed, k = args
    rng = random.Random(seed * 100003 + k)
    w, h, mono, n, cfg = random_case(rng)
    from oracle import pyoracle as orc
    from test_inter_oracle import make_frames
    import test_inter_emu as T
    try:
        frames = make_frames(w, h, n, cfg["bit_depth"], mono, seed=seed + k)
        aus = orc.encode_sequence(frames, seed=seed * 7 + k, **cfg)
    except orc.OracleError as ex:
        return (k, "generator: " + str(ex)[:80], None)
    try:
        T.check_sequence(aus, "case %d" % k, chain=rng.choice([0, 0, 2, 5, 16]))      # per picture, or the decoder's look-ahead chains
    except Exception as ex:   # noqa
        return (k, "MISMATCH " + str(ex)[:300], (w, h, mono, n, cfg))
    return (k, "ok", None)


def run_tracks_case(args):
    """several random tracks (one bit-depth class: 8-bit and deeper tracks never share a launch set) decoded side by side: their chains in ONE batch
    per round (layout_batch_plan_chains), random chain lengths per track"""
    seed, k = args
    rng = random.Random(seed * 100019 + k)
    from oracle import pyoracle as orc
    from test_inter_oracle import make_frames
    import test_inter_emu as T
    depth = rng.choice([8, 8, 10])
    tracks, chains, names, cfgs = [], [], [], []
    for t in range(rng.choice([2, 3, 4, 6])):
        w, h, mono, n, cfg = random_case(rng)
        cfg["bit_depth"] = depth
        try:
            frames = make_frames(w, h, n, depth, mono, seed=seed + 31 * k + t)
            tracks.append(orc.encode_sequence(frames, seed=seed * 7 + 13 * k + t, **cfg))
        except orc.OracleError:
            continue
        chains.append(rng.choice([1, 2, 3, 5, 16])); names.append("case %d track %d" % (k, t)); cfgs.append((w, h, mono, n, cfg))
    if len(tracks) < 2:
        return (k, "generator: fewer than two tracks", None)
    try:
        T._check_tracks(tracks, chains, names)
    except Exception as ex:   # noqa
        return (k, "MISMATCH " + str(ex)[:300], (chains, cfgs))
    return (k, "ok", None)


if __name__ == "__main__":
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    if len(sys.argv) > 4 and sys.argv[4] == "tracks":
        run_case = run_tracks_case      # noqa: F811 - the multi-track form of the sweep
    import multiprocessing as mp
    from test_parse_emu import emu
    emu()   # build once before the workers start
    with mp.Pool(procs) as pool:
        res = pool.map(run_case, [(seed, k) for k in range(count)], chunksize=1)
    bad = [r for r in res if r[1].startswith("MISMATCH")]
    skipped = [r for r in res if r[1].startswith("generator")]
    print("seed %d: %d cases, %d ok, %d refused by the generator, %d MISMATCHES" % (seed, count, sum(r[1] == "ok" for r in res), len(skipped), len(bad)))
    for r in skipped[:5]:
        print("  refused:", r[0], r[1])
    for r in bad[:10]:
        print("  ", r)
    sys.exit(1 if bad else 0)
