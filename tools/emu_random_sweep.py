"""Randomised sweep of the whole emulated device pipeline against the oracle (CPU only): random picture sizes and random combinations of the
generator's coding tools (WPP / tiles / slices / dependent segments, PCM, lossless CUs, transform skip, scaling lists, bit depth, CTB / TB sizes,
QP, filters), every decoded plane compared bit by bit; both parser schedulings.  usage: python tools/emu_random_sweep.py <seed> <count> [procs]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_case(rng):
    log2_ctb = rng.choice([4, 5, 6, 6])
    cfg = dict(log2_ctb=log2_ctb, log2_min_cb=rng.choice([3, min(4, log2_ctb)]), qp=rng.choice([4, 12, 22, 27, 34, 40, 48]))
    cfg["log2_max_tb"] = rng.choice([x for x in (3, 4, 5) if x <= log2_ctb and x >= 2] or [log2_ctb])
    cfg["max_transform_hierarchy_depth_intra"] = rng.choice([d for d in (0, 1, 2, 3) if d <= log2_ctb - 2])   # 7.4.3.2.1: at most CtbLog2SizeY - MinTbLog2SizeY
    cfg["wpp"] = rng.choice([0, 1])
    if rng.random() < 0.3:
        cfg["tile_cols"], cfg["tile_rows"] = rng.choice([1, 2, 3]), rng.choice([1, 2])
        cfg["loop_filter_across_tiles"] = rng.choice([0, 1])
    if rng.random() < 0.4:
        cfg["num_slices"] = rng.choice([2, 3, 4])
        cfg["loop_filter_across_slices"] = rng.choice([0, 1])
    if rng.random() < 0.35 and cfg.get("tile_cols", 1) * cfg.get("tile_rows", 1) == 1:
        cfg["dependent_segments"] = rng.choice([2, 3, 5])
    if rng.random() < 0.3:
        cfg["pcm_pct"] = rng.choice([10, 30, 60])
        cfg["pcm_loop_filter_disabled"] = rng.choice([0, 1])
    if rng.random() < 0.3:
        cfg["lossless_pct"] = rng.choice([10, 40, 100])
    cfg["transform_skip"] = rng.choice([0, 0, 1])
    cfg["scaling_list"] = rng.choice([0, 0, 0, 1, 2, 3])
    cfg["bit_depth"] = rng.choice([8, 8, 10, 12])
    cfg["stress"] = rng.choice([0, 1])
    cfg["sao"] = rng.choice([0, 1, 1])
    cfg["deblock_disable"] = rng.choice([0, 0, 1])
    cfg["sign_data_hiding"] = rng.choice([0, 1])
    cfg["cu_qp_delta"] = rng.choice([0, 1])
    if cfg["cu_qp_delta"]:
        cfg["diff_cu_qp_delta_depth"] = rng.choice([0, 1, min(2, log2_ctb - cfg["log2_min_cb"])])
    cfg["strong_intra_smoothing"] = rng.choice([0, 1])
    cfg["cb_qp_offset"], cfg["cr_qp_offset"] = rng.choice([0, 0, 3, -5]), rng.choice([0, 0, -4, 6])
    cfg["zero_residual_pct"] = rng.choice([0, 0, 30])
    w, h = rng.choice([8, 24, 64, 72, 136, 200, 264, 328]), rng.choice([8, 16, 41, 64, 72, 136, 200])
    cf = rng.choice([1, 1, 1, 0])
    if cf == 0 and rng.random() < 0.5:
        w, h = w - rng.choice([0, 1, 3]), h - rng.choice([0, 1, 5])     # odd sizes (the generator only takes them for 4:0:0)
    elif h & 1:
        h += 1
    # 4:4:4 for a quarter and 4:2:2 for a fifth of the colour cases, drawn from a generator of its own so that the other dimensions of a seed stay
    # what they were
    if cf == 1:
        r = random.Random(rng.random()).random()
        if r < 0.25:
            cf = 3
        elif r < 0.45:
            cf = 2
    return w, h, cf, cfg


def run_case(args):
    seed, pool = args
    import numpy as np
    from oracle import pyoracle as orc
    import test_pipeline_emu as tpe
    rng = random.Random(seed)
    w, h, cf, cfg = random_case(rng)
    os.environ["HIPDEC_PARSE_POOL"] = str(pool)
    try:
        planes = orc.synth_image(w, h, cfg["bit_depth"], cf, seed=seed)
        stream = orc.encode(planes, **cfg)
    except Exception as e:   # a combination the generator refuses
        return seed, "skip", str(e)[:80]
    try:
        got = tpe.decode_emu([stream])[0]
        ref = orc.decode(stream)
        if len(got) != len(ref["planes"]):
            return seed, "FAIL", "plane count"
        for c in range(len(got)):
            if not np.array_equal(got[c], ref["planes"][c]):
                return seed, "FAIL", "component %d differs: %s %dx%d cf=%d pool=%d" % (c, cfg, w, h, cf, pool)
        return seed, "ok", ""
    except AssertionError as e:
        return seed, "FAIL", "%s: %s %dx%d cf=%d pool=%d" % (str(e)[:60], cfg, w, h, cf, pool)


if __name__ == "__main__":
    seed0, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as p:
        res = p.map(run_case, [(seed0 + i, i & 1) for i in range(count)], chunksize=1)
    bad = [r for r in res if r[1] == "FAIL"]
    print("cases", len(res), "ok", sum(1 for r in res if r[1] == "ok"), "skipped", sum(1 for r in res if r[1] == "skip"), "FAILED", len(bad))
    for r in bad[:20]:
        print(r)
    for r in [r for r in res if r[1] == "skip"][:5]:
        print(r)
    sys.exit(1 if bad else 0)
