"""Randomised sweep of sequence tracks through the DECODER OBJECT of the emulated library (tests/emu/libheifhip_emu.so; no GPU): the random tracks of
tools/emu_random_sweep_inter.py - GOP structures, reference counts, TMVP, weights, long-term references, open GOPs, slices / tiles / WPP, bit depths,
monochrome - plus a hidden picture (pic_output_flag = 0) now and then, pushed the way libheif pushes them (one sample per push, or several, or the whole
track at once) with a random look-ahead; checks what only the decoder object does: access-unit splitting, look-ahead chains, reference holds across chains,
the bumping process (output order, user_data of the coding sample, hidden pictures), flush.
usage: HIPDEC_LIBRARY=tests/emu/libheifhip_emu.so HIPDEC_DEV_AB=1 python tools/emu_random_sweep_decoder.py <seed> <count> [procs]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def run_case(args):
    seed, k = args
    rng = random.Random(seed * 100043 + k)
    import ctypes as C
    import numpy as np
    from emu_random_sweep_inter import random_case
    from oracle import pyoracle as orc
    from test_inter_oracle import make_frames
    import libheif_amd
    from libheif_amd.decoder import HipDecoder
    w, h, mono, n, cfg = random_case(rng)
    if cfg["b_frames"] and not cfg.get("open_gop") and rng.random() < 0.25:
        cfg["hidden_poc"] = rng.randrange(1, n)
    try:
        aus = orc.encode_sequence(make_frames(w, h, n, cfg["bit_depth"], mono, seed=seed + k), seed=seed * 7 + k, **cfg)
    except orc.OracleError as ex:
        return (k, "generator: " + str(ex)[:80])
    refs = orc.decode_sequence(aus)
    by_poc = {r["poc"]: r for r in refs}
    coding = [r["poc"] for r in refs]
    hidden = cfg.get("hidden_poc", 0)
    lib = libheif_amd.load_library()
    lib.hipdec_set_sequence_lookahead.argtypes = [C.c_int]; lib.hipdec_set_sequence_lookahead.restype = None
    lib.hipdec_set_sequence_lookahead(rng.choice([0, 1, 2, 5, 32]))
    d = HipDecoder()
    got = []
    try:
        mode = rng.choice(["one", "one", "some", "all"])
        k0 = 0
        while k0 < len(aus):
            take = 1 if mode == "one" or k0 == 0 else (len(aus) - k0 if mode == "all" else rng.choice([1, 2, 3]))
            d.push_data(b"".join(aus[k0:k0 + take]))
            # (several samples in one push: one user_data for all of them - only checked when every push held one sample)
            r = d.next_picture(user_data=300 + k0)
            while r is not None:
                got.append(r); r = d.next_picture()
            k0 += take
        r = d.next_picture(flush=True)
        while r is not None:
            got.append(r); r = d.next_picture(flush=True)
        shown = [p for p in sorted(by_poc) if not (hidden and p == hidden)]
        if len(got) != len(shown):
            return (k, "MISMATCH %d pictures came out, %d expected (%s)" % (len(got), len(shown), cfg))
        for (img, ud), poc in zip(got, shown):
            if mode == "one" and ud != 300 + coding.index(poc):
                return (k, "MISMATCH user_data %d for PicOrderCnt %d (%s)" % (ud, poc, cfg))
            for c in range(len(by_poc[poc]["planes"])):
                if not (img.planes[c] == by_poc[poc]["planes"][c]).all():
                    return (k, "MISMATCH PicOrderCnt %d plane %d (%s, push mode %s)" % (poc, c, cfg, mode))
    except Exception as ex:   # noqa
        return (k, "MISMATCH %r (%s)" % (ex, cfg))
    finally:
        d.free()
    return (k, "ok")


if __name__ == "__main__":
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(run_case, [(seed, k) for k in range(count)], chunksize=1)
    bad = [r for r in res if r[1].startswith("MISMATCH")]
    print("seed %d: %d cases, %d ok, %d refused by the generator, %d MISMATCHES" % (seed, count, sum(r[1] == "ok" for r in res), sum(r[1].startswith("generator") for r in res), len(bad)))
    for r in bad[:10]:
        print("  ", r)
    sys.exit(1 if bad else 0)
