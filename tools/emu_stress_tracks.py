"""Stress of the sequence path's HOST logic on the emulated library (tests/emu/libheifhip_emu.so; no GPU): many tracks decoded side by side by threads with
random pauses between pushes, random look-ahead, tracks of different structure / size / length, some corrupt - the chain coalescer must neither deadlock nor
mix tracks up: every picture of every good track equals the oracle's in output order with its own user_data, every damaged track fails alone.
usage: HIPDEC_LIBRARY=tests/emu/libheifhip_emu.so HIPDEC_DEV_AB=1 python tools/emu_stress_tracks.py <seed> <rounds> [threads]"""
import os, random, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
from oracle import pyoracle as orc
from test_inter_oracle import make_frames
import libheif_amd
from libheif_amd import HipDecError
from libheif_amd.decoder import HipDecoder, chain_stats

seed, rounds = int(sys.argv[1]), int(sys.argv[2])
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = random.Random(seed)
lib = libheif_amd.load_library()
lib.hipdec_set_sequence_lookahead.argtypes = [C.c_int]
lib.hipdec_set_sequence_lookahead.restype = None

SPECS = [dict(n=9, temporal_mvp=1, weighted_pred=1, inter_num_refs=3), dict(n=12, b_frames=2, b_ref=1, inter_num_refs=2, temporal_mvp=1),
         dict(n=7, w=136, h=104, b_frames=1, temporal_mvp=1, long_term_ref=1), dict(n=10, w=70, h=42, amp=1, inter_num_refs=2, global_mv_y=17),
         dict(n=10, b_frames=2, b_ref=1, temporal_mvp=1, inter_num_refs=2, open_gop=2), dict(n=6, temporal_mvp=0, log2_ctb=5), dict(n=5, bit_depth=10, b_frames=1, temporal_mvp=1)]
pool = []
for k, cfg in enumerate(SPECS):
    cfg = dict(cfg)
    n, w, h, bd = cfg.pop("n"), cfg.pop("w", 104), cfg.pop("h", 72), cfg.get("bit_depth", 8)
    aus = orc.encode_sequence(make_frames(w, h, n, bd), qp=28, global_mv_x=cfg.pop("global_mv_x", -8), global_mv_y=cfg.pop("global_mv_y", -4), seed=90 + k, **cfg)
    refs = orc.decode_sequence(aus)
    pool.append((aus, {r["poc"]: r for r in refs}, [r["poc"] for r in refs]))


def damaged(aus):
    aus = list(aus)
    k = rng.randrange(1, len(aus))
    b = bytearray(aus[k])
    for _ in range(rng.choice([1, 3, 8])):
        b[rng.randrange(6, len(b))] ^= 1 << rng.randrange(8)
    aus[k] = bytes(b)
    return aus


def play(aus, pauses, out):
    d = HipDecoder()
    got = []
    try:
        for k, au in enumerate(aus):
            if pauses[k]:
                time.sleep(pauses[k])
            d.push_data(au)
            r = d.next_picture(user_data=700 + k)
            while r is not None:
                got.append(r); r = d.next_picture()
        r = d.next_picture(flush=True)
        while r is not None:
            got.append(r); r = d.next_picture(flush=True)
        out.append(("ok", got))
    except HipDecError as e:
        out.append(("error", str(e)))
    finally:
        d.free()


t_start = time.time()
stats0 = chain_stats()
for rnd in range(rounds):
    lib.hipdec_set_sequence_lookahead(rng.choice([0, 1, 3, 8, 32]))
    jobs = []
    for t in range(n_threads):
        aus, by_poc, coding = pool[rng.randrange(len(pool))]
        bad = rng.random() < float(os.environ.get("STRESS_DAMAGED", "0.15"))
        pauses = [rng.choice([0, 0, 0, 0.001, 0.01, 0.05]) for _ in aus]
        jobs.append((damaged(aus) if bad else aus, by_poc, coding, bad, pauses, []))
    threads = [threading.Thread(target=play, args=(j[0], j[4], j[5])) for j in jobs]
    for th in threads: th.start()
    deadline = time.time() + 600
    for th in threads:
        th.join(max(1.0, deadline - time.time()))
        if th.is_alive():
            print("DEADLOCK / timeout in round", rnd); os._exit(2)
    for t, (aus, by_poc, coding, bad, pauses, out) in enumerate(jobs):
        kind, res = out[0]
        if bad:
            if kind == "ok":      # the damage may have hit bits that still decode: then the pictures may differ, nothing to check
                continue
            continue
        assert kind == "ok", "round %d track %d: a good track failed: %s" % (rnd, t, res)
        assert len(res) == len(aus), (rnd, t, len(res), len(aus))
        for out_idx, (img, ud) in enumerate(res):
            assert ud == 700 + coding.index(out_idx), (rnd, t, out_idx, ud)
            for c in range(3):
                assert (img.planes[c] == by_poc[out_idx]["planes"][c]).all(), "round %d track %d POC %d plane %d" % (rnd, t, out_idx, c)
lib.hipdec_set_sequence_lookahead(32)
s1 = chain_stats()
print("seed %d: %d rounds x %d threads ok in %.0f s; chains %d, launch sets %d, shared launch sets %d" %
      (seed, rounds, n_threads, time.time() - t_start, s1[0] - stats0[0], s1[1] - stats0[1], s1[2] - stats0[2]))
