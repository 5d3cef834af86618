"""Stress of the still-image coalescer's HOST logic on the emulated library (tests/emu/libheifhip_emu.so; no GPU): many threads, each decoding stills through
its own decoder instances the way libheif's grid / multi-threaded hosts do - mixed sizes and bit depths (8-bit and 10-bit requests never share a launch set:
the group is split), some streams damaged (a bad item is isolated by halving the group), random pauses - no deadlock, every good still equals the oracle's.
usage: HIPDEC_LIBRARY=tests/emu/libheifhip_emu.so HIPDEC_DEV_AB=1 python tools/emu_stress_stills.py <seed> <rounds> [threads]"""
import os, random, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc
from libheif_amd import HipDecError
from libheif_amd.decoder import HipDecoder, coalesce_stats

seed, rounds = int(sys.argv[1]), int(sys.argv[2])
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = random.Random(seed)
pool = []
for k, (w, h, cfg) in enumerate([(200, 136, dict()), (136, 72, dict(stress=1, wpp=0, log2_ctb=4, log2_max_tb=4)), (264, 136, dict(tile_cols=3, tile_rows=2)),
                                 (70, 42, dict()), (200, 136, dict(bit_depth=10)), (128, 72, dict(num_slices=3)), (96, 64, dict(chroma_format_idc=0))]):
    cf = cfg.pop("chroma_format_idc", 1)
    s = orc.encode(orc.synth_image(w, h, cfg.get("bit_depth", 8), cf, seed=20 + k), **cfg)
    pool.append((s, orc.decode(s)))


def work(jobs, out):
    for s, ref, pause in jobs:
        if pause:
            time.sleep(pause)
        d = HipDecoder()
        try:
            d.push_data(s)
            img = d.decode_next_image()
            out.append(("ok", img, ref))
        except HipDecError as e:
            out.append(("error", str(e), ref))
        finally:
            d.free()


t0 = time.time()
c0 = coalesce_stats()
for rnd in range(rounds):
    plans, outs = [], []
    for t in range(n_threads):
        jobs = []
        for _ in range(rng.choice([1, 2, 4])):
            s, ref = pool[rng.randrange(len(pool))]
            bad = rng.random() < float(os.environ.get("STRESS_DAMAGED", "0.12"))
            if bad:
                b = bytearray(s)
                for _ in range(rng.choice([1, 4, 16])):
                    b[rng.randrange(len(b) // 2, len(b))] ^= 1 << rng.randrange(8)
                s, ref = bytes(b), None
            jobs.append((s, ref, rng.choice([0, 0, 0.0005, 0.003])))
        plans.append(jobs); outs.append([])
    th = [threading.Thread(target=work, args=(plans[t], outs[t])) for t in range(n_threads)]
    for x in th: x.start()
    deadline = time.time() + 600
    for x in th:
        x.join(max(1.0, deadline - time.time()))
        if x.is_alive():
            print("DEADLOCK / timeout in round", rnd); os._exit(2)
    for t in range(n_threads):
        assert len(outs[t]) == len(plans[t])
        for kind, res, ref in outs[t]:
            if ref is None:
                continue      # a damaged stream: an error, or pictures that may differ
            assert kind == "ok", "round %d: a good still failed: %s" % (rnd, res)
            for c in range(len(ref["planes"])):
                assert (res.planes[c] == ref["planes"][c]).all(), "round %d thread %d plane %d" % (rnd, t, c)
c1 = coalesce_stats()
print("seed %d: %d rounds x %d threads ok in %.0f s; requests %d, launch sets %d, requests that shared a set %d" %
      (seed, rounds, n_threads, time.time() - t0, c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2]))
