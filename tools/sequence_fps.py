"""Frames per second of sequence tracks through the decoder object, the way libheif drives it (one sample per push, polling for the next picture in
output order; GPU box, dev tool): one 1280x720 track alone - every picture is one CABAC critical path, the instance has one sample at a time -
and T tracks decoded side by side by T threads (their decodes coalesce into shared launch sets).
usage: python tools/sequence_fps.py [frames] [tracks]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as orc
from test_inter_oracle import make_frames
from libheif_amd.decoder import HipDecoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
tracks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w, h = 1280, 720
frames = make_frames(w, h, n)
kinds = {"intra only": None,
         "lowdelay (IPPP, 2 refs, TMVP, weighted)": dict(inter_num_refs=2, temporal_mvp=1, weighted_pred=1),
         "unrestricted (IBBP, TMVP)": dict(b_frames=2, inter_num_refs=2, temporal_mvp=1)}
only = os.environ.get("SEQ_KIND")
for name, kw in kinds.items():
    if only and only not in name:
        continue
    if kw is None:
        aus = [orc.encode(f, qp=27, vui_matrix=6) for f in frames]
        aus = [aus[0]] + [b"".join(x for x in __import__("test_sequence_gpu")._nals(a) if (x[4] >> 1) & 63 < 32) for a in aus[1:]]
        ref = None
    else:
        aus = orc.encode_sequence(frames, qp=27, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=30, **kw)
        ref = {r["poc"]: r for r in orc.decode_sequence(aus)}

    def play(check):
        d = HipDecoder()
        got = 0
        for au in aus:
            d.push_data(au)
            r = d.next_picture()
            while r is not None:
                if check and ref is not None:
                    assert (r[0].planes[0] == ref[got]["planes"][0]).all(), "POC %d" % got
                got += 1
                r = d.next_picture()
        r = d.next_picture(flush=True)
        while r is not None:
            got += 1
            r = d.next_picture(flush=True)
        d.free()
        assert got == len(aus)

    # SEQ_SWEEP="pipeline:lookahead,..." (hipdec_set_sequence_pipeline / _lookahead between the runs); default: what the environment says
    from libheif_amd.decoder import chain_stats, set_sequence_pipeline
    import test_sequence_gpu as _ts
    sweep = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("SEQ_SWEEP", "").split(",") if c] or [None]
    for combo in sweep:
        tag = ""
        if combo:
            set_sequence_pipeline(combo[0]); _ts._set_lookahead(combo[1])
            tag = "pipeline %d look-ahead %2d: " % combo
        play(True)                                   # warm-up + correctness against the oracle
        t0 = time.perf_counter(); play(False); one = time.perf_counter() - t0
        before = chain_stats()
        th = [threading.Thread(target=play, args=(False,)) for _ in range(tracks)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        many = time.perf_counter() - t0
        after = chain_stats()
        bytes_per_frame = sum(len(a) for a in aus) / len(aus)
        print("%s%-42s %d x %dx%d pictures (%.0f KB per picture): 1 track %.1f fps (%.1f ms per picture); %d tracks side by side %.1f fps in total (%d chains in %d launch sets)" %
              (tag, name, n, w, h, bytes_per_frame / 1e3, n / one, one / n * 1e3, tracks, tracks * n / many, after[0] - before[0], after[1] - before[1]), flush=True)
