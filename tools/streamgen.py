"""Synthetic HEVC-intra input generator for bench.py and the tests.

There is no HEVC encoder in the image, so the synthetic streams BASELINE.json's configs call for
are produced by the test-only encoder that lives beside the oracle (oracle/hevc_testenc.c).  This
module only *generates inputs* (outside any timed region); it is never part of the decode path.
Streams are cached on disk (they are deterministic functions of their parameters).
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CACHE = os.environ.get("HIPDEC_STREAM_CACHE", "/tmp/hipdec_streams")


def _key(w, h, bit_depth, seed, cfg):
    s = repr((w, h, bit_depth, seed, sorted(cfg.items()), 2))
    return hashlib.sha1(s.encode()).hexdigest()[:20]


def stream_path(w, h, seed, bit_depth, cfg):
    return os.path.join(CACHE, "s_%dx%d_%s.hevc" % (w, h, _key(w, h, bit_depth, seed, cfg)))


def make_stream(w, h, seed=1, bit_depth=8, **cfg):
    """One coded picture in libheif's plugin framing ([u32 BE length][NAL]...), cached."""
    os.makedirs(CACHE, exist_ok=True)
    path = stream_path(w, h, seed, bit_depth, cfg)
    if os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    from oracle import pyoracle as orc
    cfg = dict(cfg)
    planes = orc.synth_image(w, h, bit_depth, cfg.pop("chroma_format_idc", 1), seed=seed)    # (the encoder takes the chroma format from the plane shapes)
    data = orc.encode(planes, bit_depth=bit_depth, **cfg)
    tmp = path + ".%d.tmp" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)
    return data


def _job(a):
    w, h, seed, bd, cfg = a
    return make_stream(w, h, seed, bd, **cfg)


def make_streams(specs, workers=None):
    """specs: list of (w, h, seed, bit_depth, cfg dict).  Generated in parallel processes."""
    todo = list(specs)
    if len(todo) <= 1 or all(os.path.exists(stream_path(*a)) for a in todo):
        return [_job(a) for a in todo]
    import multiprocessing as mp
    workers = min(len(todo), workers or max(1, (os.cpu_count() or 2) - 1), 192)
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_job, todo)
