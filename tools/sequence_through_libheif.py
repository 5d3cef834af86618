"""Frames per second of a 720p sequence track through the REAL libheif (oracle/_ref/libheif.so: heif_track_decode_next_image, libheif's Track_Visual pushes the
samples into the plugin and polls it, sequences/track_visual.cc:175-330) with libheifhip.so as the decoder plugin - T application threads (tools/dropin_host.c,
no Python in the timed loop) each playing the same image-sequence file from its first to its last picture, again and again.  Host to host: file bytes in
memory -> every picture's planes in a heif_image.
usage: python tools/sequence_through_libheif.py [--frames 257] [--threads 1,16] [--seconds 5] [--pipeline 3,1]"""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(frames=257, threads_list=(1, 16), seconds=5.0, pipelines=(3, 1), w=1280, h=720, verbose=False, aus=None):
    from oracle import pyoracle as orc
    from test_inter_oracle import make_frames
    import heic_util as hu
    import libheif_amd
    ref = os.path.join(ROOT, "oracle", "_ref", "libheif.so")
    if not os.path.exists(ref):
        raise RuntimeError("oracle/_ref/libheif.so is not built (make -C oracle ref)")
    exe = os.path.join(ROOT, "build", "dropin_host")
    src = os.path.join(ROOT, "tools", "dropin_host.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-pthread", src, "-ldl", "-o", exe])
    if aus is None:
        aus = orc.encode_sequence(make_frames(w, h, frames), qp=27, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=30, inter_num_refs=2, temporal_mvp=1, weighted_pred=1,
                                  vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)   # (libheif's defaults: no colour conversion behind the decoder - context.cc:1533-1543)
    frames = len(aus)
    tmp = tempfile.mkdtemp(prefix="hipdec_seq_")
    pth = os.path.join(tmp, "track.heic")
    with open(pth, "wb") as f:
        f.write(hu.build_sequence(aus, w, h))
    out = {"workload": "one %dx%d IPPP track of %d pictures (2 refs, TMVP, weighted prediction; %.0f KB per picture) as an image-sequence file, T threads x heif_track_decode_next_image() "
                       "over the whole track through the real libheif + plugin, host to host" % (w, h, frames, sum(len(a) for a in aus) / len(aus) / 1e3), "runs": []}
    try:
        for D in pipelines:
            for T in threads_list:
                env = dict(os.environ, HIPDEC_SEQ_PIPELINE=str(D))
                r = subprocess.run([exe, ref, libheif_amd.library_path(), str(T), str(seconds), "0", pth], capture_output=True, text=True, timeout=seconds * 6 + 180, env=env)
                if r.returncode != 0:
                    out["runs"].append({"chains_in_flight": D, "threads": T, "error": (r.stderr or r.stdout)[-300:]})
                    continue
                n, dt, mpx = r.stdout.split()[-8:][:3]
                out["runs"].append({"chains_in_flight": D, "threads": T, "track_plays": int(n), "seconds": float(dt), "fps": round(float(mpx) * 1e6 / (w * h), 1)})
                if verbose: print(json.dumps(out["runs"][-1]), flush=True)
    finally:
        os.unlink(pth); os.rmdir(tmp)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=257)
    ap.add_argument("--threads", default="1,16")
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--pipeline", default="3,1")
    a = ap.parse_args()
    res = measure(a.frames, [int(x) for x in a.threads.split(",")], a.seconds, [int(x) for x in a.pipeline.split(",")], verbose=True)
    print(json.dumps(res))
