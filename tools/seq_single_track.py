"""One 720p IPPP track through the decoder object (dev tool for kernel timelines: rocprofv3 --kernel-trace -- python tools/seq_single_track.py [frames])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc
from test_inter_oracle import make_frames
from libheif_amd.decoder import HipDecoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 97
frames = make_frames(1280, 720, n)
aus = orc.encode_sequence(frames, qp=27, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=30, inter_num_refs=2, temporal_mvp=1, weighted_pred=1)
def play():
    d = HipDecoder(); got = 0
    for au in aus:
        d.push_data(au)
        r = d.next_picture()
        while r is not None:
            got += 1; r = d.next_picture()
    r = d.next_picture(flush=True)
    while r is not None:
        got += 1; r = d.next_picture(flush=True)
    d.free(); assert got == len(aus)
play()
t0 = time.perf_counter(); play(); dt = time.perf_counter() - t0
print("%d pictures: %.1f fps" % (n, n / dt), flush=True)
